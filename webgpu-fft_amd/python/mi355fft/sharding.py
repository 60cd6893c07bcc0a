"""Batch sharding across the GPUs of one node: one process per GPU, rank r owns the
contiguous transforms [r*B, (r+1)*B) of the global batch; there is no data-path collective — transforms
never interact (kernels/nd_line_base.js:38-40).  torch.distributed (RCCL on GPUs, gloo in the CPU tests)
is used only for barriers around the timed region and for reducing elapsed time / error norms."""
import os


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank, world, global_batch):
    """contiguous, balanced: the first (global_batch % world) ranks take one extra transform"""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class Group:
    """thin wrapper so bench.py (nccl) and the CPU tests (gloo) share one code path"""

    def __init__(self, backend=None, device=None, allow_fallback=False):
        self.rank, self.local_rank, self.world = rank_info()
        self.active = self.world > 1
        self.device = device
        self.dist = None
        self.backend = None
        if self.active:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            self.backend = backend or "gloo"
            try:
                kwargs = {}
                if self.backend == "nccl" and device is not None:
                    kwargs["device_id"] = device
                dist.init_process_group(self.backend, **kwargs)
            except Exception as e:
                # RCCL unavailable.  Only the single-GPU rehearsal (ranks sharing one device: allow_fallback) may carry on over
                # gloo — the collectives here are barriers and scalar reductions, so CPU tensors are an exact substitute there;
                # on a real multi-GPU run the failure is reported, never papered over.
                if not allow_fallback or self.backend == "gloo":
                    raise
                import sys
                print(f"[mi355fft.sharding] {self.backend} init failed ({e}); using gloo for barriers/reductions", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                self.backend = "gloo"
                self.device = None
                dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def _reduce(self, values, op):
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        if self.active:
            self.dist.all_reduce(t, op=op)
        return [float(v) for v in t]

    def reduce_max(self, values):
        import torch.distributed as dist
        return self._reduce(values, dist.ReduceOp.MAX)

    def reduce_sum(self, values):
        import torch.distributed as dist
        return self._reduce(values, dist.ReduceOp.SUM)

    def close(self):
        if self.active:
            self.dist.destroy_process_group()
