// kern_xcd.hpp — fused two-pass transforms: both passes of an N = N1*N2 four-step (or of a 2-D plane) in ONE persistent launch.
//
// Why (profiles/r01_xcd_fused_ab.log): the two-kernel route runs the whole chip in lock-step (all CUs
// load, then all compute, then all store, one launch pair per chunk).  Here groups of workgroups walk the batch
// independently, drift out of phase with each other and keep the fabric busy in both directions; the intermediate of a
// transform lives in a small per-group workspace slot that is re-used for every transform.  Measured at N = 2^20: 186
// GPoints/s against 155 for the two-kernel route.  (The intermediate still crosses the fabric twice — 8 MiB per transform
// does not survive in a 4 MiB L2 — see the PMC summary under profiles/.)
//
// Modes:
//   shared (transforms > 1 MiB)  the workgroups of one XCD — one per CU, two where 256 threads and <= 80 KB of LDS leave room —
//                 are divided into `split` groups; a group works on one transform at a time:
//                   registration: every workgroup reads its XCC id, takes a rank inside that XCD and waits (bounded) until all
//                                 gridDim.x workgroups have registered.  Nothing assumes a placement: a group is BY
//                                 CONSTRUCTION a set of workgroups behind one L2.
//                   per transform t (group g takes t = g, g+G, ...):
//                     phase A  column tiles r, r+s, ... of x_t -> W      (kern_lines.hpp PASS_A stages)
//                     group barrier
//                     phase B  row tiles r, r+s, ... of W -> out_t       (PASS_B stages, four-step roots generated per tile;
//                                                                         TWO_D: ROW stages, natural order, no roots)
//                   (W is double-buffered, so the A/B barrier of the next transform is the only one needed)
//                 Hand-off protocol (same-XCD by construction, MI355X_MICROARCH.md "Workgroup dispatch ... visibility"):
//                 producers' stores are complete in the shared L2 after `s_waitcnt vmcnt(0)`; one lane per workgroup adds to
//                 the group's monotonic counter and polls it with relaxed agent-scope loads; every consumer workgroup then
//                 invalidates its CU's L1 with an agent-scope acquire before reading.  No L2 write-back is needed because
//                 producer and consumer share that L2.  Every spin is bounded: on a timeout the workgroup raises a sticky
//                 error word and returns (queue_wait reports it).  All workgroups must be co-resident.
//   solo (transforms <= 1 MiB)  every workgroup walks whole transforms alone: phase A into its own slot, a workgroup barrier
//                 + L1 invalidate, phase B out of it.  No registration, no cross-workgroup synchronisation, any grid size.
#pragma once
#include "kern_lines.hpp"

#ifndef MI355_XCD_NT
#define MI355_XCD_NT 1   /* same-box A/B with two groups per XCD: 185-188 GPoints/s on, 175-182 off (profiles/r01_xcd_fused_ab.log) */
#endif

#ifndef MI355_XCD_W_NT_ST
#define MI355_XCD_W_NT_ST 0   /* experiment: nontemporal stores of the intermediate */
#endif
#ifndef MI355_XCD_W_NT_LD
#define MI355_XCD_W_NT_LD 0   /* experiment: nontemporal loads of the intermediate */
#endif

namespace mi355 {

constexpr bool XCD_NT = MI355_XCD_NT != 0;   // nontemporal x loads / output stores in the fused kernel

struct XcdCtl {                  // zeroed by a memset step before every launch
  unsigned long long reg_packed; // registration: byte x = workgroups registered with XCC id x (x < 8).  ONE word, so that a workgroup
                                 // that sees the grid complete has by construction also seen every per-XCD count (two counters
                                 // updated by relaxed atomics could be observed out of step)
  unsigned reg_pad[15];
  unsigned bar[512][16];         // one 64-byte line per group (XCC id x split): [0] A->B barrier, [1] B->A barrier (one-slot mode)
};

struct XcdFusedArgs {
  const cf* in;
  cf* out;
  cf* wslots;                    // 32*split slots of N points (two per group, `split` groups per XCC id)
  XcdCtl* ctl;
  unsigned* sticky_error;        // device-wide error word (bit 0: registration timeout, bit 1: barrier timeout)
  const cf* tw_a;                // stage tables of the pass A / pass B line configs
  const cf* tw_b;
  const cf* tw_lo;               // four-step roots: e^{-2 pi i m/N} = HI[m >> shift] * LO[m & mask]
  const cf* tw_hi;
  long long num_transforms;
  long long N;                   // N1 * N2 (r2c: the real length)
  long long in_pitch, out_pitch; // complex elements between consecutive transforms (c2c: N, N; r2c: N/2, N/2 + 1)
  float scale;
  int fs_shift;
  unsigned fs_lo_mask;
  unsigned spin_limit;           // polls before a wait gives up
  unsigned split;                // groups per XCD (1..32): the workgroups of an XCD are divided by rank
  unsigned slots;                // workspace slots per group: 2 (one barrier per transform) or 1 (two barriers, half the footprint)
  unsigned solo;                 // 1: every workgroup is its own group (transforms of <= 1 MiB): no registration, no cross-
                                 // workgroup barrier, no co-residency requirement — the grid may be any size, one slot per workgroup
  // fftconv pipeline (kern_regtile.hpp fft_xcd_conv1m_kernel): K kernel spectra of N points each, back to back
  const cf* mul;
  unsigned conv_k, conv_conj;    // kernels per data line; 1: correlation (conjugated spectra)
  long long out_kernel_pitch;    // complex elements between the outputs of consecutive kernels of one data line (out_pitch: between data lines)
  // rank-1 views of the VIEW instances (ioView / zeroPad of a four-step line as predicates of the first loads and the last stores, SideMap
  // semantics of plan.hpp): element i of a line is read inside [v_in_lo, v_in_hi) and is 0 elsewhere; element k is stored inside
  // [v_out_lo, v_out_hi) only, as 0 outside [v_zlo, v_zhi).  `in` / `out` already carry the maps' offsets.
  int v_in_lo, v_in_hi, v_out_lo, v_out_hi, v_zlo, v_zhi;
};

// roots for one PASS_B tile, generated per tile: anchors by exact table lookup every 8th element, the 7 in between by
// multiplying with the per-thread step e^{-2 pi i (N2/R0) k1/N} (error <= 8 roundings; the hoisted form of
// kern_lines.hpp would need 64 VGPRs per row tile, and a workgroup owns two)
template <class C>
MI_DEV void fourstep_apply_chain(cf (&v)[C::E], const XcdFusedArgs& a, unsigned k1, int u) {
  using I = StageInfo<C, 0>;
  constexpr int NA = (I::R + 7) / 8;   // anchors per butterfly
  const auto root = [&](unsigned m) { return cmul(a.tw_hi[m >> a.fs_shift], a.tw_lo[m & a.fs_lo_mask]); };
  const cf step = root(k1 * (unsigned)(C::N / I::R));
#pragma unroll
  for (int b = 0; b < I::NB; ++b) {
    cf anchor[NA];
#pragma unroll
    for (int g = 0; g < NA; ++g) anchor[g] = root(k1 * (unsigned)(u + b * C::TPL + 8 * g * (C::N / I::R)));
#pragma unroll
    for (int g = 0; g < NA; ++g) {
      cf w = anchor[g];
#pragma unroll
      for (int j = 0; j < 8 && 8 * g + j < I::R; ++j) {
        cf& x = v[b * I::R + 8 * g + j];
        x = cmul(x, w);
        if (j < 7) w = cmul(w, step);
      }
    }
  }
}

// split barrier over the workgroups of one XCD.  arrive: this workgroup's stores are complete in the shared L2
// (s_waitcnt vmcnt(0) in every wave, then the workgroup barrier), one lane bumps the group counter.  wait: one lane polls
// (relaxed agent-scope loads, bounded), then an agent-scope acquire drops this CU's L1 before anyone reads.
// MI355_XCD_RELEASE=1 (A/B builds only): the form the hardware guide lists — lane 0 issues an agent-scope release (buffer_wbl2:
// write back this XCD's dirty L2 lines) before the add.  The shipped form omits it: producer and consumers of a group sit behind
// the SAME L2 by construction (groups are formed from the XCC_ID register), so the stores are visible to them once they have left
// the CU (vmcnt(0)), and writing W back to memory first is exactly the fabric traffic the kernel is trying not to wait for.
// Measured: profiles/r02_xcd_handoff_release_ab.log.
#ifndef MI355_XCD_RELEASE
#define MI355_XCD_RELEASE 0
#endif
MI_DEV void xcd_arrive(unsigned* counter) {
  MI_WAIT_VMEM();
  __syncthreads();
  if (threadIdx.x == 0) {
#if MI355_XCD_RELEASE && !defined(MI355_HOST_EMU)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    MI_WAIT_VMEM();
#endif
    MI_ATOMIC_ADD_U32(counter, 1u);
  }
}
// solo mode: producer and consumer are the same workgroup — its stores are complete (vmcnt(0) + workgroup barrier) and its
// CU's L1 is invalidated (the slot is re-used for every transform, so stale lines of the previous one may sit there)
MI_DEV void xcd_local_handoff() {
  MI_WAIT_VMEM();
  __syncthreads();
  MI_ACQUIRE_AGENT();
}
#ifndef MI355_EXP_NO_WAIT
#define MI355_EXP_NO_WAIT 0   /* timing-only builds: the group barriers do not wait (results wrong by construction): bounds what the waits cost */
#endif
MI_DEV bool xcd_wait(unsigned* counter, unsigned target, unsigned spin_limit, unsigned* sticky, unsigned* s_flag) {
  if (threadIdx.x == 0) {
    unsigned ok = MI355_EXP_NO_WAIT;
    for (unsigned it = 0; it < spin_limit && !MI355_EXP_NO_WAIT; ++it) {
      if (MI_ATOMIC_LOAD_U32(counter) >= target) { ok = 1; break; }
      MI_SLEEP();
    }
    MI_ACQUIRE_AGENT();
    if (!ok) MI_ATOMIC_OR_U32(sticky, 2u);
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// Registration: every workgroup reads its XCC id, takes a rank inside that XCD and waits (bounded) until the whole grid has
// registered, then derives its group from the registered counts.  s_words (LDS): [0] group slot, [1] rank, [2] group size,
// [3] ok, [4] group index, [5] groups, [6] barrier flag.  Returns false on timeout (sticky error bit 0 raised).
MI_DEV bool xcd_register(XcdCtl* ctl, unsigned split_arg, unsigned spin_limit, unsigned* sticky, unsigned* s_words) {
  if (threadIdx.x == 0) {
    const unsigned x = MI_XCC_ID() & 15u;
    unsigned ok = 0, r = 0;
    unsigned long long snap = 0;
    // (a byte per XCC id: a 256th registration on one id would carry into its neighbour's count and mis-group silently, so grids
    // beyond 8 x 255 workgroups and a full byte are refused: sticky bit 4 -> queue_wait reports it)
    if (x < 8u && gridDim.x <= 8u * 255u) {   // gfx950 has 8 XCDs; anything else is reported, never mis-grouped
      const unsigned long long old = MI_ATOMIC_ADD_U64(&ctl->reg_packed, 1ull << (8u * x));
      r = (unsigned)(old >> (8u * x)) & 0xffu;
      if (r == 0xffu) MI_ATOMIC_OR_U32(sticky, 16u);
      for (unsigned it = 0; it < spin_limit; ++it) {
        snap = MI_ATOMIC_LOAD_U64(&ctl->reg_packed);
        unsigned total = 0;
        for (unsigned k = 0; k < 8; ++k) total += (unsigned)(snap >> (8u * k)) & 0xffu;
        if (total >= gridDim.x) { ok = 1; break; }
        MI_SLEEP();
      }
    }
    // an XCD's workgroups may be split into up to 8 groups by rank (each with its own workspace slots and barrier counter):
    // smaller groups run out of phase with each other inside one XCD at the price of sharing its L2
    const unsigned split = split_arg ? split_arg : 1u;
    unsigned groups = 0, gi = 0, mine = 0, sub = 0, gsz = 0;
    for (unsigned k = 0; k < 8; ++k) {
      const unsigned cnt = (unsigned)(snap >> (8u * k)) & 0xffu;
      if (!cnt) continue;
      const unsigned per = (cnt + split - 1u) / split, nsub = (cnt + per - 1u) / per;
      if (k == x) {
        sub = r / per; mine = r - sub * per;
        gsz = (sub + 1u) * per <= cnt ? per : cnt - sub * per;
        gi = groups + sub;
      }
      groups += nsub;
    }
    s_words[0] = x * split + sub; s_words[1] = mine; s_words[2] = gsz; s_words[3] = ok; s_words[4] = gi; s_words[5] = groups;
    if (!ok) MI_ATOMIC_OR_U32(sticky, gridDim.x > 8u * 255u ? 16u : 1u);
  }
  __syncthreads();
  return s_words[3] != 0;
}

// Tile order inside a group.  Tiles narrower than 16 lines move 64-byte segments; a workgroup then takes PAIRS of adjacent
// tiles back to back (and leaves their accesses temporal) so that the two halves of every 128-byte line meet in the L2
// within a few microseconds instead of crossing the fabric as partial lines.
#ifndef MI355_XCD_PAIR
#define MI355_XCD_PAIR 1
#endif
// Whether the paired tiles keep nontemporal accesses is measured per kernel (profiles/r01_xcd_fused_ab.log): the c2c kernel
// is faster with them (147.6 vs 141.1 GPoints/s at N = 2^21), the r2c kernel without (222 vs 202 at N = 2^21).
template <class Cfg, bool PAIR_NT = true> struct PairOf {
  static constexpr int C = (MI355_XCD_PAIR && Cfg::T * 8 < 128) ? 128 / (Cfg::T * 8) : 1;
  static constexpr bool NT = XCD_NT && (C == 1 || PAIR_NT);
};
template <class C> MI_DEV long long xcd_tile(long long i, unsigned rank, unsigned gsize) {
  constexpr int P = PairOf<C>::C;
  return ((i / P) * (long long)gsize + rank) * P + (i % P);
}

// both passes use the same radix plan: one copy of the stage tables serves both
template <class CA, class CB> struct XcdTables {
  static constexpr bool SHARED = CA::N == CB::N && CA::R0 == CB::R0 && CA::R1 == CB::R1 && CA::R2 == CB::R2;
  static constexpr int ELEMS = CA::TW_ELEMS + (SHARED ? 0 : CB::TW_ELEMS);
};

template <class CA, class CB, bool VIEW = false>
__global__ void __launch_bounds__(CA::THREADS) fft_xcd_fused_kernel(const XcdFusedArgs f) {
  static_assert(CA::THREADS == CB::THREADS, "both passes run in the same workgroup");
  static_assert(CA::IN_COL && CA::OUT_COL && !CB::IN_COL, "PASS_A, then PASS_B (four-step) or ROW (two-dimensional)");
  // TWO_D: a 2-D transform of an [N1][N2] array is the same two passes without the four-step roots and with the rows written
  // back in natural order (the second pass is a ROW configuration)
  constexpr bool TWO_D = !CB::OUT_COL;
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  constexpr int DATA = CA::DATA_ELEMS > CB::DATA_ELEMS ? CA::DATA_ELEMS : CB::DATA_ELEMS;
  using TB = XcdTables<CA, CB>;
  cf* tw_a = lds + DATA;
  cf* tw_b = TB::SHARED ? tw_a : tw_a + CA::TW_ELEMS;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw_a + TB::ELEMS);
  const int t = threadIdx.x;
  for (int i = t; i < CA::TW_ELEMS; i += CA::THREADS) tw_a[i] = f.tw_a[i];
  if constexpr (!TB::SHARED) { for (int i = t; i < CB::TW_ELEMS; i += CA::THREADS) tw_b[i] = f.tw_b[i]; }

  if (f.solo) {
    if (t == 0) { s_words[0] = blockIdx.x; s_words[1] = 0; s_words[2] = 1; s_words[3] = 1; s_words[4] = blockIdx.x; s_words[5] = gridDim.x; }
    __syncthreads();
  } else if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  const long long N1 = CA::N, N2 = CB::N;
  LineArgs aa{}, ab{};
  aa.tw = f.tw_a; aa.num_tiles = N2 / CA::T; aa.num_lines = N2;
  aa.in_S = N2; aa.in_outer_stride = f.N; aa.out_S = N2; aa.out_outer_stride = f.N; aa.scale = 1.0f; aa.fs_group = 1;
  ab.tw = f.tw_b; ab.num_tiles = N1 / CB::T; ab.num_lines = N1;
  ab.in_S = 1; ab.in_outer_stride = N2; ab.out_S = TWO_D ? 1 : N1; ab.out_outer_stride = TWO_D ? N2 : f.N; ab.scale = f.scale; ab.fs_group = N1;
  if constexpr (VIEW) {
    static_assert(!TWO_D && CB::NSTAGES >= 2 && CB::NSTAGES <= 3, "rank-1 views of four-step lines");
    aa.v_in_lo = f.v_in_lo; aa.v_in_hi = f.v_in_hi; ab.v_out_lo = f.v_out_lo; ab.v_out_hi = f.v_out_hi; ab.v_zlo = f.v_zlo; ab.v_zhi = f.v_zhi;
  }
  // Two workspace slots per group, alternated per transform: the barrier between A(k+1) and B(k+1) also orders "everyone
  // finished reading slot s in B(k)" before "anyone overwrites slot s in A(k+2)" — one group barrier per transform.
  // (Measured: running A(k+1) ahead of the wait for barrier k, to hide the barrier, loses more than it gains: 150 vs 165
  // GPoints/s at N = 2^20.)
  const bool two_slots = f.slots != 1u && !f.solo;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)f.N;
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)f.N;
    // ---- phase A: column FFTs of transform tr into this XCD's workspace slot ----
    aa.in = f.in + tr * f.in_pitch; aa.out = W;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CA>(i, rank, gsize);
      if (tile - (i % PairOf<CA>::C) >= aa.num_tiles) break;
      if (tile >= aa.num_tiles) continue;
      cf v[CA::E];
      stage_read<CA, 0, PairOf<CA>::NT && !VIEW, VIEW>(v, aa, tile, t, lds);       // x streams past the L2
      stage_compute_write<CA, 0>(v, aa, tile, t, lds, tw_a, nullptr);
      if constexpr (CA::NSTAGES >= 2) {
        __syncthreads();
        stage_read<CA, 1>(v, aa, tile, t, lds);
        __syncthreads();
        stage_compute_write<CA, 1, MI355_XCD_W_NT_ST != 0>(v, aa, tile, t, lds, tw_a, nullptr);
      }
      if constexpr (CA::NSTAGES == 3) {
        __syncthreads();
        stage_read<CA, 2>(v, aa, tile, t, lds);
        __syncthreads();
        stage_compute_write<CA, 2, MI355_XCD_W_NT_ST != 0>(v, aa, tile, t, lds, tw_a, nullptr);
      }
      __syncthreads();   // LDS is re-used by the next tile
    }
    if (f.solo) xcd_local_handoff();
    else {
      xcd_arrive(&f.ctl->bar[gslot][0]);
      if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
    // ---- phase B: four-step roots on load, row FFTs, transposed store to the output ----
    ab.in = W; ab.out = f.out + tr * f.out_pitch;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CB>(i, rank, gsize);
      if (tile - (i % PairOf<CB>::C) >= ab.num_tiles) break;
      if (tile >= ab.num_tiles) continue;
      cf v[CB::E];
      stage_read<CB, 0, MI355_XCD_W_NT_LD != 0>(v, ab, tile, t, lds);
      if constexpr (!TWO_D) {
        int line, u; thread_map<CB, 0>(t, line, u);
        fourstep_apply_chain<CB>(v, f, (unsigned)(tile * CB::T + line), u);
      }
      stage_compute_write<CB, 0, PairOf<CB>::NT>(v, ab, tile, t, lds, tw_b, nullptr);
      if constexpr (CB::NSTAGES >= 2) {
        __syncthreads();
        stage_read<CB, 1>(v, ab, tile, t, lds);
        __syncthreads();
        stage_compute_write<CB, 1, PairOf<CB>::NT, false, false, VIEW && CB::NSTAGES == 2>(v, ab, tile, t, lds, tw_b, nullptr);   // output streams past the L2
      }
      if constexpr (CB::NSTAGES == 3) {
        __syncthreads();
        stage_read<CB, 2>(v, ab, tile, t, lds);
        __syncthreads();
        stage_compute_write<CB, 2, PairOf<CB>::NT, false, false, VIEW>(v, ab, tile, t, lds, tw_b, nullptr);
      }
      __syncthreads();
    }
    if (f.solo) __syncthreads();   // this workgroup has read its slot before its next phase A overwrites it
    else if (!two_slots) {   // everyone has read the slot before phase A of the next transform overwrites it
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

template <class CA, class CB> struct XcdFusedCfg {
  static constexpr int DATA = CA::DATA_ELEMS > CB::DATA_ELEMS ? CA::DATA_ELEMS : CB::DATA_ELEMS;
  static constexpr int LDS_BYTES = (DATA + XcdTables<CA, CB>::ELEMS) * 8 + 64;
  static constexpr int THREADS = CA::THREADS;
  static_assert(LDS_BYTES <= 160 * 1024, "fused tile does not fit LDS");
};

}  // namespace mi355
