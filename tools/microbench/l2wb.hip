// l2wb.hip — does an XCD's L2 keep re-written lines (write-back) inside ONE kernel, and do same-XCD readers hit them?
// Development probe for the XCD-resident four-step.  Per XCD a region of S bytes is split over the
// XCD's 32 workgroups.  Modes:
//   0  every workgroup re-writes its own slice R times                                (write-only: WRITE_SIZE / launch vs S*R)
//   1  write own slice, XCD barrier, read the slice of rank+1 with sc1 loads, barrier (exchange: FETCH_SIZE / launch vs S*R)
//   2  as 1 but plain loads behind an agent acquire (buffer_inv sc1) instead of sc1 loads
//   3  as 1 while the same workgroup also streams 2*slice bytes HBM->HBM per repetition (nontemporal), like the real kernel
// Run plain for timings; under `rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE` (separate passes) for bytes per dispatch:
// dispatches appear in the order of the lines printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xf; }
struct Ctl { unsigned reg_total; unsigned pad0[15]; unsigned reg_xcd[16]; unsigned bar[8][16]; unsigned err; };

__device__ __forceinline__ f4 ld_sc1(const f4* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// fence-free barrier over the 32 workgroups of one XCD (monotonic counter in the shared L2)
__device__ bool xbar(Ctl* c, unsigned xcc, unsigned target, unsigned* s_ok, bool acquire) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&c->bar[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned ok = 0;
    for (int it = 0; it < 2000000; ++it) {
      if (__hip_atomic_load(&c->bar[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (acquire) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if (!ok) atomicAdd(&c->err, 1u);
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

template <int MODE>
__global__ void __launch_bounds__(512) k_l2(f4* buf, size_t region_f4, int reps, Ctl* c, const f4* hin, f4* hout, size_t stream_f4_per_wg, float* sink) {
  __shared__ unsigned s_x, s_r, s_ok;
  if (threadIdx.x == 0) {
    s_x = xcc_id() & 7;
    s_r = __hip_atomic_fetch_add(&c->reg_xcd[s_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = 0;
    for (int it = 0; it < 2000000; ++it) { if (__hip_atomic_load(&c->reg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { s_ok = 1; break; } __builtin_amdgcn_s_sleep(4); }
    if (s_ok && __hip_atomic_load(&c->reg_xcd[s_x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32) s_ok = 0;
    if (!s_ok) atomicAdd(&c->err, 1u);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned xcc = s_x, r = s_r;
  const size_t slice = region_f4 / 32;                    // f4 elements per workgroup
  f4* mine = buf + (size_t)xcc * region_f4 + (size_t)r * slice;
  const f4* theirs = buf + (size_t)xcc * region_f4 + (size_t)((r + 1) & 31) * slice;
  const f4* sin = hin + ((size_t)blockIdx.x * reps) * stream_f4_per_wg;
  f4* sout = hout + ((size_t)blockIdx.x * reps) * stream_f4_per_wg;
  float acc = 0.f;
  unsigned bar = 0;
  for (int rep = 0; rep < reps; ++rep) {
    f4 v; v.x = (float)rep; v.y = (float)threadIdx.x; v.z = 1.f; v.w = 2.f;
    for (size_t i = threadIdx.x; i < slice; i += 512) mine[i] = v;
    if (MODE == 3) {
      for (size_t i = threadIdx.x; i < stream_f4_per_wg; i += 512) {
        f4 s = __builtin_nontemporal_load(sin + (size_t)rep * stream_f4_per_wg + i);
        s.x += 1.f;
        __builtin_nontemporal_store(s, sout + (size_t)rep * stream_f4_per_wg + i);
      }
    }
    if (MODE >= 1) {
      if (!xbar(c, xcc, (++bar) * 32, &s_ok, MODE == 2)) return;
      for (size_t i = threadIdx.x; i < slice; i += 512) {
        f4 w = (MODE == 2) ? theirs[i] : ld_sc1(theirs + i);
        acc += w.x + w.w;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!xbar(c, xcc, (++bar) * 32, &s_ok, false)) return;
    } else {
      asm volatile("" ::: "memory");
    }
  }
  if (acc == 123.456f) *sink = acc;
}

int main() {
  const int REPS = 64;
  const size_t MAXS = (size_t)16 << 20;
  f4* buf; Ctl* c; float* sink; f4 *hin, *hout;
  const size_t stream_per_wg = (size_t)128 << 10;              // mode 3: 128 KiB in + 128 KiB out per workgroup per repetition (the real kernel's ratio per exchange round)
  CK(hipMalloc(&buf, MAXS * 8)); CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&hin, stream_per_wg * 256 * REPS)); CK(hipMalloc(&hout, stream_per_wg * 256 * REPS));
  CK(hipMemset(buf, 0, MAXS * 8)); CK(hipMemset(hin, 0, stream_per_wg * 256 * REPS)); CK(hipMemset(hout, 0, stream_per_wg * 256 * REPS));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {(size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)3 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)16 << 20};
  int disp = 0;
  for (int mode = 0; mode < 4; ++mode) {
    for (size_t S : sizes) {
      const size_t region_f4 = S / 16;
      CK(hipMemset(c, 0, sizeof(Ctl)));
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_l2<0>, dim3(256), dim3(512), 0, 0, buf, region_f4, REPS, c, hin, hout, stream_per_wg / 16, sink); break;
        case 1: hipLaunchKernelGGL(k_l2<1>, dim3(256), dim3(512), 0, 0, buf, region_f4, REPS, c, hin, hout, stream_per_wg / 16, sink); break;
        case 2: hipLaunchKernelGGL(k_l2<2>, dim3(256), dim3(512), 0, 0, buf, region_f4, REPS, c, hin, hout, stream_per_wg / 16, sink); break;
        default: hipLaunchKernelGGL(k_l2<3>, dim3(256), dim3(512), 0, 0, buf, region_f4, REPS, c, hin, hout, stream_per_wg / 16, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      Ctl hc; CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
      const double wbytes = (double)S * 8 * REPS;
      printf("dispatch %2d mode %d S/XCD=%5zu KiB reps=%d: %8.3f ms  %7.2f us/rep  written %7.1f MiB/launch (%.0f GB/s of stores%s)  err=%u\n", disp++, mode, S >> 10, REPS, ms,
             ms * 1e3 / REPS, wbytes / 1048576.0, wbytes / ms / 1e6, mode >= 1 ? ", same again read" : "", hc.err);
    }
  }
  return 0;
}
