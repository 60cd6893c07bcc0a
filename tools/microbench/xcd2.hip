// Prototype of the data movement + synchronisation skeleton of an "XCD-resident four-step" kernel (no FFT math):
// per XCD one N=2^20 transform at a time; 32 workgroups x 512 threads; column tiles in, exchange through a 4 MiB L2-resident
// buffer in two rounds, transposed rows out.  Measures us per transform to decide whether the real kernel is worth building.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xf; }
struct Ctl { unsigned reg_total; unsigned reg_xcd[8]; unsigned bar[8][16]; unsigned err; };

__device__ bool xcd_barrier(Ctl* c, unsigned xcc, unsigned target) {
  __shared__ unsigned s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&c->bar[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned ok = 0;
    for (int it = 0; it < 4000000; ++it) {
      if (__hip_atomic_load(&c->bar[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!ok) atomicAdd(&c->err, 1u);
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

template <int MODE, bool NT = false>   // 0: full skeleton; 1: no barriers/exchange (HBM part only); 2: exchange only (no HBM)
__global__ void __launch_bounds__(512) k_proto(const cf* x, cf* out, cf* hbuf, int transforms, Ctl* c, unsigned long long* ticks) {
  extern __shared__ char pad[];
  __shared__ unsigned s_x, s_r, s_ok;
  if (threadIdx.x == 0) {
    s_x = xcc_id() & 7;
    s_r = __hip_atomic_fetch_add(&c->reg_xcd[s_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = 0;
    for (int it = 0; it < 2000000; ++it) { if (__hip_atomic_load(&c->reg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { s_ok = 1; break; } __builtin_amdgcn_s_sleep(4); }
    if (__hip_atomic_load(&c->reg_xcd[s_x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32) s_ok = 0;
    if (!s_ok) atomicAdd(&c->err, 1u);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned xcc = s_x, r = s_r;
  const int t = threadIdx.x, cc = t % 16, u = t / 16;          // COL map for column tiles
  const int row = t / 16, j = t % 16;                          // exchange-read map: 32 rows x 16 lanes
  cf* H = hbuf + (size_t)xcc * (1024 * 512);                   // [1024 rows][512 col'] = 4 MiB
  unsigned bar = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int n = 0; n < transforms; ++n) {
    const size_t tr = (size_t)(n * 8 + xcc);
    const cf* xt = x + tr * 1048576;
    cf* ot = out + tr * 1048576;
    cf a[32], b[32], z1[32], z2[32];
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = NT ? __builtin_nontemporal_load(&xt[(unsigned)((u + 32 * q) * 1024 + 32 * r + cc)]) : xt[(unsigned)((u + 32 * q) * 1024 + 32 * r + cc)];
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) { cf v = {(float)q, (float)t}; a[q] = v; }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) a[q] = a[q] * 1.0001f + a[(q + 1) & 31];
    if (MODE != 1) {
      if (n > 0 && !xcd_barrier(c, xcc, (++bar) * 32)) return;     // B4: everyone finished reading H of the previous transform
#pragma unroll
      for (int q = 0; q < 32; ++q) H[(unsigned)((u + 32 * q) * 512 + 16 * r + cc)] = a[q];
    }
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < 32; ++q) b[q] = NT ? __builtin_nontemporal_load(&xt[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 + cc)]) : xt[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 + cc)];
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) { cf v = {(float)t, (float)q}; b[q] = v; }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) b[q] = b[q] * 1.0001f + b[(q + 1) & 31];
    if (MODE != 1) {
      if (!xcd_barrier(c, xcc, (++bar) * 32)) return;              // B1
#pragma unroll
      for (int q = 0; q < 32; ++q) z1[q] = H[(unsigned)((32 * r + row) * 512 + 16 * q + j)];
      if (!xcd_barrier(c, xcc, (++bar) * 32)) return;              // B2
#pragma unroll
      for (int q = 0; q < 32; ++q) H[(unsigned)((u + 32 * q) * 512 + 16 * r + cc)] = b[q];
      if (!xcd_barrier(c, xcc, (++bar) * 32)) return;              // B3
#pragma unroll
      for (int q = 0; q < 32; ++q) z2[q] = H[(unsigned)((32 * r + row) * 512 + 16 * q + j)];
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) { z1[q] = a[q]; z2[q] = b[q]; }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) { z1[q] = z1[q] + z2[(q + 3) & 31]; z2[q] = z2[q] * 0.5f; }
    if (MODE != 2) {
      // transposed stores: two row tiles of 16 rows, 128-B segments (k1 = 32r + cc [+16]), k2 = u + 32q
#pragma unroll
      for (int q = 0; q < 32; ++q) { if (NT) __builtin_nontemporal_store(z1[q], &ot[(unsigned)((u + 32 * q) * 1024 + 32 * r + cc)]); else ot[(unsigned)((u + 32 * q) * 1024 + 32 * r + cc)] = z1[q]; }
#pragma unroll
      for (int q = 0; q < 32; ++q) { if (NT) __builtin_nontemporal_store(z2[q], &ot[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 + cc)]); else ot[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 + cc)] = z2[q]; }
    } else if (z1[0].x == 12345.678f) ot[t] = z1[1];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// Prototype 2: per-XCD fused two-pass.  Phase A: 2 column tiles per workgroup, x -> W slot (pass-A pattern); XCD barrier;
// phase B: 2 row tiles per workgroup, W slot (rows) -> out (transposed 128-B segments); XCD barrier.  W slot = 8 MiB per XCD,
// reused every transform: whatever stays in the 4 MiB L2 / the Infinity Cache never touches HBM.
template <bool REVERSE_B>
__global__ void __launch_bounds__(512) k_fused2(const cf* x, cf* out, cf* wbuf, int transforms, Ctl* c) {
  extern __shared__ char pad[];
  __shared__ unsigned s_x, s_r, s_ok;
  if (threadIdx.x == 0) {
    s_x = xcc_id() & 7;
    s_r = __hip_atomic_fetch_add(&c->reg_xcd[s_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = 0;
    for (int it = 0; it < 2000000; ++it) { if (__hip_atomic_load(&c->reg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { s_ok = 1; break; } __builtin_amdgcn_s_sleep(4); }
    if (__hip_atomic_load(&c->reg_xcd[s_x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32) s_ok = 0;
    if (!s_ok) atomicAdd(&c->err, 1u);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned xcc = s_x, r = s_r;
  const int t = threadIdx.x, cc = t % 16, u = t / 16;
  const int row = t / 32, j = t % 32;
  cf* W = wbuf + (size_t)xcc * 1048576;
  unsigned bar = 0;
  for (int n = 0; n < transforms; ++n) {
    const size_t tr = (size_t)(n * 8 + xcc);
    const cf* xt = x + tr * 1048576;
    cf* ot = out + tr * 1048576;
    for (int half = 0; half < 2; ++half) {
      cf a[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = xt[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 * half + cc)];
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = a[q] * 1.0001f + a[(q + 1) & 31];
#pragma unroll
      for (int q = 0; q < 32; ++q) W[(unsigned)((u + 32 * q) * 1024 + 32 * r + 16 * half + cc)] = a[q];
    }
    if (!xcd_barrier(c, xcc, (++bar) * 32)) return;
    for (int half = 0; half < 2; ++half) {
      const unsigned rr = REVERSE_B ? (31 - r) : r;      // which row tiles this workgroup takes in phase B
      cf b[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) b[q] = W[(unsigned)((32 * rr + 16 * half + row) * 1024 + j + 32 * q)];
#pragma unroll
      for (int q = 0; q < 32; ++q) b[q] = b[q] * 0.5f + b[(q + 3) & 31];
#pragma unroll
      for (int q = 0; q < 32; ++q) ot[(unsigned)((u + 32 * q) * 1024 + 32 * rr + 16 * half + cc)] = b[q];
    }
    if (!xcd_barrier(c, xcc, (++bar) * 32)) return;
  }
}

int main() {
  const unsigned LDS = 100 * 1024;
  const int TR = 64;                       // transforms per XCD
  const size_t n = (size_t)TR * 8 * 1048576;
  cf *x, *out, *h; Ctl* c; unsigned long long* ticks;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&h, (size_t)8 * 1024 * 512 * 8)); CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&ticks, 256 * 8));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(out, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int mode, const char* name) {
    float best = 1e30f; unsigned err = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(c, 0, sizeof(Ctl)));
      CK(hipEventRecord(e0));
      if (mode == 0) { CK(hipFuncSetAttribute((const void*)k_proto<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL(k_proto<0>, dim3(256), dim3(512), LDS, 0, x, out, h, TR, c, ticks); }
      if (mode == 1) { CK(hipFuncSetAttribute((const void*)k_proto<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL(k_proto<1>, dim3(256), dim3(512), LDS, 0, x, out, h, TR, c, ticks); }
      if (mode == 3) { CK(hipFuncSetAttribute((const void*)k_proto<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL((k_proto<0, true>), dim3(256), dim3(512), LDS, 0, x, out, h, TR, c, ticks); }
      if (mode == 4) { CK(hipFuncSetAttribute((const void*)k_proto<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL((k_proto<1, true>), dim3(256), dim3(512), LDS, 0, x, out, h, TR, c, ticks); }
      if (mode == 2) { CK(hipFuncSetAttribute((const void*)k_proto<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL(k_proto<2>, dim3(256), dim3(512), LDS, 0, x, out, h, TR, c, ticks); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      Ctl hc; CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost)); err += hc.err;
    }
    const double per_tr = best * 1e3 / (TR * 8);
    printf("%-46s %8.3f ms for %d transforms: %6.3f us/transform -> %6.1f GPoints/s (N=2^20)  err=%u\n", name, best, TR * 8, per_tr, 1048576.0 / per_tr / 1e3, err);
  };
  run(1, "HBM part only (16 B/pt, no exchange)");
  run(2, "exchange only (2 rounds through L2, 4 barriers)");
  run(0, "full skeleton");
  {
    cf* w; CK(hipMalloc(&w, (size_t)8 * 1048576 * 8));
    for (int rev = 0; rev < 2; ++rev) {
      float best = 1e30f; unsigned err = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(c, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        if (rev) { CK(hipFuncSetAttribute((const void*)k_fused2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL(k_fused2<true>, dim3(256), dim3(512), LDS, 0, x, out, w, TR, c); }
        else { CK(hipFuncSetAttribute((const void*)k_fused2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); hipLaunchKernelGGL(k_fused2<false>, dim3(256), dim3(512), LDS, 0, x, out, w, TR, c); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        Ctl hc; CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost)); err += hc.err;
      }
      const double per_tr = best * 1e3 / (TR * 8);
      printf("%-46s %8.3f ms for %d transforms: %6.3f us/transform -> %6.1f GPoints/s (N=2^20)  err=%u\n", rev ? "fused two-pass per XCD via W slot (B reversed)" : "fused two-pass per XCD via 8 MiB W slot", best, TR * 8, per_tr,
             1048576.0 / per_tr / 1e3, err);
    }
  }
  run(4, "HBM part only, nontemporal loads/stores");
  run(3, "full skeleton, nontemporal HBM loads/stores");
  return 0;
}
