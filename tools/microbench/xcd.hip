// XCD-level feasibility probes for the "one transform resident in one XCD" design (development tool).
//  1. block -> XCC_ID placement   2. HBM bandwidth with K of 8 XCDs streaming
//  3. cost of a 32-workgroup same-XCD barrier   4. same-XCD exchange through L2: bandwidth and staleness
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 0xf; }

struct Ctl {
  unsigned reg_total;        // registration counter (all blocks)
  unsigned reg_xcd[8];       // blocks registered per XCD
  unsigned bar[8][16];       // per-XCD monotonic barrier counters (padded)
  unsigned err;              // timeouts / stale words
  unsigned long long stale;
};

// registration: returns rank within XCD; waits (bounded) until every block has registered
__device__ bool register_block(Ctl* c, unsigned& xcc, unsigned& rank, unsigned& gsize) {
  __shared__ unsigned s_x, s_r, s_g, s_ok;
  if (threadIdx.x == 0) {
    s_x = xcc_id() & 7;
    s_r = __hip_atomic_fetch_add(&c->reg_xcd[s_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = 0;
    for (int it = 0; it < 2000000; ++it) {
      if (__hip_atomic_load(&c->reg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { s_ok = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    s_g = __hip_atomic_load(&c->reg_xcd[s_x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!s_ok) atomicAdd(&c->err, 1u);
  }
  __syncthreads();
  xcc = s_x; rank = s_r; gsize = s_g;
  return s_ok != 0;
}

// same-XCD barrier over `gsize` workgroups.  RELEASE: also write back L2 (agent release) — needed only cross-XCD.
template <bool RELEASE>
__device__ bool xcd_barrier(Ctl* c, unsigned xcc, unsigned target) {
  __shared__ unsigned s_ok2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (RELEASE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&c->bar[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned ok = 0;
    for (int it = 0; it < 4000000; ++it) {
      if (__hip_atomic_load(&c->bar[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: drop this CU's L1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!ok) atomicAdd(&c->err, 1u);
    s_ok2 = ok;
  }
  __syncthreads();
  return s_ok2 != 0;
}

__global__ void __launch_bounds__(256) k_placement(unsigned* out) {
  extern __shared__ char pad[];
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// streaming copy done only by blocks whose XCC id < active
__global__ void __launch_bounds__(1024) k_stream(const f4* in, f4* out, size_t n_per_xcd, unsigned active, Ctl* c) {
  extern __shared__ char pad[];
  unsigned xcc, rank, gsize;
  if (!register_block(c, xcc, rank, gsize)) return;
  if (xcc >= active) return;
  const f4* src = in + (size_t)xcc * n_per_xcd;
  f4* dst = out + (size_t)xcc * n_per_xcd;
  for (size_t i = (size_t)rank * 1024 + threadIdx.x; i + 3 * (size_t)gsize * 1024 < n_per_xcd; i += 4 * (size_t)gsize * 1024) {
    f4 a = src[i], b = src[i + (size_t)gsize * 1024], d = src[i + 2 * (size_t)gsize * 1024], e = src[i + 3 * (size_t)gsize * 1024];
    dst[i] = a; dst[i + (size_t)gsize * 1024] = b; dst[i + 2 * (size_t)gsize * 1024] = d; dst[i + 3 * (size_t)gsize * 1024] = e;
  }
}

template <bool RELEASE>
__global__ void __launch_bounds__(1024) k_barrier(Ctl* c, int iters, unsigned long long* cycles) {
  extern __shared__ char pad[];
  unsigned xcc, rank, gsize;
  if (!register_block(c, xcc, rank, gsize)) return;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
    if (!xcd_barrier<RELEASE>(c, xcc, (unsigned)(it + 1) * gsize)) return;
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;   // 100 MHz ticks
}

// exchange: round r: block (xcc, rank) writes its slab (slab_f2 float2, 8-B stores like the FFT kernels) tagged with
// (round, rank, idx); barrier; reads the slab of rank+1 and checks every word; barrier.
template <bool RELEASE>
__global__ void __launch_bounds__(1024) k_exchange(f2* w, size_t slab_f2, int rounds, Ctl* c, unsigned long long* cycles) {
  extern __shared__ char pad[];
  unsigned xcc, rank, gsize;
  if (!register_block(c, xcc, rank, gsize)) return;
  f2* base = w + (size_t)xcc * 32 * slab_f2;
  unsigned bar = 0;
  unsigned long long stale = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; ++r) {
    f2* mine = base + (size_t)rank * slab_f2;
    for (size_t i = threadIdx.x; i < slab_f2; i += 1024) { f2 v = {(float)(r * 64 + rank), (float)i}; mine[i] = v; }
    if (!xcd_barrier<RELEASE>(c, xcc, (++bar) * gsize)) return;
    const f2* other = base + (size_t)((rank + 1) % gsize) * slab_f2;
    const float want = (float)(r * 64 + (rank + 1) % gsize);
    for (size_t i = threadIdx.x; i < slab_f2; i += 1024) { f2 v = other[i]; if (v.x != want || v.y != (float)i) ++stale; }
    if (!xcd_barrier<false>(c, xcc, (++bar) * gsize)) return;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (stale) atomicAdd(&c->stale, stale);
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const unsigned LDS = 100 * 1024;   // forces one workgroup per CU
  Ctl* c; CK(hipMalloc(&c, sizeof(Ctl)));
  unsigned* place; CK(hipMalloc(&place, 4096 * 4));
  CK(hipFuncSetAttribute((const void*)k_placement, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL(k_placement, dim3(256), dim3(256), LDS, 0, place);
  std::vector<unsigned> hp(256); CK(hipMemcpy(hp.data(), place, 256 * 4, hipMemcpyDeviceToHost));
  int counts[16] = {0}, rr = 0;
  for (int b = 0; b < 256; ++b) { counts[hp[b] & 15]++; if ((hp[b] & 7) == ((hp[0] + b) & 7)) rr++; }
  printf("placement: per-XCC block counts:"); for (int x = 0; x < 8; ++x) printf(" %d", counts[x]);
  printf("   round-robin-consistent blocks: %d/256 (block0 on XCC %u)\n", rr, hp[0]);

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t per_xcd = ((size_t)512 << 20) / 16;     // 512 MiB per XCD region
  f4 *in, *out; CK(hipMalloc(&in, per_xcd * 16 * 8)); CK(hipMalloc(&out, per_xcd * 16 * 8));
  CK(hipMemset(in, 1, per_xcd * 16 * 8)); CK(hipMemset(out, 0, per_xcd * 16 * 8));
  CK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  for (unsigned active : {1u, 2u, 4u, 8u}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(c, 0, sizeof(Ctl)));
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_stream, dim3(256), dim3(1024), LDS, 0, in, out, per_xcd, active, c); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("stream copy with %u XCD(s) active (32 WGs x 1024 thr each): %7.3f ms  %7.1f GB/s total, %7.1f GB/s per XCD\n", active, best,
           2.0 * per_xcd * 16 * active / best / 1e6, 2.0 * per_xcd * 16 / best / 1e6);
  }
  unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
  std::vector<unsigned long long> hc(256);
  Ctl hctl;
  auto report = [&](const char* name, double per_iter_div) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hctl, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    unsigned long long mx = 0; for (auto v : hc) mx = v > mx ? v : mx;
    printf("%-52s max %9.3f us per iteration   err=%u stale=%llu\n", name, mx / 100.0 / per_iter_div, hctl.err, hctl.stale);
  };
  CK(hipFuncSetAttribute((const void*)k_barrier<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_barrier<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(cyc, 0, 256 * 8));
  hipLaunchKernelGGL(k_barrier<false>, dim3(256), dim3(1024), LDS, 0, c, 2000, cyc); report("xcd barrier (32 WG, acquire only)", 2000);
  CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(cyc, 0, 256 * 8));
  hipLaunchKernelGGL(k_barrier<true>, dim3(256), dim3(1024), LDS, 0, c, 2000, cyc); report("xcd barrier (32 WG, release + acquire)", 2000);

  CK(hipFuncSetAttribute((const void*)k_exchange<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_exchange<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  for (size_t kb : {16, 64, 128}) {
    const size_t slab = kb * 1024 / 8;
    f2* w; CK(hipMalloc(&w, slab * 8 * 32 * 8));
    for (int rel = 0; rel < 2; ++rel) {
      CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(cyc, 0, 256 * 8));
      const int rounds = 200;
      if (rel) hipLaunchKernelGGL(k_exchange<true>, dim3(256), dim3(1024), LDS, 0, w, slab, rounds, c, cyc);
      else hipLaunchKernelGGL(k_exchange<false>, dim3(256), dim3(1024), LDS, 0, w, slab, rounds, c, cyc);
      char name[128]; snprintf(name, sizeof name, "exchange %zu KB/WG via L2, %s", kb, rel ? "release+acquire" : "acquire only (same-XCD)");
      report(name, rounds);
    }
    CK(hipFree(w));
  }
  return 0;
}
