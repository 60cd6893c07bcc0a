// microbench: times individual kernels / access patterns on one MI355X (development tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kern_lines.hpp"
#include "plan.hpp"
using namespace mi355;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void copy_f4(const float4* in, float4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// tile copy with pass A's access pattern: tile = [1024 rows][T cols] of a [b][1024][1024] array; 512 threads
template <int T, int ROWS_PER_THREAD>
__global__ void __launch_bounds__(512) copy_tile(const cf* in, cf* out, long long tiles) {
  const int t = threadIdx.x;
  const int c = t % T, u = t / T;           // u in [0, 512/T)
  constexpr int UCNT = 512 / T;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long o = tile / (1024 / T), i0 = (tile % (1024 / T)) * T;
    const cf* p = in + o * 1048576 + i0;
    cf* q = out + o * 1048576 + i0;
    cf v[ROWS_PER_THREAD];
#pragma unroll
    for (int r = 0; r < ROWS_PER_THREAD; ++r) v[r] = p[(unsigned)((u + r * UCNT) * 1024 + c)];
#pragma unroll
    for (int r = 0; r < ROWS_PER_THREAD; ++r) q[(unsigned)((u + r * UCNT) * 1024 + c)] = v[r];
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> float run(F&& f, int reps = 5) {
    f();  // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
    }
    last_avg = sum / reps;
    return best;
  }
  float last_avg = 0;
};

static std::vector<float2h> line_tables(int R0, int R1, int R2, int N) {
  std::vector<float2h> t;
  if (R1 > 1) for (int q = 1; q < R1; ++q) for (int k = 0; k < R0; ++k) t.push_back(root_of_unity((long long)q * k, (long long)R0 * R1));
  if (R2 > 1) for (int q = 1; q < R2; ++q) for (int k = 0; k < R0 * R1; ++k) t.push_back(root_of_unity((long long)q * k, N));
  if (t.empty()) t.push_back(float2h{1, 0});
  return t;
}

template <class C> void run_lines(const char* name, Timer& tm, LineArgs a, long long tiles, unsigned grid, double bytes) {
  a.num_tiles = tiles; a.num_lines = tiles * C::T;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_lines_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  float ms = tm.run([&] { hipLaunchKernelGGL(fft_lines_kernel<C>, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, 0, a); });
  CK(hipGetLastError());
  printf("%-44s grid=%5u thr=%4d lds=%6d  best %8.3f ms avg %8.3f  %7.1f GB/s\n", name, grid, C::THREADS, C::LDS_BYTES, ms, tm.last_avg, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
  const long long B = argc > 1 ? atoll(argv[1]) : 256;   // transforms of 2^20
  const long long N = 1048576;
  const size_t bytes = (size_t)B * N * 8;
  cf *in, *out;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
  CK(hipMemset(in, 0, bytes)); CK(hipMemset(out, 0, bytes));
  {  // non-trivial data
    std::vector<float> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    for (size_t off = 0; off + h.size() * 4 <= bytes; off += h.size() * 4) CK(hipMemcpy((char*)in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  Timer tm;
  const double rw = 2.0 * bytes;
  int cus = 256;
  {
    float ms = tm.run([&] { hipLaunchKernelGGL(copy_f4, dim3(cus * 8), dim3(256), 0, 0, (const float4*)in, (float4*)out, bytes / 16); });
    printf("%-44s best %8.3f ms  %7.1f GB/s\n", "copy_f4 (grid-stride, 2048 blocks)", ms, rw / ms / 1e6);
  }
#define TILE(T, R, G) { long long tiles = B * (1024 / T); float ms = tm.run([&] { hipLaunchKernelGGL((copy_tile<T, R>), dim3(G), dim3(512), 0, 0, in, out, tiles); }); \
    printf("copy_tile T=%-3d rows/thr=%-3d grid=%-5d             best %8.3f ms  %7.1f GB/s\n", T, R, (int)(G), ms, rw / ms / 1e6); }
  TILE(16, 32, 256) TILE(16, 32, 512) TILE(16, 32, 1024) TILE(16, 32, 2048)
  TILE(32, 64, 256) TILE(32, 64, 512) TILE(32, 64, 1024)
  TILE(64, 128, 256) TILE(64, 128, 512)
  TILE(8, 16, 512) TILE(8, 16, 2048)

  // twiddle tables
  auto up = [&](const std::vector<float2h>& v) { cf* d; CK(hipMalloc(&d, v.size() * 8)); CK(hipMemcpy(d, v.data(), v.size() * 8, hipMemcpyHostToDevice)); return d; };
  std::vector<float2h> lo(1024), hi(1024);
  for (int l = 0; l < 1024; ++l) { lo[l] = root_of_unity(l, N); hi[l] = root_of_unity((long long)l << 10, N); }
  cf* tlo = up(lo); cf* thi = up(hi);
  cf* t3232 = up(line_tables(32, 32, 1, 1024));
  cf* t16164 = up(line_tables(16, 16, 4, 1024));
  LineArgs a{};
  a.in = in; a.out = out; a.tw = t3232; a.tw_lo = tlo; a.tw_hi = thi; a.scale = 1.0f; a.fs_shift = 10; a.fs_lo_mask = 1023;
  // pass A: COL/COL, S = 1024, outer = N
  a.in_S = 1024; a.in_outer_stride = N; a.out_S = 1024; a.out_outer_stride = N;
  run_lines<LineCfg<1024, 32, 32, 1, 16, true, true, false, false, 1>>("passA 32x32 T16 twid", tm, a, B * 64, 256, rw);
  run_lines<LineCfg<1024, 32, 32, 1, 16, true, true, false, false, 0>>("passA 32x32 T16 NO twid", tm, a, B * 64, 256, rw);
  a.tw = t16164;
  run_lines<LineCfg<1024, 16, 16, 4, 16, true, true, false, false, 1>>("passA 16x16x4 T16 twid (1024 thr)", tm, a, B * 64, 256, rw);
  run_lines<LineCfg<1024, 16, 16, 4, 16, true, true, false, false, 0>>("passA 16x16x4 T16 NO twid", tm, a, B * 64, 256, rw);
  // pass B: ROW in (row stride 1024), COL out S = 1024
  a.tw = t3232;
  a.in_S = 1; a.in_outer_stride = 1024; a.out_S = 1024; a.out_outer_stride = N;
  a.fs_group = 1024;
  run_lines<LineCfg<1024, 32, 32, 1, 16, false, true, false, false, 0>>("passB 32x32 T16 NO twid", tm, a, B * 64, 256, rw);
  run_lines<LineCfg<1024, 32, 32, 1, 16, false, true, false, false, 2>>("passB 32x32 T16 twid-in hoisted", tm, a, B * 64, 256, rw);
  run_lines<LineCfg<1024, 32, 32, 1, 16, false, true, false, false, 2>>("passB 32x32 T16 twid-in per-tile (grid 250)", tm, a, B * 64, 250, rw);
  a.tw = t16164;
  run_lines<LineCfg<1024, 16, 16, 4, 16, false, true, false, false, 0>>("passB 16x16x4 T16 (1024 thr)", tm, a, B * 64, 256, rw);
  // ROW/ROW
  a.tw = t3232;
  a.in_S = 1; a.in_outer_stride = 1024; a.out_S = 1; a.out_outer_stride = 1024;
  run_lines<LineCfg<1024, 32, 32, 1, 4, false, false, false, false, 0>>("row 32x32 T4 (128 thr)", tm, a, B * 1024 / 4, 256 * 4, rw);
  run_lines<LineCfg<1024, 32, 32, 1, 8, false, false, false, false, 0>>("row 32x32 T8 (256 thr)", tm, a, B * 1024 / 8, 256 * 2, rw);
  run_lines<LineCfg<1024, 32, 32, 1, 2, false, false, false, false, 0>>("row 32x32 T2 (64 thr)", tm, a, B * 1024 / 2, 256 * 8, rw);
  a.tw = t16164;
  run_lines<LineCfg<1024, 16, 16, 4, 4, false, false, false, false, 0>>("row 16x16x4 T4 (256 thr)", tm, a, B * 1024 / 4, 256 * 4, rw);
  run_lines<LineCfg<1024, 16, 16, 4, 2, false, false, false, false, 0>>("row 16x16x4 T2 (128 thr)", tm, a, B * 1024 / 2, 256 * 8, rw);
  run_lines<LineCfg<1024, 16, 16, 4, 1, false, false, false, false, 0>>("row 16x16x4 T1 (64 thr)", tm, a, B * 1024, 256 * 16, rw);
  return 0;
}
