// bw.hip — how fast can this chip stream?  Copy / read / write kernels over 4 GiB buffers with different access widths,
// unroll depths, grid sizes and cache policies (development tool; results in profiles/).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <class V, int U, bool NT> __global__ void __launch_bounds__(256) k_copy(const V* __restrict__ in, V* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    V v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) v[j] = NT ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
    for (int j = 0; j < U; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) { if (NT) __builtin_nontemporal_store(v[j], out + i); else out[i] = v[j]; } }
  }
}
template <class V, int U> __global__ void __launch_bounds__(256) k_read(const V* __restrict__ in, float* sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  float acc = 0;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
#pragma unroll
    for (int j = 0; j < U; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) acc += in[i].x; }
  }
  if (acc == 123.456f) *sink = acc;
}
template <class V, int U> __global__ void __launch_bounds__(256) k_write(V* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  V z; z = 1.0f;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
#pragma unroll
    for (int j = 0; j < U; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) out[i] = z; }
  }
}
template <class F> float timeit(F&& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  void *in, *out; float* sink;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  printf("hipMemcpyDtoD 4 GiB: %.1f GB/s (read+write)\n", 2.0 * bytes / timeit([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }) / 1e6);
  for (unsigned grid : {256u, 512u, 1024u, 2048u, 4096u, 16384u}) {
#define RUN(V, U, NT) { const size_t n = bytes / sizeof(V); const float ms = timeit([&] { hipLaunchKernelGGL((k_copy<V, U, NT>), dim3(grid), dim3(256), 0, 0, (const V*)in, (V*)out, n); }); \
    printf("copy  %-3s U=%d NT=%d grid=%-6u %8.1f GB/s\n", #V, U, (int)NT, grid, 2.0 * bytes / ms / 1e6); }
    RUN(f2, 4, false) RUN(f4, 1, false) RUN(f4, 2, false) RUN(f4, 4, false) RUN(f4, 8, false) RUN(f4, 4, true) RUN(f4, 8, true)
#undef RUN
  }
  for (unsigned grid : {512u, 2048u, 16384u}) {
    { const size_t n = bytes / 16; const float ms = timeit([&] { hipLaunchKernelGGL((k_read<f4, 8>), dim3(grid), dim3(256), 0, 0, (const f4*)in, sink, n); }); printf("read  f4 U=8 grid=%-6u %8.1f GB/s\n", grid, bytes / ms / 1e6); }
    { const size_t n = bytes / 16; const float ms = timeit([&] { hipLaunchKernelGGL((k_write<f4, 8>), dim3(grid), dim3(256), 0, 0, (f4*)out, n); }); printf("write f4 U=8 grid=%-6u %8.1f GB/s\n", grid, bytes / ms / 1e6); }
  }
  return 0;
}
