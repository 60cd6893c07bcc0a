// MALL / HBM behaviour probes (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int U> __global__ void __launch_bounds__(256) read_k(const f4* in, f4* sink, size_t n) {
  f4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < n; i += stride) { f4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x == 1234.5f) sink[0] = acc;
}
template <int U> __global__ void __launch_bounds__(256) write_k(f4* out, size_t n, float s) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4 v = {s, s + 1, s + 2, s + 3};
  for (; i < n; i += stride) out[i] = v;
}
template <int U, bool NT> __global__ void __launch_bounds__(256) copy_k(const f4* in, f4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) v[u] = __builtin_nontemporal_load(&in[i + u * stride]); else v[u] = in[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
// block-contiguous copy: each block owns a contiguous slab (like tiles), U f4 per thread per iteration
template <int U> __global__ void __launch_bounds__(256) copy_slab(const f4* in, f4* out, size_t n) {
  const size_t per_iter = 256 * U;
  for (size_t base = (size_t)blockIdx.x * per_iter; base < n; base += (size_t)gridDim.x * per_iter) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + u * 256 + threadIdx.x] = v[u];
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> float run(F&& f, int reps = 5) {
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; }
    return best;
  }
};

int main() {
  const size_t big = (size_t)4 << 30;
  f4 *a, *b, *sink;
  CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, big)); CK(hipMemset(b, 0, big));
  Timer tm;
  printf("--- copy variants over 2 GiB in + 2 GiB out\n");
  const size_t n2g = ((size_t)2 << 30) / 16;
#define COPY(U, NT, G) { float ms = tm.run([&] { hipLaunchKernelGGL((copy_k<U, NT>), dim3(G), dim3(256), 0, 0, a, b, n2g); }); printf("copy_k U=%d NT=%d grid=%5d  %8.3f ms %7.1f GB/s\n", U, (int)NT, (int)(G), ms, 2.0 * n2g * 16 / ms / 1e6); }
  COPY(1, false, 2048) COPY(4, false, 2048) COPY(4, false, 4096) COPY(8, false, 2048) COPY(4, false, 1024) COPY(4, true, 2048) COPY(8, true, 2048) COPY(4, false, 8192) COPY(4, false, 16384)
#define SLAB(U, G) { float ms = tm.run([&] { hipLaunchKernelGGL((copy_slab<U>), dim3(G), dim3(256), 0, 0, a, b, n2g); }); printf("copy_slab U=%d grid=%5d      %8.3f ms %7.1f GB/s\n", U, (int)(G), ms, 2.0 * n2g * 16 / ms / 1e6); }
  SLAB(4, 2048) SLAB(8, 2048) SLAB(8, 1024) SLAB(16, 1024) SLAB(8, 4096)
  { float ms = tm.run([&] { CK(hipMemcpyAsync(b, a, (size_t)2 << 30, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpyAsync D2D 2 GiB       %8.3f ms %7.1f GB/s\n", ms, 2.0 * ((size_t)2 << 30) / ms / 1e6); }
  printf("--- read-only / write-only over 2 GiB\n");
  { float ms = tm.run([&] { hipLaunchKernelGGL((read_k<4>), dim3(2048), dim3(256), 0, 0, a, sink, n2g); }); printf("read_k U=4           %8.3f ms %7.1f GB/s\n", ms, n2g * 16.0 / ms / 1e6); }
  { float ms = tm.run([&] { hipLaunchKernelGGL((read_k<8>), dim3(4096), dim3(256), 0, 0, a, sink, n2g); }); printf("read_k U=8 g4096     %8.3f ms %7.1f GB/s\n", ms, n2g * 16.0 / ms / 1e6); }
  { float ms = tm.run([&] { hipLaunchKernelGGL((write_k<1>), dim3(2048), dim3(256), 0, 0, b, n2g, 1.0f); }); printf("write_k              %8.3f ms %7.1f GB/s\n", ms, n2g * 16.0 / ms / 1e6); }
  printf("--- repeated read of a buffer of size S (10 passes inside one timing): MALL read hit rate\n");
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
    const size_t n = (mb << 20) / 16;
    float ms = tm.run([&] { for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((read_k<4>), dim3(2048), dim3(256), 0, 0, a, sink, n); });
    printf("read x10 S=%5zu MiB   %8.3f ms  %7.1f GB/s\n", mb, ms, 10.0 * n * 16 / ms / 1e6);
  }
  printf("--- write S then read S (same buffer), alternating 10 times: does a read after a write hit?\n");
  for (size_t mb : {16, 32, 64, 128, 256, 512}) {
    const size_t n = (mb << 20) / 16;
    float msw = tm.run([&] { for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((write_k<1>), dim3(2048), dim3(256), 0, 0, b, n, (float)r); });
    float msrw = tm.run([&] { for (int r = 0; r < 10; ++r) { hipLaunchKernelGGL((write_k<1>), dim3(2048), dim3(256), 0, 0, b, n, (float)r); hipLaunchKernelGGL((read_k<4>), dim3(2048), dim3(256), 0, 0, b, sink, n); } });
    printf("S=%4zu MiB: write x10 %8.3f ms (%7.1f GB/s)   write+read x10 %8.3f ms -> read part %7.1f GB/s\n", mb, msw, 10.0 * n * 16 / msw / 1e6, msrw,
           10.0 * n * 16 / (msrw - msw) / 1e6);
  }
  printf("--- chunked pipeline model: copy X(HBM, streaming 4 GiB) -> W(S) then W(S) -> Y(HBM), chunk S\n");
  for (size_t mb : {16, 32, 64, 128, 256}) {
    const size_t n = (mb << 20) / 16;
    const size_t chunks = ((size_t)2 << 30) / (mb << 20);
    f4* w = sink;  // placeholder
    CK(hipMalloc(&w, mb << 20));
    float ms = tm.run([&] { for (size_t c = 0; c < chunks; ++c) { hipLaunchKernelGGL((copy_k<4, false>), dim3(2048), dim3(256), 0, 0, a + c * n, w, n); hipLaunchKernelGGL((copy_k<4, false>), dim3(2048), dim3(256), 0, 0, w, b + c * n, n); } }, 3);
    printf("S=%4zu MiB chunks=%4zu: %8.3f ms  -> %7.1f GB/s algorithmic (2 GiB in + 2 GiB out)\n", mb, chunks, ms, 2.0 * ((size_t)2 << 30) / ms / 1e6);
    CK(hipFree(w));
  }
  return 0;
}
