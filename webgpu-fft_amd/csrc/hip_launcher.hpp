// hip_launcher.hpp — the product Launcher for dispatch.hpp: enqueues kernels on a HIP stream (the device
// queue, or a capturing stream while a command list is being turned into a hipGraph).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <map>
#include <utility>

#include "dispatch.hpp"

namespace mi355 {

struct HipLauncher {
  hipStream_t stream = nullptr;
  bool prepare_only = false;   // only raise dynamic-LDS limits (must happen outside stream capture)
  hipError_t status = hipSuccess;
  unsigned* sticky = nullptr;  // device-wide error word raised by kernels with bounded waits (kern_xcd.hpp)
  unsigned* sticky_error_word() const { return sticky; }
  // occupancy_query: launch_concurrent() does not launch but asks the runtime how many workgroups of that kernel (block size,
  // dynamic LDS) one CU holds, and keeps the smallest answer: the planner's co-residency assumption is checked at plan creation
  bool occupancy_query = false;
  int min_blocks_per_cu = 1 << 30;
  // kernels whose workgroups synchronise with each other: same launch on hardware (the grid is sized so that every
  // workgroup is resident: one per CU); the emulation needs its blocks to run concurrently
  template <class... P, class... A>
  void launch_concurrent(void (*kernel)(P...), unsigned grid, unsigned block, unsigned smem, A&&... args) {
    if (occupancy_query) {
      if (status != hipSuccess) return;
      if (smem > 48 * 1024) raise_lds_limit(reinterpret_cast<const void*>(kernel), smem, status);
      if (status != hipSuccess) return;
      int nb = 0;
      const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, (int)block, (size_t)smem);
      if (e != hipSuccess) { status = e; return; }
      if (nb < min_blocks_per_cu) min_blocks_per_cu = nb;
      return;
    }
    launch(kernel, grid, block, smem, static_cast<A&&>(args)...);
  }

  static void raise_lds_limit(const void* fn, unsigned smem, hipError_t& status) {
    static std::mutex mu;
    // per (device, kernel): the attribute belongs to the device's copy of the code object; kernels with a runtime LDS size
    // (kern_mixed.hpp) may ask for more later
    static std::map<std::pair<int, const void*>, unsigned> raised;
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, fn);
    const auto it = raised.find(key);
    if (it != raised.end() && it->second >= smem) return;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { status = e; return; }
    raised[key] = smem;
  }

  template <class... P, class... A>
  void launch(void (*kernel)(P...), unsigned grid, unsigned block, unsigned smem, A&&... args) {
    if (status != hipSuccess) return;
    if (smem > 48 * 1024) raise_lds_limit(reinterpret_cast<const void*>(kernel), smem, status);
    if (prepare_only || status != hipSuccess) return;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), smem, stream, static_cast<P>(args)...);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) status = e;
  }
  void copy(void* dst, const void* src, size_t bytes) {
    if (prepare_only || status != hipSuccess) return;
    // large aligned, non-overlapping ranges: the one-shot streaming kernel (kern_generic.hpp); anything else: the runtime's copy
    const uintptr_t d = reinterpret_cast<uintptr_t>(dst), s = reinterpret_cast<uintptr_t>(src);
    if (bytes >= (1u << 20) && bytes % 16 == 0 && d % 16 == 0 && s % 16 == 0 && (d + bytes <= s || s + bytes <= d)) {
      const unsigned long long n16 = bytes / 16, slabs = (n16 + 1023) / 1024;
      launch(stream_copy_kernel, (unsigned)(slabs < 0x3fffffull ? slabs : 0x3fffffull), 256u, 0u, (const f4v*)src, (f4v*)dst, n16);
      return;
    }
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) status = e;
  }
};

// per-family instantiations live in lines_fam*.hip so that device code compiles in parallel
extern template bool launch_lines_family<FAM_ROW_SMALL, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);
extern template bool launch_lines_family<FAM_ROW_1K, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);
extern template bool launch_lines_family<FAM_ROW_BIG, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);
extern template bool launch_lines_family<FAM_PASS_A, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);
extern template bool launch_lines_family<FAM_PASS_B, HipLauncher>(int, const LineArgs&, unsigned, HipLauncher&);

extern template bool launch_mixedct<HipLauncher>(int, const MixedArgs&, unsigned, HipLauncher&);
extern template bool launch_line_reg<HipLauncher>(int, const MixedArgs&, unsigned, HipLauncher&);
extern template bool launch_xcd_res<HipLauncher>(int, const XcdFusedArgs&, unsigned, HipLauncher&);

}  // namespace mi355
