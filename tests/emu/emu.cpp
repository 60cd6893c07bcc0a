// emu.cpp — host emulation of the HIP kernels for the GPU-less CPU test tier (tests/test_emu_*.py).
//
// TEST INFRASTRUCTURE ONLY.  Compiles webgpu-fft_amd/csrc/kern_*.hpp with -DMI355_HOST_EMU (see
// csrc/platform.hpp) and runs a planned transform on host memory: one std::thread per GPU thread, a
// pthread barrier for __syncthreads().  It exists so kernel index math / LDS layouts / barrier placement
// and the planner can be checked against the oracle (and under ASan/UBSan) where there is no GPU.
// Nothing here is linked into libmi355fft.so; the product has no CPU fallback.
#include <pthread.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "dispatch.hpp"

namespace emu {
thread_local dim3_t t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local char* t_smem = nullptr;
unsigned g_xcds = 2;
static thread_local pthread_barrier_t* t_barrier = nullptr;
static thread_local pthread_barrier_t* t_wave_barrier = nullptr;
void sync_threads() { pthread_barrier_wait(t_barrier); }
void sync_wave() { pthread_barrier_wait(t_wave_barrier); }
void yield_thread() { std::this_thread::yield(); }
}  // namespace emu

namespace {

struct EmuLauncher {
  // every kernel is a grid-stride loop: a few blocks exercise the stride path (MI355_EMU_MAX_GRID overrides,
  // e.g. to make grid*T a multiple of the four-step group so the hoisted-roots path runs)
  unsigned max_grid = std::getenv("MI355_EMU_MAX_GRID") ? (unsigned)std::atoi(std::getenv("MI355_EMU_MAX_GRID")) : 3;
  template <class... P, class... A>
  void launch(void (*kernel)(P...), unsigned grid, unsigned block, unsigned smem, A&&... args) {
    const unsigned g = std::min(grid, max_grid);
    std::vector<char> shared(std::max<size_t>(smem, 64 * 1024) + 64);
    char* smem_base = shared.data() + (64 - (reinterpret_cast<uintptr_t>(shared.data()) & 63)) % 64;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, block);
    const unsigned waves = (block + 63) / 64;
    std::vector<pthread_barrier_t> wbar(waves);
    for (unsigned w = 0; w < waves; ++w) pthread_barrier_init(&wbar[w], nullptr, std::min(64u, block - 64 * w));
    std::vector<std::thread> th;
    th.reserve(block);
    auto body = [&](unsigned tid) {
      emu::t_barrier = &bar;
      emu::t_wave_barrier = &wbar[tid / 64];
      emu::t_smem = smem_base;
      emu::t_blockDim.x = block;
      emu::t_gridDim.x = g;
      emu::t_threadIdx.x = tid;
      for (unsigned b = 0; b < g; ++b) {
        emu::t_blockIdx.x = b;
        kernel(args...);
        pthread_barrier_wait(&bar);  // next block reuses the shared memory
      }
    };
    for (unsigned t = 0; t < block; ++t) th.emplace_back(body, t);
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&bar);
    for (auto& w : wbar) pthread_barrier_destroy(&w);
  }
  void copy(void* dst, const void* src, size_t bytes) { std::memmove(dst, src, bytes); }

  unsigned sticky = 0;
  unsigned* sticky_error_word() { return &sticky; }
  // kernels whose workgroups synchronise with each other (kern_xcd.hpp): every block gets its own threads,
  // barrier and shared memory, and all blocks run at the same time
  template <class... P, class... A>
  void launch_concurrent(void (*kernel)(P...), unsigned grid, unsigned block, unsigned smem, A&&... args) {
    const size_t bytes = std::max<size_t>(smem, 64 * 1024) + 64;
    std::vector<std::vector<char>> shared(grid, std::vector<char>(bytes));
    std::vector<pthread_barrier_t> bars(grid);
    for (auto& b : bars) pthread_barrier_init(&b, nullptr, block);
    const unsigned waves = (block + 63) / 64;
    std::vector<pthread_barrier_t> wbars((size_t)grid * waves);
    for (unsigned b = 0; b < grid; ++b)
      for (unsigned w = 0; w < waves; ++w) pthread_barrier_init(&wbars[(size_t)b * waves + w], nullptr, std::min(64u, block - 64 * w));
    std::vector<std::thread> th;
    th.reserve((size_t)grid * block);
    for (unsigned b = 0; b < grid; ++b)
      for (unsigned t = 0; t < block; ++t)
        th.emplace_back([&, b, t] {
          emu::t_barrier = &bars[b];
          emu::t_wave_barrier = &wbars[(size_t)b * waves + t / 64];
          emu::t_smem = shared[b].data() + (64 - (reinterpret_cast<uintptr_t>(shared[b].data()) & 63)) % 64;
          emu::t_blockDim.x = block; emu::t_gridDim.x = grid; emu::t_threadIdx.x = t; emu::t_blockIdx.x = b;
          kernel(args...);
        });
    for (auto& t : th) t.join();
    for (auto& b : bars) pthread_barrier_destroy(&b);
    for (auto& w : wbars) pthread_barrier_destroy(&w);
  }
};

}  // namespace

extern "C" {

// registry metadata must agree with the LineCfg constants the kernels were compiled with
int emu_check_registry(char* msg, size_t msg_bytes) {
  using namespace mi355;
  int cur = 0, bad = 0;
  const auto& reg = line_kernel_registry();
#define CHECK(N_, R0_, R1_, R2_, T_, IC, OC, SI, SO, TW)                                                     \
  {                                                                                                          \
    using C = LineCfg<N_, R0_, R1_, R2_, T_, IC, OC, SI, SO, TW>;                                            \
    const LineKernelMeta& m = reg[(size_t)cur];                                                              \
    if (m.threads != C::THREADS || m.lds_bytes != C::LDS_BYTES || m.tw_elems != C::TW_ELEMS || m.N != N_) {  \
      std::snprintf(msg, msg_bytes, "registry mismatch at id %d (N=%d)", cur, N_);                           \
      ++bad;                                                                                                 \
    }                                                                                                        \
    ++cur;                                                                                                   \
  }
#define LINE_ROW(N, R0, R1, R2, T) CHECK(N, R0, R1, R2, T, false, false, false, false, 0) CHECK(N, R0, R1, R2, T, false, false, true, true, 0)
#define LINE_ROW_TRIG(N, R0, R1, R2, T) CHECK(N, R0, R1, R2, T, false, false, false, false, 4)
#define LINE_PASS_A(N, R0, R1, R2, T) CHECK(N, R0, R1, R2, T, true, true, false, false, 0) CHECK(N, R0, R1, R2, T, true, true, true, false, 0) CHECK(N, R0, R1, R2, T, true, true, true, true, 0)
#define LINE_PASS_B(N, R0, R1, R2, T) CHECK(N, R0, R1, R2, T, false, true, false, false, 2) CHECK(N, R0, R1, R2, T, false, true, false, true, 2)
#define LINE_COL_RAGGED(N, R0, R1, R2, T) CHECK(N, R0, R1, R2, T, true, true, false, false, 3) CHECK(N, R0, R1, R2, T, true, true, true, true, 3)
#include "line_kernels.def"
#undef LINE_ROW
#undef LINE_ROW_TRIG
#undef LINE_PASS_A
#undef LINE_PASS_B
#undef LINE_COL_RAGGED
#undef CHECK
  if (cur != (int)reg.size()) { std::snprintf(msg, msg_bytes, "registry size %d != %zu", cur, reg.size()); ++bad; }
  return bad;
}

// Plans `desc` and runs it on host buffers.  input/output/kernel are host pointers of at least the byte
// extents the plan reports (checked).  output == NULL for in-place plans.  Returns the planner status or
// 100+ for harness errors; err receives the message; route (optional) the plan's route string.
int emu_run_plan(const mi355fft_plan_desc* desc, void* input, uint64_t input_bytes, void* output, uint64_t output_bytes, void* kernel,
                 uint64_t kernel_bytes, int force_generic, uint64_t chunk_bytes, char* err, size_t err_bytes, char* route, size_t route_bytes,
                 int* launches) {
  using namespace mi355;
  PlannerOptions opt;
  opt.force_generic = force_generic;
  if (chunk_bytes) opt.chunk_bytes = chunk_bytes;
  opt.compute_units = std::getenv("MI355_EMU_CUS") ? std::atoi(std::getenv("MI355_EMU_CUS")) : 2;
  if (const char* e = std::getenv("MI355_EMU_XCD_FUSED")) opt.xcd_fused = std::atoi(e); else opt.xcd_fused = 0;
  if (const char* e = std::getenv("MI355_EMU_XCD_SPLIT")) opt.xcd_split = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_FUSE_VIEWS")) opt.fuse_views = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_RES")) opt.xcd_res = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_RES_DEPTH")) opt.xcd_res_depth = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_SLOTS")) opt.xcd_slots = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_MIXED_LINES")) opt.mixed_lines = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_MIXED_CT")) opt.mixed_ct = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_MAX_LINE")) opt.max_line = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_2D")) opt.xcd_2d = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_RT")) opt.xcd_rt = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_XCD_HX")) opt.xcd_hx = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_CONV_PIPELINE")) opt.conv_pipeline = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_LINES_R2C")) opt.lines_r2c = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_LINES_C2R")) opt.lines_c2r = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_TRIG_REAL")) opt.trig_real = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_TRIG_FUSED")) opt.trig_fused = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_TRIG_ALT")) opt.trig_alt = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_LINE32K")) opt.line32k = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_SOLO_MAX_KB")) opt.solo_max_kb = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_CONV_LINES")) opt.conv_lines = std::atoi(e);
  if (const char* e = std::getenv("MI355_EMU_CONV_FUSED_MAX_POINTS")) opt.conv_fused_max_points = std::atoll(e);
  emu::g_xcds = std::getenv("MI355_EMU_XCDS") ? (unsigned)std::atoi(std::getenv("MI355_EMU_XCDS")) : 2u;
  PlanIR ir;
  std::string e;
  const int rc = build_plan(*desc, opt, ir, e);
  if (rc) { std::snprintf(err, err_bytes, "%s", e.c_str()); return rc; }
  if (route) std::snprintf(route, route_bytes, "%s", ir.route.c_str());
  if (launches) *launches = (int)ir.steps.size();
  if (input_bytes < ir.in_bytes) { std::snprintf(err, err_bytes, "input too small: need %llu", (unsigned long long)ir.in_bytes); return 100; }
  if (!desc->in_place && output_bytes < ir.out_bytes) { std::snprintf(err, err_bytes, "output too small: need %llu", (unsigned long long)ir.out_bytes); return 101; }
  if (kernel_bytes < ir.kernel_bytes) { std::snprintf(err, err_bytes, "kernel too small: need %llu", (unsigned long long)ir.kernel_bytes); return 102; }
  std::vector<char> work(ir.work_bytes + 256);
  void* base[5] = {input, desc->in_place ? input : output, work.data(), kernel, ir.table.data()};
  EmuLauncher l;
  auto lines_fn = [&](int family, int id, const LineArgs& a, unsigned grid) -> bool {
    switch (family) {
      case FAM_ROW_SMALL: return launch_lines_family<FAM_ROW_SMALL>(id, a, grid, l);
      case FAM_ROW_1K: return launch_lines_family<FAM_ROW_1K>(id, a, grid, l);
      case FAM_ROW_BIG: return launch_lines_family<FAM_ROW_BIG>(id, a, grid, l);
      case FAM_PASS_A: return launch_lines_family<FAM_PASS_A>(id, a, grid, l);
      case FAM_PASS_B: return launch_lines_family<FAM_PASS_B>(id, a, grid, l);
    }
    return false;
  };
  for (const Step& s : ir.steps) {
    void* ptr[5];
    for (int i = 0; i < 5; ++i) ptr[i] = s.p[i].buf == BUF_NONE ? nullptr : (char*)base[s.p[i].buf] + s.p[i].off;
    auto xcd_fn = [&](int id, const XcdFusedArgs& a, unsigned grid) { return launch_xcd_fused(id, a, grid, l); };
    if (!dispatch_step(s, ptr, l, lines_fn, xcd_fn)) { std::snprintf(err, err_bytes, "no kernel for step kind %d variant %d", (int)s.kind, s.variant); return 103; }
  }
  if (l.sticky) { std::snprintf(err, err_bytes, "XCD-fused kernel gave up waiting (sticky=%u)", l.sticky); return 104; }
  return 0;
}

// Planner only (host logic tests: routes, guards, workspace sizes) — nothing is run, so shapes far beyond host memory are fine.
int emu_plan_only(const mi355fft_plan_desc* desc, int compute_units, char* err, size_t err_bytes, char* route, size_t route_bytes, int* launches,
                  uint64_t* work_bytes) {
  using namespace mi355;
  PlannerOptions opt = planner_options_from_env();
  opt.compute_units = compute_units > 0 ? compute_units : 256;
  PlanIR ir;
  std::string e;
  const int rc = build_plan(*desc, opt, ir, e);
  if (rc) { std::snprintf(err, err_bytes, "%s", e.c_str()); return rc; }
  if (route) std::snprintf(route, route_bytes, "%s", ir.route.c_str());
  if (launches) *launches = (int)ir.steps.size();
  if (work_bytes) *work_bytes = ir.work_bytes;
  return 0;
}

// support kernels: PRNG twin + reductions
int emu_fill_random(float* out, uint64_t row_floats, uint64_t rows, uint32_t seed0, uint64_t first_transform) {
  EmuLauncher l;
  l.launch(mi355::fill_random_kernel, 2u, 256u, 0u, out, (unsigned long long)row_floats, (unsigned long long)rows, (unsigned)seed0,
           (unsigned long long)first_transform);
  return 0;
}
int emu_diff_sumsq(const float* a, const float* b, double alpha, uint64_t count, double* out) {
  EmuLauncher l;
  double partial[2] = {0, 0};
  l.launch(mi355::diff_sumsq_kernel, 2u, 256u, 0u, a, b, alpha, (unsigned long long)count, (double*)partial);
  *out = partial[0] + partial[1];
  return 0;
}

}  // extern "C"
