// plan.cpp — see plan.hpp.  Pure host C++ (no HIP).
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mi355 {

// ---- registry --------------------------------------------------------------------------------------
namespace {
constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
constexpr int ilog2c(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

LineKernelMeta make_meta(int id, int N, int R0, int R1, int R2, int T, bool ic, bool oc, bool si, bool so, int twid) {
  // mirrors LineCfg in kern_lines.hpp (checked against the device constants by tests/emu and at library load)
  LineKernelMeta m{};
  m.id = id; m.N = N; m.R0 = R0; m.R1 = R1; m.R2 = R2; m.T = T;
  m.in_col = ic; m.out_col = oc; m.swap_in = si; m.swap_out = so; m.twid = twid;
  const int nst = R1 == 1 ? 1 : (R2 == 1 ? 2 : 3);
  const int rmax = cmax3(R0, R1, R2);
  const int tpl = N / rmax;
  m.threads = T * tpl;
  const bool idx_major = ic && oc;
  const int padsh = ilog2c(R0);
  const int pitch_raw = N + (N >> padsh);
  const int pmod = (!ic && oc && T <= 32) ? 32 / T : 2;
  const int pitch = ((pitch_raw + 31) / 32) * 32 + pmod;
  const int data = nst == 1 ? 0 : (idx_major ? N * T : T * pitch);
  const int tw1 = nst >= 2 ? (R1 - 1) * R0 : 0;
  const int tw2 = nst == 3 ? (R2 - 1) * R0 * R1 : 0;
  m.tw_elems = tw1 + tw2;
  const int lo = twid == 1 ? 1024 : 0;
  const int tw_lds = tw1 + (tw2 * 8 <= (twid == 4 ? 8 * 1024 : MI355_TW2_LDS_MAX) ? tw2 : 0);        // LineCfg::TW2_IN_LDS
  m.lds_bytes = (data + tw_lds + lo) * 8;
  return m;
}
}  // namespace

const std::vector<LineKernelMeta>& line_kernel_registry() {
  static const std::vector<LineKernelMeta> reg = [] {
    std::vector<LineKernelMeta> r;
    int id = 0;
#define LINE_ROW(N, R0, R1, R2, T)                                             \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, false, false, false, false, 0)); \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, false, false, true, true, 0));
#define LINE_ROW_TRIG(N, R0, R1, R2, T)                                        \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, false, false, false, false, 4));
#define LINE_PASS_A(N, R0, R1, R2, T)                                        \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, true, true, false, false, 0)); \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, true, true, true, false, 0));  \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, true, true, true, true, 0));
#define LINE_PASS_B(N, R0, R1, R2, T)                                         \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, false, true, false, false, 2)); \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, false, true, false, true, 2));
#define LINE_COL_RAGGED(N, R0, R1, R2, T)                                    \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, true, true, false, false, 3)); \
  r.push_back(make_meta(id++, N, R0, R1, R2, T, true, true, true, true, 3));
#include "line_kernels.def"
#undef LINE_ROW
#undef LINE_ROW_TRIG
#undef LINE_PASS_A
#undef LINE_PASS_B
#undef LINE_COL_RAGGED
    return r;
  }();
  return reg;
}

const LineKernelMeta* find_line_kernel(int N, bool in_col, bool out_col, bool swap_in, bool swap_out, int twid) {
  for (const auto& m : line_kernel_registry())
    if (m.N == N && m.in_col == in_col && m.out_col == out_col && m.swap_in == swap_in && m.swap_out == swap_out && m.twid == twid)
      return &m;
  return nullptr;
}

const std::vector<ConvKernelMeta>& conv_kernel_registry() {
  static const std::vector<ConvKernelMeta> reg = [] {
    std::vector<ConvKernelMeta> r;
    int id = 0;
#define X(N, R0, R1, TL) r.push_back(ConvKernelMeta{id++, N, R0, R1, TL});
    MI355_CONV_KERNEL_LIST(X)
#undef X
    return r;
  }();
  return reg;
}

#define MI_RT_COMMA(n) n,
const std::vector<XcdKernelMeta>& xcd_kernel_registry() {
  static const std::vector<XcdKernelMeta> reg = [] {
    std::vector<XcdKernelMeta> r;
    int id = 0;
#define XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, INV, REAL)                                       \
  {                                                                                                       \
    const LineKernelMeta ma = make_meta(0, N1, A0, A1, A2, TA, true, true, INV, false, 0);                \
    const LineKernelMeta mb = make_meta(0, N2, B0, B1, B2, TB, false, (REAL) != 3, false, INV, 0);        \
    XcdKernelMeta m{id++, N1, N2, {A0, A1, A2}, {B0, B1, B2}, TA, TB, INV, ma.threads, 0, REAL, 0};       \
    const int da = ma.lds_bytes - ma.tw_elems * 8, db = mb.lds_bytes - mb.tw_elems * 8;                   \
    const bool shared = N1 == N2 && A0 == B0 && A1 == B1 && A2 == B2;                                     \
    m.lds_bytes = (da > db ? da : db) + (ma.tw_elems + (shared ? 0 : mb.tw_elems)) * 8 + 64;              \
    r.push_back(m);                                                                                       \
  }
#define X(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB)                                                          \
  XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, false, 0)                                          \
  XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, true, 0)
    MI355_XCD_KERNEL_LIST(X)
#undef X
#define X(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB) XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, false, 1)
    MI355_XCD_R2C_KERNEL_LIST(X)
#undef X
#define X(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB) XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, false, 2)
    MI355_XCD_C2R_KERNEL_LIST(X)
#undef X
#define X(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB)                                                          \
  XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, false, 3)                                              \
  XCD_META(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB, true, 3)
    MI355_XCD_2D_KERNEL_LIST(X)
#undef X
#undef XCD_META
    // register-tile instances (kern_regtile.hpp XcdRtCfg): 512 threads; LDS = max(pass A's LDS tile, one exchange half) + tables
    for (int n1 : {MI355_XCD_RT_KERNEL_LIST(MI_RT_COMMA)}) for (int inv = 0; inv < 2; ++inv) {
      const bool a_rt = n1 == 2048;
      const LineKernelMeta ma = make_meta(0, a_rt ? 1024 : n1, 32, (a_rt ? 1024 : n1) / 32, 1, 16, true, true, inv != 0, false, 0);
      XcdKernelMeta m{id++, n1, 2048, {32, a_rt ? 64 : n1 / 32, 1}, {64, 32, 1}, 16, 16, inv != 0, 512, 0, 0, 1};
      const int data_a = a_rt ? 0 : ma.lds_bytes - ma.tw_elems * 8, tw_a = a_rt ? 0 : ma.tw_elems * 8;
      m.lds_bytes = std::max(data_a, 16 * 32 * 32 * 8) + tw_a + 31 * 64 * 8 + 64;
      r.push_back(m);
    }
    { XcdKernelMeta m{id++, 2048, 2048, {64, 32, 1}, {64, 32, 1}, 16, 16, false, 512, (16 * 32 * 32 + 31 * 64) * 8 + 64, 1, 1}; r.push_back(m); }   // r2c 2048 x 2048 (fft_xcd_rt_r2c_kernel)
    for (int inv = 0; inv < 2; ++inv) {   // 1024 x 1024 with two workgroups per CU (fft_xcd_hx_kernel): rt = 2
      XcdKernelMeta m{id++, 1024, 1024, {32, 32, 1}, {32, 32, 1}, 16, 16, inv != 0, 512, (16 * 32 * 16 + 31 * 32) * 8 + 64, 0, 2}; r.push_back(m);
    }
    { XcdKernelMeta m{id++, 2048, 2048, {64, 32, 1}, {64, 32, 1}, 16, 16, false, 512, (16 * 32 * 32 + 31 * 64) * 8 + 64, 2, 1}; r.push_back(m); }   // c2r 2048 x 2048 (fft_xcd_rt_c2r_kernel)
    for (int inv = 0; inv < 2; ++inv) {   // 1024 x 1024 on 32-line register tiles (fft_xcd_rt1k_kernel): rt = 3
      XcdKernelMeta m{id++, 1024, 1024, {32, 32, 1}, {32, 32, 1}, 32, 32, inv != 0, 512, (32 * 32 * 16 + 31 * 32) * 8 + 64, 0, 3}; r.push_back(m);
    }
    { XcdKernelMeta m{id++, 1024, 1024, {32, 32, 1}, {32, 32, 1}, 32, 32, false, 512, (32 * 32 * 16 + 31 * 32) * 8 + 64, 4, 3}; r.push_back(m); }   // real = 4: fftconv pipeline for 2^20 points (fft_xcd_conv1m_kernel)
    { const LineKernelMeta ma = make_meta(0, 1024, 32, 32, 1, 16, true, true, false, false, 0);   // r2c 1024 x 2048: LDS-resident pass A + register-tile pass B (fft_xcd_rt_r2c_kernel<1024>)
      XcdKernelMeta m{id++, 1024, 2048, {32, 32, 1}, {64, 32, 1}, 16, 16, false, 512, 0, 1, 1};
      m.lds_bytes = std::max(ma.lds_bytes - ma.tw_elems * 8, 16 * 32 * 32 * 8) + ma.tw_elems * 8 + 31 * 64 * 8 + 64; r.push_back(m); }
    { const LineKernelMeta ma = make_meta(0, 1024, 32, 32, 1, 16, true, true, false, false, 0);   // c2r 1024 x 2048 (fft_xcd_rt_c2r_kernel<1024>)
      XcdKernelMeta m{id++, 1024, 2048, {32, 32, 1}, {64, 32, 1}, 16, 16, false, 512, 0, 2, 1};
      m.lds_bytes = std::max(ma.lds_bytes - ma.tw_elems * 8, 16 * 32 * 32 * 8) + ma.tw_elems * 8 + 31 * 64 * 8 + 64; r.push_back(m); }
    for (int inv = 0; inv < 2; ++inv) {   // 1024 x 1024 on 16-line register tiles, 256 threads, two workgroups per CU (fft_xcd_rt1k_kernel<.., 16>): rt = 5
      XcdKernelMeta m{id++, 1024, 1024, {32, 32, 1}, {32, 32, 1}, 16, 16, inv != 0, 256, (16 * 32 * 16 + 31 * 32) * 8 + 64, 0, 5}; r.push_back(m);
    }
    for (int inv = 0; inv < 2; ++inv) {   // VIEW instances of the 32-line 1024 x 1024 kernel (rank-1 ioView / zeroPad as load / store predicates): rt = 6
      XcdKernelMeta m{id++, 1024, 1024, {32, 32, 1}, {32, 32, 1}, 32, 32, inv != 0, 512, (32 * 32 * 16 + 31 * 32) * 8 + 64, 0, 6}; r.push_back(m);
    }
#define X(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB)                                                          \
  for (int inv = 0; inv < 2; ++inv) {   /* VIEW instances of the LDS-resident fused kernel: rt = 6 */     \
    const LineKernelMeta ma = make_meta(0, N1, A0, A1, A2, TA, true, true, inv != 0, false, 0);           \
    const LineKernelMeta mb = make_meta(0, N2, B0, B1, B2, TB, false, true, false, inv != 0, 0);          \
    XcdKernelMeta m{id++, N1, N2, {A0, A1, A2}, {B0, B1, B2}, TA, TB, inv != 0, ma.threads, 0, 0, 6};     \
    const int da = ma.lds_bytes - ma.tw_elems * 8, db = mb.lds_bytes - mb.tw_elems * 8;                   \
    const bool shared = N1 == N2 && A0 == B0 && A1 == B1 && A2 == B2;                                     \
    m.lds_bytes = (da > db ? da : db) + (ma.tw_elems + (shared ? 0 : mb.tw_elems)) * 8 + 64;              \
    r.push_back(m);                                                                                       \
  }
    MI355_XCD_VIEW_KERNEL_LIST(X)
#undef X
    for (int inv = 0; inv < 2; ++inv) {   // 2^21 as 2048 x 1024: 16-line register tiles down the columns, 32-line register tiles along the rows (fft_xcd_rt1k_kernel<.., 2048>): rt = 7
      XcdKernelMeta m{id++, 2048, 1024, {64, 32, 1}, {32, 32, 1}, 16, 32, inv != 0, 512, (32 * 32 * 16 + 31 * 32 + 31 * 64) * 8 + 64, 0, 7}; r.push_back(m);
    }
    return r;
  }();
  return reg;
}

PlannerOptions planner_options_from_env() {
  PlannerOptions o;
  if (const char* s = std::getenv("MI355FFT_CHUNK_BYTES")) { const int64_t v = std::atoll(s); if (v > 0) o.chunk_bytes = (uint64_t)v; }
  if (const char* s = std::getenv("MI355FFT_FORCE_GENERIC")) o.force_generic = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_MIXED_LINES")) o.mixed_lines = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_MIXED_CT")) o.mixed_ct = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_LINES_R2C")) o.lines_r2c = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_LINES_C2R")) o.lines_c2r = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_TRIG_REAL")) o.trig_real = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_TRIG_FUSED")) o.trig_fused = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_CONV_LINES")) o.conv_lines = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_CONV_FUSED_MAX_POINTS")) { const long long v = std::atoll(s); if (v >= 0) o.conv_fused_max_points = v; }
  if (const char* s = std::getenv("MI355FFT_MAX_LINE")) { const int v = std::atoi(s); if (v >= 4096) o.max_line = v; }
  if (const char* s = std::getenv("MI355FFT_MIXED_LDS_KB")) { const int v = std::atoi(s); if (v >= 8 && v <= 128) o.mixed_lds_kb = v; }
  if (const char* s = std::getenv("MI355FFT_MIXED_THREADS")) { const int v = std::atoi(s); if (v >= 64 && v <= 512 && v % 64 == 0) o.mixed_threads = v; }
  if (const char* s = std::getenv("MI355FFT_ONLY_PASS")) o.only_pass = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_FUSE_VIEWS")) o.fuse_views = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_LINES_TILES_PER_WG")) { const int v = std::atoi(s); if (v >= -1) o.lines_tiles_per_wg = v; }
  if (const char* s = std::getenv("MI355FFT_XCD_FUSED")) o.xcd_fused = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_TRIG_ALT")) o.trig_alt = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_LINE32K")) o.line32k = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_XCD_RES")) o.xcd_res = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_XCD_SPIN_LIMIT")) { const long long v = std::atoll(s); if (v >= 1 && v <= 0x7fffffffll) o.xcd_spin_limit = (unsigned)v; }
  if (const char* s = std::getenv("MI355FFT_XCD_RES_DEPTH")) { const int v = std::atoi(s); if (v == 1 || v == 2 || v == 4) o.xcd_res_depth = v; }
  if (const char* s = std::getenv("MI355FFT_XCD_SPLIT")) { const int v = std::atoi(s); if (v >= 0 && v <= 32) o.xcd_split = v; }
  if (const char* s = std::getenv("MI355FFT_XCD_R2C")) o.xcd_r2c = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_XCD_2D")) o.xcd_2d = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_XCD_RT")) o.xcd_rt = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_XCD_HX")) o.xcd_hx = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_CONV_PIPELINE")) o.conv_pipeline = std::atoi(s);
  if (const char* s = std::getenv("MI355FFT_SOLO_MAX_KB")) { const int v = std::atoi(s); if (v >= 0) o.solo_max_kb = v; }
  if (const char* s = std::getenv("MI355FFT_SOLO_CAP_MB")) { const int v = std::atoi(s); if (v >= 1) o.solo_cap_mb = v; }
  if (const char* s = std::getenv("MI355FFT_XCD_SLOTS")) { const int v = std::atoi(s); if (v >= 0 && v <= 2) o.xcd_slots = v; }
  return o;
}

const std::vector<MixedCtMeta>& mixedct_registry() {
  static const std::vector<MixedCtMeta> reg = [] {
    std::vector<MixedCtMeta> r;
    int id = 0;
#define X(N_, T_, TH_, ...) { MixedCtMeta m; m.id = id++; m.N = N_; m.T = T_; m.threads = TH_; m.radices = std::vector<int>{__VA_ARGS__}; \
    int tw = 0, nsp = 1; for (int R : m.radices) { tw += R * nsp; nsp *= R; } m.lds_bytes = (2 * T_ * (N_ + (N_ >> 5) + 1) + tw) * 8; r.push_back(m); }
    MI355_MIXEDCT_LIST(X)
#undef X
    return r;
  }();
  return reg;
}

// max_radix < 32: the one-launch mixed-radix kernel keeps a line in LDS, where a stage is cheap but its parallelism is
// N / radix butterflies per line — radices above 8 starve the workgroup (measured: N = 640 as 32*4*5 30 GPoints/s)
std::vector<int> factorize_radices(int64_t n, int max_radix) {
  static const int allowed[] = {32, 16, 8, 4, 2, 13, 11, 7, 5, 3};
  std::vector<int> out;
  for (int r : allowed) {
    if (r > max_radix && (r & (r - 1)) == 0) continue;
    while (n % r == 0 && n > 1) { out.push_back(r); n /= r; }
  }
  if (n != 1) out.clear();
  return out;
}
float2h root_of_unity(int64_t m, int64_t M) {
  m %= M;
  if (m < 0) m += M;
  const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)M;
  float2h r;
  r.x = (float)cosl(ang);
  r.y = (float)sinl(ang);
  // exact values on the axes (cosl(pi/2) is ~1e-20, harmless, but keep tables clean)
  if (4 * m == M) { r.x = 0.0f; r.y = -1.0f; }
  if (2 * m == M) { r.x = -1.0f; r.y = 0.0f; }
  if (4 * m == 3 * M) { r.x = 0.0f; r.y = 1.0f; }
  if (m == 0) { r.x = 1.0f; r.y = 0.0f; }
  return r;
}

// ---- builder ---------------------------------------------------------------------------------------
namespace {

bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }
int lg2(int64_t n) { int l = 0; while (((int64_t)1 << l) < n) ++l; return l; }
int64_t prodv(const int64_t* s, int rank) { int64_t p = 1; for (int d = 0; d < rank; ++d) p *= s[d]; return p; }
uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Builder {
  PlanIR& ir;
  const PlannerOptions& opt;
  uint64_t work_top = 0;
  // rank-1 lane layouts on the four-step sizes (build_c2c): the pitches between consecutive lines that the fused kernel of the next
  // emit_axis should use instead of N; lane_used reports that a fused launch took them (any other route ignores them)
  int64_t lane_in_pitch = 0, lane_out_pitch = 0;
  bool lane_used = false;
  Builder(PlanIR& i, const PlannerOptions& o) : ir(i), opt(o) {}

  PtrRef alloc_work(uint64_t bytes) {
    PtrRef r(BUF_WORK, (int64_t)work_top);
    work_top = align_up(work_top + bytes, 256);
    if (work_top > ir.work_bytes) ir.work_bytes = work_top;
    return r;
  }
  PtrRef add_table(const std::vector<float2h>& t) {
    // 256-byte aligned start so LDS staging loads stay aligned
    while ((ir.table.size() * sizeof(float2h)) % 256 != 0) ir.table.push_back(float2h{0, 0});
    PtrRef r(BUF_TABLE, (int64_t)(ir.table.size() * sizeof(float2h)));
    ir.table.insert(ir.table.end(), t.begin(), t.end());
    return r;
  }
  // element-wise kernels (grid-stride loops of 256 threads): one element per thread.  Round 1 capped the grid at 16 workgroups per
  // CU; one-shot grids stream 15-20 % faster on this chip (profiles/r02_copy_ceiling.log), so the cap is only an overflow guard.
  // HIP refuses a launch whose grid x block reaches 2^32 threads: every kernel here walks its items with a grid-stride loop, so
  // grids are capped well below that for workgroups of up to 1024 threads
  static constexpr int64_t MAX_BLOCKS = ((int64_t)1 << 22) - 1;
  unsigned generic_grid(int64_t total) const {
    const int64_t blocks = (total + 255) / 256;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(blocks, MAX_BLOCKS));
  }
  // streaming kernels whose work items are independent chunks: one short-lived workgroup per item.  One-shot grids stream at
  // 6.2-6.5 TB/s on this chip where resident grid-stride loops reach 5.3-5.5 (profiles/r02_copy_ceiling.log)
  // work items of r2c_post_kernel / c2r_pre_kernel (kern_generic.hpp): 512 bin pairs each; lines of at most 256 pairs sit side by side
  static int64_t split_items(int64_t lines, int64_t H) {
    const int64_t pairs = H / 4 + 1;
    if (pairs > 256) return lines * ((pairs + 511) / 512);
    int sh = 0;
    while (((int64_t)1 << sh) < pairs) ++sh;
    const int64_t per_item = 2 * (256 >> sh);
    return (lines + per_item - 1) / per_item;
  }
  unsigned oneshot_grid(int64_t items) const { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(items, MAX_BLOCKS)); }
  Step& push(StepKind k) { ir.steps.emplace_back(); ir.steps.back().kind = k; return ir.steps.back(); }

  // stage-2 roots of the register-tile passes (kern_regtile.hpp): e^{-2 pi i q2 j2/2048}, rows q2 = 1..31, j2 = 0..63 fastest
  PtrRef regtile_table() {
    std::vector<float2h> t((size_t)31 * 64);
    for (int q = 1; q < 32; ++q) for (int j = 0; j < 64; ++j) t[(size_t)(q - 1) * 64 + j] = root_of_unity((int64_t)q * j, 2048);
    return add_table(t);
  }
  // stage tables of a line kernel: stage 1 [R1-1][R0] roots of order R0*R1, stage 2 [R2-1][R0*R1] of order N
  PtrRef line_tables(const LineKernelMeta& m) {
    std::vector<float2h> t;
    if (m.R1 > 1) {
      const int64_t ns = (int64_t)m.R0 * m.R1;
      for (int q = 1; q < m.R1; ++q) for (int k = 0; k < m.R0; ++k) t.push_back(root_of_unity((int64_t)q * k, ns));
    }
    if (m.R2 > 1) {
      const int64_t nsp = (int64_t)m.R0 * m.R1;
      for (int q = 1; q < m.R2; ++q) for (int64_t k = 0; k < nsp; ++k) t.push_back(root_of_unity(q * k, m.N));
    }
    if (t.empty()) t.push_back(float2h{1, 0});
    return add_table(t);
  }

  // e^{-2 pi i k/M} for k < count as HI[k >> 10] * LO[k & 1023]: tables into slots p[2] (LO), p[3] (HI), i[3] shift, i[4] mask
  void split_roots(Step& st, int64_t M, int64_t count) {
    std::vector<float2h> lo(1024), hi((size_t)((count + 1023) >> 10));
    for (int64_t l = 0; l < 1024; ++l) lo[(size_t)l] = root_of_unity(l, M);
    for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << 10, M);
    const PtrRef plo = add_table(lo), phi = add_table(hi);   // add_table may reallocate ir.steps? no: tables live in ir.table
    st.p[2] = plo; st.p[3] = phi; st.i[3] = 10; st.i[4] = 1023;
  }

  unsigned lines_grid(const LineKernelMeta& m, int64_t tiles, bool plain_c2c = false) const {
    int64_t per_cu = 8;
    if (m.lds_bytes > 0) per_cu = std::min<int64_t>(per_cu, (160 * 1024) / m.lds_bytes);
    per_cu = std::min<int64_t>(per_cu, 2048 / m.threads);
    per_cu = std::max<int64_t>(per_cu, 1);
    // lines_tiles_per_wg > 0: a short-lived workgroup per `lines_tiles_per_wg` tiles instead of a resident grid walking the batch
    // (one-shot grids stream at 6.2-6.5 TB/s where persistent loops reach 5.3-5.5: profiles/r02_copy_ceiling.log)
    // default (0): ROW kernels of up to 1024 points take one tile per workgroup (measured +10...+14 % at N = 64..512, +6 % at 1024:
    // profiles/r02_oneshot_grids.log); longer lines would re-stage tables of a quarter of a tile or more per workgroup and the
    // PASS kernels hoist per-launch state, so they stay resident.  -1 keeps every line kernel resident.
    int tpw = opt.lines_tiles_per_wg;
    if (tpw == 0 && plain_c2c && !m.in_col && !m.out_col && m.twid == 0 && m.N <= 1024) tpw = 1;   // (the r2c / c2r / product variants measured better resident)
    if (tpw > 0) return (unsigned)std::max<int64_t>(1, std::min<int64_t>((tiles + tpw - 1) / tpw, MAX_BLOCKS));
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, per_cu * opt.compute_units));
  }

  // batched 1-D FFT along an axis of a dense array: `lines` = S*outer lines of length N, element stride S.
  //   src/dst may be the same location.  inverse => e^{+...}.  scale fused into the last launch.
  // groups per XCD for the fused kernels.  Measured (profiles/r01_xcd_fused_ab.log): more, smaller groups are better as long
  // as all their workspace slots together stay within the 256 MiB Infinity Cache (8 XCDs x split x slots x slot bytes)
  // r03 (profiles/r03_headline_rt32_ab.log): at EQUAL footprint, twice the groups with ONE slot each (two barriers per transform) beat
  // half the groups with two slots (one barrier): c2c 2^20 184 -> 194 (LDS-resident kernel), 2^17 179 -> 192, r2c 2^21 216 -> 245,
  // c2r 2^20 289 -> 300.  So: one slot per group, and as many groups per XCD (a power of two, at most 16) as keep all slots of the chip
  // within the 256 MiB Infinity Cache.  The 2048-point register-tile instances keep two slots (their single group per XCD measured
  // the same either way).  MI355FFT_XCD_SPLIT / MI355FFT_XCD_SLOTS override.
  void xcd_groups(uint64_t slot_bytes, bool two_slot_default, int64_t& split, int64_t& slots) const {
    slots = opt.xcd_slots > 0 ? opt.xcd_slots : (two_slot_default ? 2 : 1);
    if (opt.xcd_split > 0) { split = opt.xcd_split; return; }
    split = 16;
    while (split > 1 && (uint64_t)(8 * slots * split) * slot_bytes > ((uint64_t)(two_slot_default ? 320 : 256) << 20)) split >>= 1;
  }

  // 2-D c2c planes [N1][N0] (axis 0 = N0 fastest) through the fused kernel's TWO_D instances; false if none applies
  bool emit_xcd_2d(PtrRef src, PtrRef dst, int64_t N0, int64_t N1, int64_t planes, bool inverse, float scale) {
    if (opt.force_generic || opt.xcd_fused != 1 || opt.only_pass || !opt.xcd_2d) return false;
    const XcdKernelMeta* xm = nullptr;
    for (const auto& m : xcd_kernel_registry()) if (m.real == 3 && m.N1 == N1 && m.N2 == N0 && m.inverse == inverse) xm = &m;
    if (!xm) return false;
    const int64_t N = N0 * N1;
    const LineKernelMeta ma = make_meta(0, xm->N1, xm->ra[0], xm->ra[1], xm->ra[2], xm->ta, true, true, false, false, 0);
    const LineKernelMeta mb = make_meta(0, xm->N2, xm->rb[0], xm->rb[1], xm->rb[2], xm->tb, false, false, false, false, 0);
    const bool solo = (uint64_t)N * 8 <= ((uint64_t)opt.solo_max_kb_2d << 10);
    if (!solo && !opt.xcd_shared) return false;
    int64_t split = 1, grid = opt.compute_units, slots = 1;
    PtrRef wslots, ctl;
    if (solo) {
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((160 * 1024) / xm->lds_bytes, 2048 / xm->threads), 4));
      grid = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)opt.compute_units * per_cu, ((int64_t)opt.solo_cap_mb << 20) / (N * 8)), planes));
      slots = 1;
      wslots = alloc_work((uint64_t)grid * N * 8);
      ctl = alloc_work(256);
    } else {
      xcd_groups((uint64_t)N * 8, false, split, slots);
      wslots = alloc_work((uint64_t)(16 * slots * split) * N * 8);
      ctl = alloc_work(40960);
      if (xm->threads <= 256 && xm->lds_bytes <= 80 * 1024) grid *= 2;
    }
    std::vector<float2h> one(1, float2h{1, 0});
    const PtrRef ta = line_tables(ma), tb = line_tables(mb), tone = add_table(one);
    if (!solo) { Step& z = push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
    Step& st = push(ST_XCD_FUSED);
    st.variant = xm->id;
    st.p[0] = src; st.p[1] = dst; st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
    st.i[0] = planes; st.i[1] = N; st.i[2] = 0; st.i[3] = 0; st.i[9] = N; st.i[10] = N;
    st.i[4] = ta.off; st.i[5] = tb.off; st.i[6] = tone.off; st.i[7] = tone.off; st.i[8] = split; st.i[11] = slots; st.i[12] = solo ? 1 : 0; st.i[13] = opt.xcd_spin_limit;
    st.f[0] = scale;
    st.grid = (unsigned)grid;
    ir.route += std::string(solo ? "xcd-2d-solo[" : "xcd-2d[") + std::to_string(N0) + "x" + std::to_string(N1) + "] ";
    return true;
  }

  // XCD-fused r2c of `lines` dense real lines of length N into packed spectra of N/2+1 bins; false if no instance applies
  bool emit_xcd_r2c(PtrRef src, PtrRef dst, int64_t N, int64_t lines, float scale, bool c2r = false) {
    if (opt.force_generic || !opt.xcd_fused || !opt.xcd_r2c || opt.only_pass || N < 4096 || (N & (N - 1))) return false;
    const int lgf = lg2(N);
    const int64_t F1 = (int64_t)1 << (lgf / 2), F2 = N / F1;
    const XcdKernelMeta* xm = nullptr;
    for (const auto& m : xcd_kernel_registry())
      if (m.real == (c2r ? 2 : 1) && m.N1 == F1 && m.N2 == F2 && (!m.rt || (opt.xcd_rt && opt.xcd_shared))) xm = &m;
    if (!xm || (N <= 8192 && opt.xcd_fused != 2)) return false;
    const LineKernelMeta ma = make_meta(0, xm->rt ? 1024 : xm->N1, xm->rt ? 32 : xm->ra[0], xm->ra[1], xm->ra[2], xm->ta, true, true, false, false, 0);
    const LineKernelMeta mb = make_meta(0, xm->rt ? 1024 : xm->N2, xm->rt ? 32 : xm->rb[0], xm->rb[1], xm->rb[2], xm->tb, false, true, false, false, 0);
    const int64_t wsize = c2r ? (xm->rt ? (F1 / 2) * F2 : F1 * (F2 / 2 + 16)) : (F1 / 2 + 1) * F2;   // r2c: rows 0..N1/2; c2r: columns 0..N2/2 (+ padding; register tiles: N1/2 packed full rows)
    // small transforms: one workgroup per transform (solo mode, see emit_axis); the real line is N*4 bytes
    const bool solo = (uint64_t)N * 4 <= ((uint64_t)opt.solo_max_kb << 10) / (c2r ? 1 : 2) && opt.xcd_fused != 2;
    if (!solo && !opt.xcd_shared) return false;
    int64_t split = 1, grid = opt.compute_units, slots = 1;
    PtrRef wslots, ctl;
    if (solo) {
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((160 * 1024) / xm->lds_bytes, 2048 / xm->threads), 4));
      grid = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)opt.compute_units * per_cu, ((int64_t)opt.solo_cap_mb << 20) / (wsize * 8)), lines));
      slots = 1;
      wslots = alloc_work((uint64_t)grid * wsize * 8);
      ctl = alloc_work(256);
    } else {
      xcd_groups((uint64_t)wsize * 8, false, split, slots);
      // 2048 x 2048 real transforms: two groups per XCD with one slot each (8 x 2 x 16.8 MB, a little over the Infinity Cache) measured
      // ahead of one group with two slots on three of four boxes (r2c 288 vs 279, 282 vs 270, 269 vs 264, 232 vs 238: profiles/r03_regtile_ab.log)
      if (xm->rt == 1 && xm->N1 == 2048 && opt.xcd_split <= 0 && opt.xcd_slots <= 0) { split = 2; slots = 1; }
      wslots = alloc_work((uint64_t)(16 * slots * split) * wsize * 8);
      ctl = alloc_work(40960);
    }
    const int shift = N >= (1 << 20) ? 10 : lgf / 2;                // LO table of 2^shift roots, HI of N >> shift
    std::vector<float2h> lo((size_t)1 << shift), hi((size_t)std::max<int64_t>(1, N >> shift));
    for (size_t l = 0; l < lo.size(); ++l) lo[l] = root_of_unity((int64_t)l, N);
    for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << shift, N);
    const PtrRef tb = xm->rt ? regtile_table() : line_tables(mb), ta = (xm->rt && xm->N1 == 2048) ? tb : line_tables(ma), tlo = add_table(lo), thi = add_table(hi);
    if (!solo) { Step& z = push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
    Step& st = push(ST_XCD_FUSED);
    st.variant = xm->id;
    st.p[0] = src; st.p[1] = dst; st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
    st.i[0] = lines; st.i[1] = N; st.i[2] = shift; st.i[3] = ((int64_t)1 << shift) - 1;
    st.i[9] = c2r ? N / 2 + 1 : N / 2; st.i[10] = c2r ? N / 2 : N / 2 + 1;        // pitches in complex elements
    st.i[4] = ta.off; st.i[5] = tb.off; st.i[6] = tlo.off; st.i[7] = thi.off; st.i[8] = split; st.i[11] = slots; st.i[12] = solo ? 1 : 0; st.i[13] = opt.xcd_spin_limit;
    st.f[0] = scale;
    if (!solo && opt.xcd_fused != 2 && xm->threads <= 256 && xm->lds_bytes <= 80 * 1024) grid *= 2;   // as emit_axis: two co-resident workgroups per CU
    st.grid = (unsigned)grid;
    ir.route += std::string(c2r ? (solo ? "xcd-c2r-solo[N=" : xm->rt ? "xcd-c2r-rt[N=" : "xcd-c2r[N=") : (solo ? "xcd-r2c-solo[N=" : xm->rt ? "xcd-r2c-rt[N=" : "xcd-r2c[N=")) + std::to_string(xm->N1) + "x" + std::to_string(xm->N2) + "] ";
    return true;
  }

  // r2c of dense real lines of length N = 2H, H a power of two in 64..max_line: the ROW line kernel of length H with the split
  // fused behind its last stage (one launch instead of FFT + r2c_post_kernel)
  // (c2r: the mirror — the pre-split rides the first-stage loads of the INVERSE line kernel, any power-of-two half length >= 2)
  // trig = 5 / 6: the same launch as a whole DCT-II / DST-II of the real lines (kern_lines.hpp fft_lines_r2c_kernel<C, TRIG>)
  // im / om (optional, both or none): the launch reads its input side through im and writes through om (fft_lines_r2c_kernel /
  // fft_lines_c2r_kernel <.., MAPPED>; the real side's map counts floats, the packed side's complex bins)
  const LineKernelMeta* lines_r2c_kernel(int64_t N, bool c2r, bool mapped) const {
    const int64_t H = N / 2;
    if (opt.force_generic || !(c2r ? opt.lines_c2r : opt.lines_r2c) || (N & 1) || !is_pow2(H) || H < (c2r ? 2 : 64) || H > opt.max_line || (opt.xcd_fused == 2 && N == 4096)) return nullptr;
    // (N = 2^15 c2r: round 1 kept the Hermitian four-step in solo mode there, 322 vs 304; with the 16-byte pre-split accesses of r02
    // the line kernel is ahead, 347 vs 320; lines_c2r = 3 restores the old choice)   // xcd_fused == 2: emulation tests of the fused instances
    if (c2r && H > 8192 && opt.lines_c2r == 3 && !mapped) return nullptr;
    const LineKernelMeta* m = find_line_kernel((int)H, false, false, c2r, c2r, 0);
    if (!m || (!c2r && m->lds_bytes == 0)) return nullptr;
    if (mapped && (!opt.fuse_views || m->lds_bytes == 0 || m->R1 <= 1)) return nullptr;   // the mapped forms keep the line in LDS
    return m;
  }
  bool emit_lines_r2c(PtrRef src, PtrRef dst, int64_t N, int64_t lines, float scale, bool c2r = false, int trig = 0,
                      const SideMap* im = nullptr, const SideMap* om = nullptr) {
    const int64_t H = N / 2;
    const LineKernelMeta* m = lines_r2c_kernel(N, c2r, im != nullptr);
    if (!m) return false;
    // DCT-II / DST-II launches: the alternate shapes (DCT-III / DST-III on the c2r kernel measured -9...+5 % with them and keep the plain ones)
    if (trig && !c2r && opt.trig_alt) { if (const LineKernelMeta* alt = find_line_kernel((int)H, false, false, false, false, 4)) m = alt; }
    std::vector<float2h> lo(1024), hi((size_t)std::max<int64_t>(1, (H + 1023) >> 10));
    for (int64_t l = 0; l < 1024; ++l) lo[(size_t)l] = root_of_unity(l, N);
    for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << 10, N);
    // trig: the DCT phases e^{-i pi m/2N} = e^{-2 pi i m/4N}, m = 0..N/2, f64-built, directly behind the 1024 LO roots (one table)
    if (trig) for (int64_t mm = 0; mm <= H; ++mm) lo.push_back(root_of_unity(mm, 4 * N));
    Step& st = push(ST_LINES);
    st.variant = m->id;
    st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*m); st.p[3] = add_table(lo); st.p[4] = add_table(hi);
    const int64_t tiles = (lines + m->T - 1) / m->T;
    st.i[0] = tiles; st.i[1] = lines; st.i[2] = 1; st.i[3] = c2r ? H + 1 : H; st.i[4] = 1; st.i[5] = c2r ? H : H + 1; st.i[6] = 10; st.i[7] = 1023;
    st.i[9] = trig ? trig : (c2r ? 2 : 1);
    st.f[0] = scale;
    st.grid = lines_grid(*m, tiles);
    if (im) {
      st.i[10] = 1; st.imap = *im; st.omap = *om; st.imap.ax = st.omap.ax = 0;
      ir.route += std::string(c2r ? "lines-c2r-mapped[N=" : "lines-r2c-mapped[N=") + std::to_string(N) + "] ";
      return true;
    }
    ir.route += std::string(trig == 5 ? "lines-dct2[N=" : trig == 6 ? "lines-dst2[N=" : trig == 7 ? "lines-dct3[N=" : trig == 8 ? "lines-dst3[N=" : c2r ? "lines-c2r[N=" : "lines-r2c[N=") + std::to_string(N) + "] ";
    return true;
  }

  // 1-D r2c of `lines` dense real lines of even length N into packed spectra (N/2+1 bins): fused line kernel, real four-step,
  // or the half-length complex FFT + split
  int emit_r2c_even(PtrRef in, PtrRef out, int64_t N, int64_t lines, float scale, std::string& err) {
    if (emit_lines_r2c(in, out, N, lines, scale)) return MI355FFT_OK;   // one launch: split fused behind the last stage (fft_lines_r2c_kernel)
    // N = 2^16: the half-length route over the single-workgroup line of 2^15 points measured 318 vs 269 G real points/s for the
    // real four-step in solo mode (c2r: 318 vs 305)
    const bool half32k = N == 65536 && opt.line32k && opt.max_line >= 16384 && !opt.force_generic && !opt.only_pass && opt.xcd_fused != 2;
    if (!half32k && emit_xcd_r2c(in, out, N, lines, scale)) return MI355FFT_OK;     // one persistent launch: real four-step (kern_xcd_real.hpp)
    const int64_t H = N / 2, P = H + 1;
    PtrRef z = alloc_work((uint64_t)lines * H * 8);
    // the real input, read as `lines` complex lines of length H: z[n] = x[2n] + i x[2n+1]
    int rc = emit_axis(in, z, H, 1, lines, false, 1.0f, err);
    if (rc) return rc;
    Step& st = push(ST_R2C_POST);
    st.p[0] = z; st.p[1] = out;
    split_roots(st, N, H / 2 + 1);
    st.i[0] = H; st.i[1] = lines; st.i[2] = P; st.f[0] = scale;
    st.grid = oneshot_grid(split_items(lines, H));      // items of 512 bin pairs (r2c_post_kernel)
    ir.route += "r2c-split ";
    return MI355FFT_OK;
  }
  // the mirror: packed spectra -> real lines (unnormalised inverse times `scale`)
  int emit_c2r_even(PtrRef packed, PtrRef out, int64_t N, int64_t lines, float scale, std::string& err) {
    if (emit_lines_r2c(packed, out, N, lines, scale, true)) return MI355FFT_OK;   // pre-split in the first-stage loads (fft_lines_c2r_kernel)
    const bool half32k = N == 65536 && opt.line32k && opt.max_line >= 16384 && !opt.force_generic && !opt.only_pass && opt.xcd_fused != 2;
    if (!half32k && emit_xcd_r2c(packed, out, N, lines, scale, true)) return MI355FFT_OK;     // Hermitian four-step (kern_xcd_real.hpp)
    const int64_t H = N / 2, P = H + 1;
    PtrRef z = alloc_work((uint64_t)lines * H * 8);
    Step& st = push(ST_C2R_PRE);
    st.p[0] = packed; st.p[1] = z;
    split_roots(st, N, H / 2 + 1);
    st.i[0] = H; st.i[1] = lines; st.i[2] = P;
    st.grid = oneshot_grid(split_items(lines, H));      // items of 512 bin pairs (c2r_pre_kernel)
    // unnormalised inverse of length H lands x[2n] + i x[2n+1]: exactly the real output, read as complex
    int rc = emit_axis(z, out, H, 1, lines, true, scale, err);
    if (rc) return rc;
    ir.route += "c2r-split ";
    return MI355FFT_OK;
  }

  // Lane layouts (channel-lane presets, whdcn with unit stride along the line): contiguous power-of-two lines that sit at
  // arbitrary pitches on either side need no gather / scatter pass — the ROW line kernels take the two pitches as they are.
  // rank-1 view of a four-step line (ioView / zeroPad / unit-stride lanes) on a VIEW instance of the fused kernels: the maps' ranges are
  // predicates of pass A's loads and pass B's stores — no embed / zero / extract launch.  false: no instance for this length.
  bool emit_xcd_view(PtrRef in, PtrRef out, int64_t N, int64_t lines, bool inverse, float scale, const SideMap& im, const SideMap& om) {
    if (opt.force_generic || opt.xcd_fused != 1 || !opt.xcd_shared || !opt.fuse_views || opt.only_pass) return false;
    const XcdKernelMeta* xm = nullptr;
    for (const auto& m : xcd_kernel_registry()) if (m.rt == 6 && (int64_t)m.N1 * m.N2 == N && m.inverse == inverse) xm = &m;
    if (!xm) return false;
    int64_t split = 1, slots = 1;
    xcd_groups((uint64_t)N * 8, false, split, slots);
    const PtrRef wslots = alloc_work((uint64_t)(16 * slots * split) * N * 8), ctl = alloc_work(40960);
    const int shift = N >= (1 << 20) ? 10 : lg2(N) / 2;                // LO table of 2^shift roots, HI of N >> shift (as emit_axis)
    std::vector<float2h> lo((size_t)1 << shift), hi((size_t)(N >> shift));
    for (size_t l = 0; l < lo.size(); ++l) lo[l] = root_of_unity((int64_t)l, N);
    for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << shift, N);
    const LineKernelMeta ml = make_meta(0, xm->N1, xm->ra[0], xm->ra[1], xm->ra[2], xm->ta, true, true, false, false, 0);
    const LineKernelMeta mr = make_meta(0, xm->N2, xm->rb[0], xm->rb[1], xm->rb[2], xm->tb, false, true, false, false, 0);
    const PtrRef ta = line_tables(ml), tb = line_tables(mr), tlo = add_table(lo), thi = add_table(hi);
    { Step& z = push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
    Step& st = push(ST_XCD_FUSED);
    st.variant = xm->id;
    st.p[0] = in.plus(im.offset * 8); st.p[1] = out.plus(om.offset * 8); st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
    st.i[0] = lines; st.i[1] = N; st.i[2] = shift; st.i[3] = ((int64_t)1 << shift) - 1; st.i[9] = im.batch_stride; st.i[10] = om.batch_stride;
    st.i[4] = ta.off; st.i[5] = tb.off; st.i[6] = tlo.off; st.i[7] = thi.off; st.i[8] = split; st.i[11] = slots; st.i[12] = 0; st.i[13] = opt.xcd_spin_limit;
    st.f[0] = scale;
    st.imap = im; st.omap = om;
    st.grid = (unsigned)(opt.compute_units * ((xm->threads <= 256 && xm->lds_bytes <= 80 * 1024) ? 2 : 1));
    ir.route += "xcd-fused-view[N=" + std::to_string(xm->N1) + "x" + std::to_string(xm->N2) + "] ";
    return true;
  }
  bool emit_lines_pitched(PtrRef src, PtrRef dst, int64_t N, int64_t lines, bool inverse, float scale, int64_t in_pitch, int64_t out_pitch) {
    if (opt.force_generic || !is_pow2(N) || N < 2 || N > opt.max_line || (opt.xcd_fused == 2 && N == 4096)) return false;
    const LineKernelMeta* m = find_line_kernel((int)N, false, false, inverse, inverse, 0);
    if (!m) return false;
    Step& st = push(ST_LINES);
    st.variant = m->id;
    st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*m);
    const int64_t tiles = (lines + m->T - 1) / m->T;
    st.i[0] = tiles; st.i[1] = lines; st.i[2] = 1; st.i[3] = in_pitch; st.i[4] = 1; st.i[5] = out_pitch;
    st.f[0] = scale;
    st.grid = lines_grid(*m, tiles, true);
    ir.route += "lines[N=" + std::to_string(N) + ",pitch=" + std::to_string(in_pitch) + "/" + std::to_string(out_pitch) + "] ";
    return true;
  }

  // ---- mapped sides: strided layouts, ioView and zeroPad carried by the first load / last store -------
  // Can the pass over an axis of length N with element stride S be a line-kernel launch (ROW for S == 1, column tiles above)?
  bool axis_mappable(int64_t N, int64_t S, bool inverse) const {
    if (opt.force_generic || !opt.fuse_views || !is_pow2(N) || N < 2) return false;
    if (S == 1) return N <= opt.max_line && !(opt.xcd_fused == 2 && N == 4096) && find_line_kernel((int)N, false, false, inverse, inverse, 0) != nullptr;
    return find_line_kernel((int)N, true, true, inverse, inverse, 0) != nullptr;
  }
  static SideMap dense_map(const int64_t* shape, int rank) {
    SideMap m;
    m.rank = rank;
    int64_t st = 1;
    for (int d = 0; d < rank; ++d) { m.dims[d] = (int)shape[d]; m.stride[d] = st; m.lo[d] = m.zlo[d] = 0; m.hi[d] = m.zhi[d] = (int)shape[d]; st *= shape[d]; }
    m.batch_stride = st;
    return m;
  }
  // first and last axis emit_nd will transform (-1: none)
  static void nd_first_last(const int64_t* shape, int rank, uint32_t axes_mask, int& first, int& last) {
    first = last = -1;
    for (int a = 0; a < rank; ++a)
      if (shape[a] > 1 && (axes_mask == 0 || ((axes_mask >> a) & 1u))) { if (first < 0) first = a; last = a; }
  }
  // one line-kernel launch over axis `ax` with both sides given as maps (kern_lines.hpp fft_lines_mapped_kernel)
  int emit_axis_mapped(PtrRef src, PtrRef dst, int64_t N, int64_t S, int64_t outer, bool inverse, float scale, SideMap im, SideMap om, int ax) {
    const bool col = S > 1;
    const LineKernelMeta* m = find_line_kernel((int)N, col, col, inverse, inverse, 0);
    if (!m) return MI355FFT_ERR_UNSUPPORTED;
    const int64_t lines = S * outer, tiles = (lines + m->T - 1) / m->T;
    im.ax = om.ax = ax;
    Step& st = push(ST_LINES);
    st.variant = m->id;
    st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*m);
    st.i[0] = tiles; st.i[1] = lines; st.i[2] = S; st.i[3] = S * N; st.i[4] = S; st.i[5] = S * N; st.i[10] = 1;
    st.f[0] = scale;
    st.imap = im; st.omap = om;
    st.grid = lines_grid(*m, tiles);
    ir.route += std::string(col ? "columns-mapped[N=" : "lines-mapped[N=") + std::to_string(N) + (col ? ",S=" + std::to_string(S) : "") + "] ";
    return MI355FFT_OK;
  }

  int emit_axis(PtrRef src, PtrRef dst, int64_t N, int64_t S, int64_t outer, bool inverse, float scale, std::string& err) {
    const int64_t lines = S * outer;
    // work buffers taken inside one axis transform are temporaries: released on every exit
    struct Scope { uint64_t& top; uint64_t mark; ~Scope() { top = mark; } } scope{work_top, work_top};
    if (N == 1) {
      if (!src.same(dst)) { Step& c = push(ST_COPY); c.p[0] = src; c.p[1] = dst; c.i[0] = lines * 8; }
      if (scale != 1.0f) { Step& s = push(ST_SCALE); s.p[0] = dst; s.i[0] = lines * 2; s.f[0] = scale; s.grid = generic_grid(lines * 2); }
      return MI355FFT_OK;
    }
    const bool p2 = is_pow2(N);
    // N = 2^15 and 2^13 (line32k >= 1; 2^14 too with line32k == 2: measured equal to its ROW kernel), dense lines: the whole line in the
    // registers of one workgroup of N/64 threads, exchanges through LDS in halves (kern_line32k.hpp, kern_line_reg.hpp).  2^15: one HBM
    // round trip where the solo four-step makes two (288 vs 210 GPoints/s); 2^13: four 128-thread workgroups per CU instead of two
    // 256-thread ones with the line in LDS (330 vs 288)
    if (!opt.force_generic && opt.max_line >= 16384 && S == 1 && !opt.only_pass && opt.xcd_fused != 2 &&
        ((opt.line32k >= 1 && (N == 32768 || N == 8192)) || (opt.line32k == 2 && (N == 16384 || N == 4096)))) {
      const int lg = lg2(N);
      const int R0 = 32, R1 = N <= 8192 ? 16 : 32;
      std::vector<float2h> t;
      for (int q = 1; q < R1; ++q) for (int k = 0; k < R0; ++k) t.push_back(root_of_unity((int64_t)q * k, (int64_t)R0 * R1));
      for (int64_t l = 0; l < 1024; ++l) t.push_back(root_of_unity(l, N));
      for (int64_t h = 0; h < N / 1024; ++h) t.push_back(root_of_unity(h << 10, N));
      Step& st = push(ST_LINES_MIXED);
      st.variant = 1000 + lg;
      st.p[0] = src; st.p[1] = dst; st.p[2] = add_table(t);
      st.i[0] = lines; st.i[1] = N; st.i[2] = 1; st.i[3] = 1; st.i[4] = 3;
      st.i[5] = inverse ? 1 : 0; st.i[7] = N / 64;
      st.f[0] = scale;
      const int64_t per_cu = N == 32768 ? 1 : (N == 16384 ? 2 : (N == 8192 ? 4 : 8));
      st.grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(lines, (int64_t)opt.compute_units * per_cu));
      ir.route += (N == 32768 ? std::string("line32k[N=32768] ") : "line-reg[N=" + std::to_string(N) + "] ");
      return MI355FFT_OK;
    }
    if (!opt.force_generic && S == 1 && p2 && N <= opt.max_line && !(opt.xcd_fused == 2 && N == 4096)) {   // xcd_fused == 2: emulation tests
      const LineKernelMeta* m = find_line_kernel((int)N, false, false, inverse, inverse, 0);
      if (m) {
        Step& st = push(ST_LINES);
        st.variant = m->id;
        st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*m);
        const int64_t tiles = (lines + m->T - 1) / m->T;
        st.i[0] = tiles; st.i[1] = lines; st.i[2] = 1; st.i[3] = N; st.i[4] = 1; st.i[5] = N;
        st.f[0] = scale;
        st.grid = lines_grid(*m, tiles, true);
        ir.route += "lines[N=" + std::to_string(N) + "] ";
        return MI355FFT_OK;
      }
    }
    // the column line kernels address a tile with 32-bit element offsets (kern_lines.hpp: voff = idx * S): a strided axis whose
    // plane spans 2^32 elements or more (32 GiB of complex data) stays on the stage route, which indexes with 64 bits
    const bool col_span_ok = (uint64_t)N * (uint64_t)S < (1ull << 32);
    if (!opt.force_generic && S > 1 && p2 && col_span_ok) {
      // an axis with stride S > 1 (N-D transforms): T adjacent lines form a column tile
      const LineKernelMeta* m = find_line_kernel((int)N, true, true, inverse, inverse, 0);
      if (m && S % m->T == 0) {
        Step& st = push(ST_LINES);
        st.variant = m->id;
        st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*m);
        const int64_t tiles = lines / m->T;
        st.i[0] = tiles; st.i[1] = lines; st.i[2] = S; st.i[3] = S * N; st.i[4] = S; st.i[5] = S * N;
        st.f[0] = scale;
        st.grid = lines_grid(*m, tiles);
        ir.route += "columns[N=" + std::to_string(N) + ",S=" + std::to_string(S) + "] ";
        return MI355FFT_OK;
      }
      // S not a multiple of the tile width (the packed axis 0 of an N-D r2c: S = N0/2 + 1): per-group tiles, ragged last one
      const LineKernelMeta* mr = find_line_kernel((int)N, true, true, inverse, inverse, 3);
      if (mr && S >= mr->T) {
        Step& st = push(ST_LINES);
        st.variant = mr->id;
        st.p[0] = src; st.p[1] = dst; st.p[2] = line_tables(*mr);
        const int64_t tpg = (S + mr->T - 1) / mr->T, tiles = outer * tpg;
        st.i[0] = tiles; st.i[1] = lines; st.i[2] = S; st.i[3] = S * N; st.i[4] = S; st.i[5] = S * N; st.i[8] = tpg;
        st.f[0] = scale;
        st.grid = lines_grid(*mr, tiles);
        ir.route += "columns-ragged[N=" + std::to_string(N) + ",S=" + std::to_string(S) + "] ";
        return MI355FFT_OK;
      }
    }
    if (!opt.force_generic && S == 1 && N == (1 << 20) && opt.xcd_res && opt.xcd_shared && !opt.only_pass && opt.compute_units % 32 == 0) {
      // XCD-resident route (kern_xcd_res.hpp): the transform stays in the registers and LDS of one XCD's 32 workgroups between
      // its passes; hand-offs go through a 4 MiB L2-resident exchange buffer per XCD.  One workgroup per CU, all co-resident.
      const int shift = 10;
      std::vector<float2h> lo((size_t)1 << shift), hi((size_t)(N >> shift));
      for (size_t l = 0; l < lo.size(); ++l) lo[l] = root_of_unity((int64_t)l, N);
      for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << shift, N);
      const LineKernelMeta mt = make_meta(0, 1024, 32, 32, 1, 16, true, true, false, false, 0);
      const PtrRef ta = line_tables(mt), tlo = add_table(lo), thi = add_table(hi);
      const PtrRef wslots = alloc_work((uint64_t)16 * 4 * (1 << 20));   // 4 channels of 1 MiB per XCC id (16 ids)
      const PtrRef ctl = alloc_work(40960);
      { Step& z = push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
      Step& st = push(ST_XCD_RES);
      st.variant = (inverse ? 1 : 0) + (opt.xcd_res == 2 || opt.xcd_res == 4 ? 2 : 0) + (opt.xcd_res >= 3 && !inverse ? 4 : 0);   // 3: stamps, 4: stamps on the skeleton
      st.p[0] = src; st.p[1] = dst; st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
      st.i[0] = lines; st.i[1] = N; st.i[2] = shift; st.i[3] = ((int64_t)1 << shift) - 1; st.i[9] = N; st.i[10] = N;
      st.i[4] = ta.off; st.i[5] = ta.off; st.i[6] = tlo.off; st.i[7] = thi.off; st.i[8] = opt.xcd_res_depth; st.i[11] = 1; st.i[12] = 0; st.i[13] = opt.xcd_spin_limit;
      st.f[0] = scale;
      st.grid = (unsigned)opt.compute_units;
      ir.route += "xcd-resident[N=1024x1024,depth=" + std::to_string(opt.xcd_res_depth) + (opt.xcd_res == 2 || opt.xcd_res == 4 ? ",skeleton" : "") + (opt.xcd_res >= 3 ? ",stamps" : "") + "] ";
      return MI355FFT_OK;
    }
    if (!opt.force_generic && S == 1 && p2 && N >= 4096 && opt.xcd_fused && !opt.only_pass) {
      // XCD-fused route: both passes in one persistent launch, one transform per XCD at a time (kern_xcd.hpp)
      const int lgf = lg2(N);
      const int64_t F1 = (int64_t)1 << (lgf / 2), F2 = N / F1;
      const XcdKernelMeta* xm = nullptr;
      for (const auto& m : xcd_kernel_registry())
        if (!m.real && m.N1 == F1 && m.N2 == F2 && m.inverse == inverse && (!m.rt || ((m.rt == 2 ? opt.xcd_hx == 1 : m.rt == 3 ? opt.xcd_hx == 2 : m.rt == 5 ? opt.xcd_hx == 3 : m.rt == 6 || m.rt == 7 ? false : opt.xcd_rt != 0) && opt.xcd_shared))) xm = &m;
      // c2c 2^21: with two groups per XCD and one slot each the LDS-resident 1024 x 2048 instance (8-line tiles taken in pairs) runs at 177
      // GPoints/s, ahead of both register-tile forms — 2048 x 1024 (16-line tiles down the columns, 32-line tiles along the rows;
      // MI355FFT_XCD_RT=3) 172, 1024 x 2048 (MI355FFT_XCD_RT=2) 162: profiles/r03_regtile_ab.log.  So the register tiles serve 2^22 only.
      if (N == ((int64_t)1 << 21) && opt.xcd_rt != 2) {
        xm = nullptr;
        for (const auto& m : xcd_kernel_registry())
          if (!m.real && m.inverse == inverse && ((opt.xcd_rt == 3 && opt.xcd_shared) ? m.rt == 7 : (!m.rt && m.N1 == F1 && m.N2 == F2))) xm = &m;
      }
      if (xm && (N > 4096 || opt.xcd_fused == 2) &&
          (opt.xcd_shared || ((uint64_t)N * 8 <= ((uint64_t)opt.solo_max_kb << 10) && opt.xcd_fused != 2))) {
        const bool a_rt = xm->rt == 1 && xm->N1 == 2048;   // register-tile passes take their stage-2 table instead of a line kernel's
        const LineKernelMeta ma = make_meta(0, a_rt ? 1024 : xm->N1, xm->ra[0], a_rt ? 32 : xm->ra[1], xm->ra[2], xm->ta, true, true, false, false, 0);
        const LineKernelMeta mb = make_meta(0, xm->rt == 1 ? 1024 : xm->N2, xm->rt == 1 ? 32 : xm->rb[0], xm->rb[1], xm->rb[2], xm->tb, false, true, false, false, 0);
        // transforms of at most 1 MiB: every workgroup walks whole transforms alone ("solo": no registration, no cross-
        // workgroup barrier, so no co-residency requirement and as many workgroups per CU as fit); larger ones are shared by
        // the groups of an XCD
        const bool solo = (uint64_t)N * 8 <= ((uint64_t)opt.solo_max_kb << 10) && opt.xcd_fused != 2;
        int64_t split = 1, grid = opt.compute_units, slots = 1;
        PtrRef wslots, ctl;
        if (solo) {
          const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((160 * 1024) / xm->lds_bytes, 2048 / xm->threads), 4));
          grid = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)opt.compute_units * per_cu, ((int64_t)opt.solo_cap_mb << 20) / (N * 8)), lines));
          slots = 1;
          wslots = alloc_work((uint64_t)grid * N * 8);          // one slot per workgroup, all of them within the Infinity Cache
          ctl = alloc_work(256);
        } else {
          xcd_groups((uint64_t)N * 8, xm->rt == 1, split, slots);
          wslots = alloc_work((uint64_t)(16 * slots * split) * N * 8);   // slots per group x `split` groups per XCC id (16 ids)
          ctl = alloc_work(40960);
        }
        const int shift = N >= (1 << 20) ? 10 : lgf / 2;                // LO table of 2^shift roots, HI of N >> shift
        std::vector<float2h> lo((size_t)1 << shift), hi((size_t)std::max<int64_t>(1, N >> shift));
        for (size_t l = 0; l < lo.size(); ++l) lo[l] = root_of_unity((int64_t)l, N);
        for (size_t h = 0; h < hi.size(); ++h) hi[h] = root_of_unity((int64_t)h << shift, N);
        const PtrRef tb = xm->rt == 1 ? regtile_table() : line_tables(mb), ta = a_rt ? tb : xm->rt == 7 ? regtile_table() : line_tables(ma), tlo = add_table(lo), thi = add_table(hi);
        if (!solo) { Step& z = push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
        Step& st = push(ST_XCD_FUSED);
        st.variant = xm->id;
        st.p[0] = src; st.p[1] = dst; st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
        st.i[0] = lines; st.i[1] = N; st.i[2] = shift; st.i[3] = ((int64_t)1 << shift) - 1;
        st.i[9] = lane_in_pitch ? lane_in_pitch : N; st.i[10] = lane_out_pitch ? lane_out_pitch : N;
        if (lane_in_pitch || lane_out_pitch) lane_used = true;
        st.i[4] = ta.off; st.i[5] = tb.off; st.i[6] = tlo.off; st.i[7] = thi.off; st.i[8] = split; st.i[11] = slots; st.i[12] = solo ? 1 : 0; st.i[13] = opt.xcd_spin_limit;
        st.f[0] = scale;
        // shared mode: every workgroup must be co-resident — one per CU, two where 256 threads and <= 80 KB of LDS leave room
        if (!solo && opt.xcd_fused != 2 && ((xm->threads <= 256 && xm->lds_bytes <= 80 * 1024) || xm->rt == 2)) grid *= 2;
        st.grid = (unsigned)grid;
        ir.route += std::string(solo ? "xcd-solo[N=" : xm->rt == 2 ? "xcd-fused-2wg[N=" : xm->rt == 3 ? "xcd-fused-rt32[N=" : xm->rt == 5 ? "xcd-fused-rt16x2[N=" : xm->rt == 7 ? "xcd-fused-rt32[N=" : xm->rt ? "xcd-fused-rt[N=" : "xcd-fused[N=") + std::to_string(xm->N1) + "x" + std::to_string(xm->N2) + "] ";
        return MI355FFT_OK;
      }
    }
    if (!opt.force_generic && S == 1 && p2 && N > 4096) {
      const int lg = lg2(N);
      const int64_t N1 = (int64_t)1 << (lg / 2), N2 = N / N1;
      const LineKernelMeta* ma = find_line_kernel((int)N1, true, true, inverse, false, 0);
      const LineKernelMeta* mb = find_line_kernel((int)N2, false, true, false, inverse, 2);
      if (ma && mb && N2 % ma->T == 0 && N1 % mb->T == 0) {
        int64_t chunk = (int64_t)(opt.chunk_bytes / (uint64_t)(N * 8));
        chunk = std::max<int64_t>(1, std::min(chunk, lines));
        const PtrRef w = alloc_work((uint64_t)chunk * N * 8);
        const PtrRef ta = line_tables(*ma), tb = line_tables(*mb);
        // four-step roots e^{-2 pi i m/N}, m = n2*k1 < N, as HI[m >> 10] * LO[m & 1023]
        std::vector<float2h> lo(1024), hi((size_t)(N >> 10));
        for (int64_t l = 0; l < 1024; ++l) lo[(size_t)l] = root_of_unity(l, N);
        for (int64_t h = 0; h < (N >> 10); ++h) hi[(size_t)h] = root_of_unity(h << 10, N);
        const PtrRef tlo = add_table(lo), thi = add_table(hi);
        for (int64_t t0 = 0; t0 < lines; t0 += chunk) {
          const int64_t c = std::min(chunk, lines - t0);
          if (opt.only_pass != 2) {
          Step& a = push(ST_LINES);
          a.variant = ma->id;
          a.p[0] = src.plus(t0 * N * 8); a.p[1] = w; a.p[2] = ta;
          a.i[0] = c * N2 / ma->T; a.i[1] = c * N2; a.i[2] = N2; a.i[3] = N; a.i[4] = N2; a.i[5] = N;
          a.f[0] = 1.0f;
          a.grid = lines_grid(*ma, a.i[0]);
          }
          if (opt.only_pass == 1) continue;
          Step& b = push(ST_LINES);
          b.variant = mb->id;
          b.p[0] = w; b.p[1] = dst.plus(t0 * N * 8); b.p[2] = tb; b.p[3] = tlo; b.p[4] = thi;
          b.i[0] = c * N1 / mb->T; b.i[1] = c * N1; b.i[2] = 1; b.i[3] = N2; b.i[4] = N1; b.i[5] = N; b.i[6] = 10; b.i[7] = 1023;
          b.f[0] = scale;
          b.grid = lines_grid(*mb, b.i[0]);
          // four-step roots e^{-2 pi i k1 n2/N} enter at pass B's loads; with a grid that is a multiple of the
          // tiles per transform every workgroup keeps the same rows k1 and computes its roots once per launch
          const int64_t tpt = N1 / mb->T;
          if ((int64_t)b.grid >= tpt) b.grid = (unsigned)((b.grid / tpt) * tpt);
          b.i[8] = N1;
        }
        ir.route += "two-pass[N=" + std::to_string(N1) + "x" + std::to_string(N2) + ",chunk=" + std::to_string(chunk) + "] ";
        return MI355FFT_OK;
      }
    }
    // generic: one global-memory Stockham stage per radix
    const std::vector<int> radices = factorize_radices(N);
    if (radices.empty()) return emit_bluestein(src, dst, N, S, outer, inverse, scale, err);
    int ns = (int)radices.size();
    const std::vector<int> lds_radices = factorize_radices(N, 8);
    // (short lines that the stage route finishes in three passes with a radix-16/32 first stage measured faster there: 640, 768)
    // (strided axes need T >= 8 adjacent lines per workgroup for coalesced accesses: N <= 512; longer ones keep the stage route —
    // measured 27 GPoints/s for the 4096-point axis of a 4096x4096 array with T = 1)
    // dense lines of a length with a compile-time-plan instance (kern_mixed_ct.hpp)
    if (opt.mixed_lines && opt.mixed_ct && !opt.force_generic && S == 1) {
      const MixedCtMeta* cm = nullptr;
      for (const auto& m : mixedct_registry()) if (m.N == N && (!cm || opt.mixed_ct != 2)) cm = &m;
      if (cm) {
        std::vector<float2h> t;
        Step& st = push(ST_LINES_MIXED);
        int64_t nsp = 1;
        for (int R : cm->radices) {
          const size_t off = t.size();
          t.resize(off + (size_t)(R * nsp));
          for (int q = 0; q < R; ++q) for (int64_t k = 0; k < nsp; ++k) t[off + (size_t)(q * nsp + k)] = root_of_unity(q * k, nsp * R);
          nsp *= R;
        }
        st.variant = cm->id + 1;
        st.p[0] = src; st.p[1] = dst; st.p[2] = add_table(t);
        st.i[0] = lines; st.i[1] = N; st.i[2] = 1; st.i[3] = cm->T; st.i[4] = (int64_t)cm->radices.size();
        st.i[5] = inverse ? 1 : 0; st.i[7] = cm->threads;
        st.f[0] = scale;
        const int64_t tiles = (lines + cm->T - 1) / cm->T;
        const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((160 * 1024) / (cm->lds_bytes + 1024), 2048 / cm->threads), 8));
        st.grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)opt.compute_units * per_cu));
        ir.route += "mixed-ct[N=" + std::to_string(N) + ",T=" + std::to_string(cm->T) + ",n=" + std::to_string(cm->radices.size()) + "] ";
        return MI355FFT_OK;
      }
    }
    const bool stages_win = ((S == 1 && N < 1024 && radices.size() == 3 && radices[0] >= 16) || (S > 1 && N > 512)) && opt.mixed_lines != 2;
    if (opt.mixed_lines && !opt.force_generic && !stages_win && lds_radices.size() >= 2 && lds_radices.size() <= 12 && N <= 4096) {
      const std::vector<int>& radices = lds_radices;
      ns = (int)radices.size();
      // all stages in one launch, T lines per workgroup in LDS (kern_mixed.hpp)
      std::vector<float2h> t;
      Step& st = push(ST_LINES_MIXED);
      int64_t nsp = 1;
      for (int s = 0; s < ns; ++s) {
        const int R = radices[s];
        st.i[8 + s] = ((int64_t)R << 32) | (int64_t)t.size();       // radix, table offset (elements)
        const size_t off = t.size();
        t.resize(off + (size_t)(R * nsp));
        for (int q = 0; q < R; ++q) for (int64_t k = 0; k < nsp; ++k) t[off + (size_t)(q * nsp + k)] = root_of_unity(q * k, nsp * R);
        nsp *= R;
      }
      // Tile and workgroup shape from the sweeps in profiles/r01_mixed_radix.log: short lines fill a 64 KB pair of buffers
      // (T lines, 256 threads); from N = 512 on one line per workgroup, threads ~ N/8, and as many workgroups per CU as fit.
      // The stage tables ride in LDS when they are small (<= 16 KB); longer ones are read through the caches.
      int64_t T, threads;
      if (opt.mixed_lds_kb > 0) { T = std::max<int64_t>(1, std::min<int64_t>(opt.mixed_lds_kb * 64 / N, 64)); threads = opt.mixed_threads; }
      else if (N < 512 || S > 1) { T = std::max<int64_t>(1, std::min<int64_t>(4096 / N, 64)); threads = 256; }   // strided axes: lanes walk T lines
      else { T = 1; threads = std::max<int64_t>(64, std::min<int64_t>(256, ((N / 8 + 63) / 64) * 64)); }
      const int64_t lds_bytes = 2 * T * (N + (N >> 5) + 1) * 8;          // kern_mixed.hpp mixed_pitch
      const int64_t tw_lds = (int64_t)t.size() * 8 <= 16 * 1024 ? (int64_t)t.size() : 0;
      st.p[0] = src; st.p[1] = dst; st.p[2] = add_table(t);
      st.i[0] = lines; st.i[1] = N; st.i[2] = S; st.i[3] = T; st.i[4] = ns;
      st.i[5] = inverse ? 1 : 0; st.i[6] = lds_bytes; st.i[7] = threads; st.i[19] = tw_lds;
      st.f[0] = scale;
      const int64_t tiles = (lines + T - 1) / T;
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((160 * 1024) / (lds_bytes + tw_lds * 8 + 1024), 2048 / threads), 8));
      st.grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)opt.compute_units * per_cu));
      ir.route += "mixed-lines[N=" + std::to_string(N) + ",S=" + std::to_string(S) + ",T=" + std::to_string(T) + ",n=" + std::to_string(ns) + "] ";
      return MI355FFT_OK;
    }
    PtrRef w0, w1;
    if (ns >= 2) w0 = alloc_work((uint64_t)lines * N * 8);
    if (ns >= 3) w1 = alloc_work((uint64_t)lines * N * 8);
    int64_t nsp = 1;
    PtrRef cur = src;
    for (int s = 0; s < ns; ++s) {
      const int R = radices[s];
      const bool last = s == ns - 1;
      PtrRef to = last ? dst : ((s % 2 == 0) ? w0 : w1);
      std::vector<float2h> t;
      if (nsp > 1) {
        t.resize((size_t)(R * nsp));
        for (int q = 0; q < R; ++q) for (int64_t k = 0; k < nsp; ++k) t[(size_t)(q * nsp + k)] = root_of_unity(q * k, nsp * R);
      } else t.push_back(float2h{1, 0});
      Step& st = push(ST_STAGE);
      st.variant = R;
      st.p[0] = cur; st.p[1] = to; st.p[2] = add_table(t);
      st.i[0] = lines * (N / R); st.i[1] = N; st.i[2] = S; st.i[3] = nsp;
      st.i[4] = (inverse && s == 0) ? 1 : 0; st.i[5] = (inverse && last) ? 1 : 0;
      st.f[0] = last ? scale : 1.0f;
      st.grid = generic_grid(st.i[0]);
      cur = to;
      nsp *= R;
    }
    ir.route += "stages[N=" + std::to_string(N) + ",S=" + std::to_string(S) + ",n=" + std::to_string(ns) + "] ";
    return MI355FFT_OK;
  }

  // Lengths with a prime factor > 13 (large_policy.js:214-228 picks Bluestein/Rader for those): chirp-z through a
  // power-of-two circular convolution of length M >= 2N-1 on the fast routes.  Axes with stride S > 1 are
  // transposed to contiguous lines first (two extra passes; a completeness route, not a tuned one).
  int emit_bluestein(PtrRef src, PtrRef dst, int64_t N, int64_t S, int64_t outer, bool inverse, float scale, std::string& err) {
    const int64_t lines = S * outer;
    if (N > ((int64_t)1 << 22)) { err = "Unsupported: Bluestein axis length " + std::to_string(N) + " exceeds 2^22"; return MI355FFT_ERR_UNSUPPORTED; }
    int64_t M = 1;
    while (M < 2 * N - 1) M <<= 1;
    // chirp a[n] = e^{-i pi n^2/N} = root(n^2 mod 2N, 2N);  B = FFT_M(b), b[n] = conj(a[n]) wrapped (b[M-n] = b[n])
    std::vector<float2h> chirp((size_t)N);
    std::vector<double> br((size_t)M, 0.0), bi((size_t)M, 0.0);
    for (int64_t n = 0; n < N; ++n) {
      const int64_t m = (int64_t)(((unsigned __int128)n * (unsigned __int128)n) % (unsigned __int128)(2 * N));
      const long double ang = -3.14159265358979323846264338327950288L * (long double)m / (long double)N;
      chirp[(size_t)n] = root_of_unity(m, 2 * N);
      const double cr = (double)cosl(ang), ci = (double)sinl(ang);
      br[(size_t)n] = cr; bi[(size_t)n] = -ci;
      if (n > 0) { br[(size_t)(M - n)] = cr; bi[(size_t)(M - n)] = -ci; }
    }
    host_fft_pow2(br, bi);
    std::vector<float2h> bt((size_t)M);
    for (int64_t m = 0; m < M; ++m) bt[(size_t)m] = float2h{(float)br[(size_t)m], (float)bi[(size_t)m]};
    const PtrRef tchirp = add_table(chirp), tb = add_table(bt);
    PtrRef lin_src = src, lin_dst = dst;
    PtrRef tr;   // dense [outer][S][N] copy of a strided axis
    if (S > 1) {
      tr = alloc_work((uint64_t)lines * N * 8);
      Step& g = push(ST_GATHER);     // transpose [outer][N][S] -> [outer][S][N]
      g.p[0] = src; g.p[1] = tr;
      g.i[0] = lines * N; g.i[1] = S * N; g.i[2] = 2; g.i[3] = 0; g.i[4] = S * N; g.i[5] = 0; g.i[6] = S * N;
      g.shape[0] = N; g.shape[1] = S; g.sa[0] = S; g.sa[1] = 1; g.sb[0] = 1; g.sb[1] = N;
      g.grid = generic_grid(g.i[0]);
      lin_src = tr; lin_dst = tr;
    }
    const PtrRef y = alloc_work((uint64_t)lines * M * 8);
    // r02: with the convolution length on a line kernel the five launches become two — the chirp and the zero-padded embed ride the
    // first-stage loads of the forward launch, the product with the chirp's spectrum its last-stage store (fft_lines_mul_kernel<C, MAPPED>),
    // and the inverse launch multiplies by the chirp and crops to N points in its store pass (fft_lines_mapped_kernel)
    {
      const LineKernelMeta* mf = nullptr;
      if (opt.fuse_views && opt.conv_lines && !opt.force_generic && S == 1 && M >= 64 && M <= opt.max_line && M * 2 < ((int64_t)1 << 31) &&
          !(opt.xcd_fused == 2 && M == 4096))
        mf = find_line_kernel((int)M, false, false, false, false, 0);
      const LineKernelMeta* mi = mf ? find_line_kernel((int)M, false, false, true, true, 0) : nullptr;
      if (mf && mi && mf->lds_bytes > 0 && mf->R1 > 1) {
        const int64_t one[1] = {M};
        SideMap win = dense_map(one, 1);          // N live points of an M-point line, lines N apart
        win.hi[0] = (int)N; win.batch_stride = N; win.ax = 0;
        const SideMap full = dense_map(one, 1);
        const int64_t flags = 1 | (inverse ? 2 : 0);
        Step& f = push(ST_LINES);
        f.variant = mf->id;
        f.p[0] = lin_src; f.p[1] = y; f.p[2] = line_tables(*mf); f.p[3] = tb; f.p[4] = tchirp;
        int64_t tiles = (lines + mf->T - 1) / mf->T;
        f.i[0] = tiles; f.i[1] = lines; f.i[2] = 1; f.i[3] = M; f.i[4] = 1; f.i[5] = M; f.i[6] = 0; f.i[7] = flags; f.i[9] = 4; f.i[10] = 1;
        f.f[0] = 1.0f;
        f.imap = win; f.omap = full;
        f.grid = lines_grid(*mf, tiles);
        Step& g = push(ST_LINES);
        g.variant = mi->id;
        g.p[0] = y; g.p[1] = lin_dst; g.p[2] = line_tables(*mi); g.p[4] = tchirp;
        tiles = (lines + mi->T - 1) / mi->T;
        g.i[0] = tiles; g.i[1] = lines; g.i[2] = 1; g.i[3] = M; g.i[4] = 1; g.i[5] = M; g.i[7] = flags; g.i[10] = 1;
        g.f[0] = (float)((double)scale / (double)M);
        g.imap = full; g.omap = win;
        g.grid = lines_grid(*mi, tiles);
        ir.route += "bluestein-lines[N=" + std::to_string(N) + ",M=" + std::to_string(M) + "] ";
        return MI355FFT_OK;
      }
    }
    Step& pre = push(ST_CHIRP_PRE);
    pre.p[0] = lin_src; pre.p[1] = y; pre.p[2] = tchirp;
    pre.i[0] = N; pre.i[1] = M; pre.i[2] = lines; pre.i[3] = inverse ? 1 : 0; pre.i[4] = 0;
    pre.grid = generic_grid(lines * M);
    int rc = emit_axis(y, y, M, 1, lines, false, 1.0f, err);
    if (rc) return rc;
    Step& pm = push(ST_POINTWISE);
    pm.p[0] = y; pm.p[1] = y; pm.p[2] = tb;
    pm.i[0] = M; pm.i[1] = lines * M; pm.i[2] = 0; pm.f[0] = 1.0f;
    pm.grid = generic_grid(lines * M);
    rc = emit_axis(y, y, M, 1, lines, true, 1.0f, err);
    if (rc) return rc;
    Step& post = push(ST_CHIRP_POST);
    post.p[0] = y; post.p[1] = lin_dst; post.p[2] = tchirp;
    post.i[0] = N; post.i[1] = M; post.i[2] = lines; post.i[3] = 0; post.i[4] = inverse ? 1 : 0;
    post.f[0] = (float)((double)scale / (double)M);
    post.grid = generic_grid(lines * N);
    if (S > 1) {
      Step& sc = push(ST_SCATTER);   // back to [outer][N][S]
      sc.p[0] = tr; sc.p[1] = dst;
      sc.i[0] = lines * N; sc.i[1] = S * N; sc.i[2] = 2; sc.i[3] = 0; sc.i[4] = S * N; sc.i[5] = 0; sc.i[6] = S * N;
      sc.shape[0] = N; sc.shape[1] = S; sc.sa[0] = S; sc.sa[1] = 1; sc.sb[0] = 1; sc.sb[1] = N;
      sc.grid = generic_grid(sc.i[0]);
    }
    ir.route += "bluestein[N=" + std::to_string(N) + ",M=" + std::to_string(M) + "] ";
    return MI355FFT_OK;
  }

  // in-place radix-2 FFT in f64 (plan-time tables only)
  static void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
      size_t bit = n >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
      const size_t half = len >> 1;
      for (size_t k = 0; k < half; ++k) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)len;
        const double wr = (double)cosl(ang), wi = (double)sinl(ang);
        for (size_t i = k; i < n; i += len) {
          const size_t j = i + half;
          const double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
          re[j] = re[i] - tr; im[j] = im[i] - ti;
          re[i] += tr; im[i] += ti;
        }
      }
    }
  }

  // all axes of a dense [batch][shape] complex array, src -> dst
  // axes_mask: bit a set => axis a is transformed (createFftPlan({axes})); 0 => every axis from first_axis on
  // imap / omap (optional): the FIRST transformed axis reads `src` through imap, the LAST one writes `fin` through omap; the
  // passes between work on the dense array at dst.  The caller has checked axis_mappable() for those axes.
  int emit_nd(PtrRef src, PtrRef dst, const int64_t* shape, int rank, int64_t batch, bool inverse, float scale, std::string& err,
              int first_axis = 0, uint32_t axes_mask = 0, const SideMap* imap = nullptr, const SideMap* omap = nullptr, PtrRef fin = PtrRef()) {
    if (imap || omap) {
      int fa, la;
      nd_first_last(shape, rank, axes_mask, fa, la);
      const int64_t total = prodv(shape, rank);
      PtrRef cur = src;
      int64_t S = 1;
      for (int a = 0; a < rank; ++a) {
        const int64_t N = shape[a];
        if (N > 1 && (axes_mask == 0 || ((axes_mask >> a) & 1u))) {
          const int64_t outer = batch * (total / (S * N));
          const bool use_i = a == fa && imap, use_o = a == la && omap;
          const PtrRef wr = use_o ? fin : dst;
          int rc;
          if (use_i || use_o) rc = emit_axis_mapped(cur, wr, N, S, outer, inverse, a == la ? scale : 1.0f, use_i ? *imap : dense_map(shape, rank), use_o ? *omap : dense_map(shape, rank), a);
          else rc = emit_axis(cur, wr, N, S, outer, inverse, a == la ? scale : 1.0f, err);
          if (rc) { if (err.empty()) err = "no line kernel for a mapped axis"; return rc; }
          cur = wr;
        }
        S *= N;
      }
      return MI355FFT_OK;
    }
    PtrRef cur = src;
    int64_t S = 1;
    for (int a = 0; a < first_axis; ++a) S *= shape[a];
    const int64_t total = prodv(shape, rank);
    bool any = false;
    int last_axis = -1;
    const auto wanted = [&](int a) { return axes_mask == 0 || ((axes_mask >> a) & 1u); };
    for (int a = first_axis; a < rank; ++a) if (shape[a] > 1 && wanted(a)) last_axis = a;
    int start = first_axis;
    // axes 0 and 1 of square power-of-two planes in ONE persistent launch (kern_xcd.hpp TWO_D): columns, barrier, rows
    if (first_axis == 0 && rank >= 2 && wanted(0) && wanted(1) &&
        emit_xcd_2d(cur, dst, shape[0], shape[1], batch * (total / (shape[0] * shape[1])), inverse, last_axis == 1 ? scale : 1.0f)) {
      cur = dst; any = true; S = shape[0] * shape[1]; start = 2;
    }
    for (int a = start; a < rank; ++a) {
      const int64_t N = shape[a];
      if (N > 1 && wanted(a)) {
        const int64_t outer = batch * (total / (S * N));
        const int rc = emit_axis(cur, dst, N, S, outer, inverse, a == last_axis ? scale : 1.0f, err);
        if (rc) return rc;
        cur = dst;
        any = true;
      }
      S *= N;
    }
    if (!any) return emit_axis(cur, dst, 1, 1, batch * total, inverse, scale, err);
    return MI355FFT_OK;
  }

  void emit_strided(bool gather, PtrRef phys, PtrRef dense, const mi355fft_side_layout& lay, const int64_t* shape, int rank, int64_t batch,
                    const int64_t* dense_shape, const int64_t* dense_sub_offset, int64_t dense_batch_stride, int64_t extra_phys_offset,
                    bool real_elements = false) {
    Step& st = push(gather ? ST_GATHER : ST_SCATTER);
    st.i[7] = real_elements ? 1 : 0;
    st.p[0] = gather ? phys : dense;
    st.p[1] = gather ? dense : phys;
    const int64_t per = prodv(shape, rank);
    st.i[0] = batch * per; st.i[1] = per; st.i[2] = rank;
    int64_t dstride = 1, doff = 0, pstride = 1;
    for (int d = 0; d < rank; ++d) {
      st.shape[d] = shape[d];
      st.sa[d] = lay.strided ? lay.strides[d] : pstride;
      st.sb[d] = dstride;
      doff += (dense_sub_offset ? dense_sub_offset[d] : 0) * dstride;
      dstride *= dense_shape[d];
      pstride *= shape[d];
    }
    st.i[3] = (lay.strided ? lay.offset_elements : 0) + extra_phys_offset;
    st.i[4] = (lay.strided && lay.batch_stride_elements > 0) ? lay.batch_stride_elements : per;
    st.i[5] = doff;
    st.i[6] = dense_batch_stride;
    st.grid = generic_grid(st.i[0]);
  }
};

double scale_factor(int normalize, bool inverse, double n_total) {  // runtime/common.js:35-40
  if (normalize == MI355FFT_NORM_NONE) return 1.0;
  if (normalize == MI355FFT_NORM_UNITARY) return 1.0 / std::sqrt(n_total);
  return inverse ? 1.0 / n_total : 1.0;
}

// bytes a strided side must cover: offset + (batch-1)*batchStride + sum (shape_d-1)*stride_d + 1 elements
// (runtime/tensor_descriptor.js:113-121)
uint64_t strided_extent_elems(const mi355fft_side_layout& l, const int64_t* shape, int rank, int64_t batch, int64_t extra) {
  int64_t last = l.offset_elements + extra + (batch - 1) * (l.batch_stride_elements > 0 ? l.batch_stride_elements : prodv(shape, rank));
  for (int d = 0; d < rank; ++d) last += (shape[d] - 1) * l.strides[d];
  return (uint64_t)(last + 1);
}

int validate_common(const mi355fft_plan_desc& d, std::string& err) {
  if (d.struct_size != sizeof(mi355fft_plan_desc)) { err = "mi355fft_plan_desc.struct_size mismatch (ABI)"; return MI355FFT_ERR_INVALID; }
  if (d.rank < 1 || d.rank > MI355FFT_MAX_RANK) { err = "shape must be an array of one or more positive dimensions"; return MI355FFT_ERR_INVALID; }
  for (int i = 0; i < d.rank; ++i)
    if (d.shape[i] <= 0) { err = "shape elements must be positive ints"; return MI355FFT_ERR_INVALID; }
  if (d.batch <= 0) { err = "batch must be positive int; got " + std::to_string(d.batch); return MI355FFT_ERR_INVALID; }
  if (d.normalize < 0 || d.normalize > 2) { err = "normalize must be one of \"none\", \"backward\", \"unitary\""; return MI355FFT_ERR_INVALID; }
  if (prodv(d.shape, d.rank) * d.batch > ((int64_t)1 << 40)) { err = "Unsupported: more than 2^40 points in one plan"; return MI355FFT_ERR_UNSUPPORTED; }
  for (const mi355fft_side_layout* l : {&d.input, &d.output})
    if (l->strided)
      for (int i = 0; i < d.rank; ++i)
        if (l->strides[i] <= 0) { err = "layout strides must be positive ints"; return MI355FFT_ERR_INVALID; }
  return MI355FFT_OK;
}

// intersection of the view box [off, off+vshape) with the logical domain [0, lshape): region extent, its start in
// logical coordinates and in view coordinates.  false when empty.
bool view_region(const mi355fft_io_view& v, const int64_t* lshape, int rank, int64_t* ext, int64_t* lstart, int64_t* vstart) {
  for (int i = 0; i < rank; ++i) {
    const int64_t s = std::max<int64_t>(0, v.offset[i]), e = std::min<int64_t>(lshape[i], v.offset[i] + v.shape[i]);
    if (e <= s) return false;
    ext[i] = e - s; lstart[i] = s; vstart[i] = s - v.offset[i];
  }
  return true;
}

void emit_zero_outside(Builder& b, PtrRef data, const mi355fft_zero_range& z, const int64_t* shape, int rank, int64_t batch, bool real = false) {
  Step& st = b.push(ST_ZERO_OUTSIDE);
  st.p[0] = data; st.i[3] = real ? 1 : 0;
  const int64_t per = prodv(shape, rank);
  st.i[0] = per * batch; st.i[1] = per; st.i[2] = rank;
  for (int i = 0; i < rank; ++i) { st.shape[i] = shape[i]; st.sa[i] = z.start[i]; st.sb[i] = z.end[i]; }
  st.grid = b.generic_grid(st.i[0]);
}

int validate_views(const mi355fft_plan_desc& d, std::string& err, const int64_t* read_shape = nullptr, const int64_t* write_shape = nullptr) {
  for (const mi355fft_io_view* v : {&d.io_input, &d.io_output})
    if (v->enabled)
      for (int i = 0; i < d.rank; ++i)
        if (v->shape[i] <= 0) { err = "ioView shape must be an array of positive ints"; return MI355FFT_ERR_INVALID; }
  const char* names[2] = {"zeroPad.read", "zeroPad.write"};
  const int64_t* shapes[2] = {read_shape ? read_shape : d.shape, write_shape ? write_shape : d.shape};   // r2c writes / c2r reads the PACKED domain
  int k = 0;
  for (const mi355fft_zero_range* z : {&d.zero_read, &d.zero_write}) {
    if (z->enabled)
      for (int i = 0; i < d.rank; ++i) {
        if (z->start[i] < 0 || z->end[i] < 0 || z->start[i] > z->end[i] || z->end[i] > shapes[k][i]) {
          err = std::string(names[k]) + ": need 0 <= start[" + std::to_string(i) + "] <= end[" + std::to_string(i) + "] <= shape[" + std::to_string(i) + "]";
          return MI355FFT_ERR_INVALID;
        }
      }
    ++k;
  }
  return MI355FFT_OK;
}

// ---- a side of a plan as an address map over its logical domain (kern_lines.hpp side_line) ----
// Input side: element i of the logical domain `lshape` is read at offset + sum (i_d - viewOffset_d) * stride_d inside the box where
// the ioView.input window, the logical domain and the zeroPad.read range meet, and is zero elsewhere.
SideMap input_side_map(const mi355fft_plan_desc& d, const int64_t* lshape, int rank) {
  const bool vin = d.io_input.enabled != 0;
  const int64_t* ishape = vin ? d.io_input.shape : lshape;
  SideMap m = Builder::dense_map(lshape, rank);
  int64_t dense = 1;
  for (int i = 0; i < rank; ++i) {
    m.stride[i] = d.input.strided ? d.input.strides[i] : dense;
    dense *= ishape[i];
    const int64_t voff = vin ? d.io_input.offset[i] : 0;
    int64_t lo = std::max<int64_t>(0, voff), hi = vin ? std::min<int64_t>(lshape[i], voff + d.io_input.shape[i]) : lshape[i];
    if (d.zero_read.enabled) { lo = std::max(lo, d.zero_read.start[i]); hi = std::min(hi, d.zero_read.end[i]); }
    if (hi < lo) hi = lo;
    m.lo[i] = (int)lo; m.hi[i] = (int)hi;
    m.offset -= voff * m.stride[i];
  }
  m.offset += d.input.strided ? d.input.offset_elements : 0;
  m.batch_stride = d.input.strided && d.input.batch_stride_elements > 0 ? d.input.batch_stride_elements : dense;
  return m;
}
// Output side: element i is written inside the ioView.output window only, as zero outside the zeroPad.write range.
SideMap output_side_map(const mi355fft_plan_desc& d, const int64_t* lshape, int rank) {
  const bool vout = d.io_output.enabled != 0;
  const int64_t* oshape = vout ? d.io_output.shape : lshape;
  SideMap m = Builder::dense_map(lshape, rank);
  int64_t dense = 1;
  for (int i = 0; i < rank; ++i) {
    m.stride[i] = d.output.strided ? d.output.strides[i] : dense;
    dense *= oshape[i];
    const int64_t voff = vout ? d.io_output.offset[i] : 0;
    int64_t lo = std::max<int64_t>(0, voff), hi = vout ? std::min<int64_t>(lshape[i], voff + d.io_output.shape[i]) : lshape[i];
    if (hi < lo) hi = lo;
    m.lo[i] = (int)lo; m.hi[i] = (int)hi;
    if (d.zero_write.enabled) { m.zlo[i] = (int)d.zero_write.start[i]; m.zhi[i] = (int)d.zero_write.end[i]; }
    m.offset -= voff * m.stride[i];
  }
  m.offset += d.output.strided ? d.output.offset_elements : 0;
  m.batch_stride = d.output.strided && d.output.batch_stride_elements > 0 ? d.output.batch_stride_elements : dense;
  return m;
}
uint64_t side_bytes(const mi355fft_side_layout& lay, const mi355fft_io_view& view, const int64_t* lshape, int rank, int64_t batch, int64_t elem) {
  const int64_t* pshape = view.enabled ? view.shape : lshape;
  return lay.strided ? strided_extent_elems(lay, pshape, rank, batch, 0) * elem : (uint64_t)prodv(pshape, rank) * batch * elem;
}

// ---- the two sides of an r2c / c2r plan (the same staging build_c2c does inline, for real or complex elements) ----
// Input: strided layout, ioView.input (embed into the zero-filled logical domain) and zeroPad.read produce a dense logical
// array in the workspace; a plain dense input is used where it lies.  `phys_n`: elements of one physical item.
PtrRef stage_side_input(const mi355fft_plan_desc& d, Builder& b, PtrRef in, const int64_t* lshape, bool real, uint64_t& in_bytes) {
  const int rank = d.rank;
  const int64_t elem = real ? 4 : 8, n = prodv(lshape, rank);
  const bool vin = d.io_input.enabled != 0;
  const int64_t* ishape = vin ? d.io_input.shape : lshape;
  const int64_t in_n = prodv(ishape, rank);
  in_bytes = d.input.strided ? strided_extent_elems(d.input, ishape, rank, d.batch, 0) * elem : (uint64_t)in_n * d.batch * elem;
  if (!(d.input.strided || vin || d.zero_read.enabled)) return in;
  const PtrRef src = b.alloc_work((uint64_t)n * d.batch * elem);
  if (vin) {
    int64_t ext[8], ls[8], vs[8];
    const bool any = view_region(d.io_input, lshape, rank, ext, ls, vs);
    bool covers = any;
    for (int i = 0; any && i < rank; ++i) covers = covers && ext[i] == lshape[i];
    if (!covers) { Step& z = b.push(ST_ZERO); z.p[0] = src; z.i[0] = n * d.batch * (elem / 4); z.grid = b.generic_grid(z.i[0]); }
    if (any) {
      mi355fft_side_layout lay = d.input;
      int64_t vstride = 1, poff = 0;
      for (int i = 0; i < rank; ++i) { poff += vs[i] * (lay.strided ? lay.strides[i] : vstride); vstride *= ishape[i]; }
      if (!lay.strided) { lay.strided = 1; int64_t st = 1; for (int i = 0; i < rank; ++i) { lay.strides[i] = st; st *= ishape[i]; } lay.offset_elements = 0; lay.batch_stride_elements = in_n; }
      else if (lay.batch_stride_elements <= 0) lay.batch_stride_elements = in_n;
      b.emit_strided(true, in, src, lay, ext, rank, d.batch, lshape, ls, n, poff, real);
    }
    b.ir.route += "embed ";
  } else if (d.input.strided) {
    b.emit_strided(true, in, src, d.input, lshape, rank, d.batch, lshape, nullptr, n, 0, real);
    b.ir.route += "gather ";
  } else {
    Step& c = b.push(ST_COPY); c.p[0] = in; c.p[1] = src; c.i[0] = n * d.batch * elem;
  }
  if (d.zero_read.enabled) { emit_zero_outside(b, src, d.zero_read, lshape, rank, d.batch, real); b.ir.route += "zero-read "; }
  return src;
}
// Output: where the transform should write (the caller's buffer, or a dense logical staging array when a strided layout /
// ioView.output follows), and the steps that finish the side afterwards.
PtrRef side_output_target(const mi355fft_plan_desc& d, Builder& b, PtrRef out, const int64_t* lshape, bool real, uint64_t& out_bytes) {
  const int rank = d.rank;
  const int64_t elem = real ? 4 : 8;
  const int64_t* oshape = d.io_output.enabled ? d.io_output.shape : lshape;
  out_bytes = d.output.strided ? strided_extent_elems(d.output, oshape, rank, d.batch, 0) * elem : (uint64_t)prodv(oshape, rank) * d.batch * elem;
  if (d.output.strided || d.io_output.enabled) return b.alloc_work((uint64_t)prodv(lshape, rank) * d.batch * elem);
  return out;
}
int finish_side_output(const mi355fft_plan_desc& d, Builder& b, PtrRef out, PtrRef dst, const int64_t* lshape, bool real, std::string& err) {
  const int rank = d.rank;
  const int64_t elem = real ? 4 : 8, n = prodv(lshape, rank);
  if (d.zero_write.enabled) { emit_zero_outside(b, dst, d.zero_write, lshape, rank, d.batch, real); b.ir.route += "zero-write "; }
  if (d.io_output.enabled) {
    const int64_t* oshape = d.io_output.shape;
    const int64_t out_n = prodv(oshape, rank);
    if (d.io_output.clear_outside) {
      if (d.output.strided) { err = "Unsupported: ioView.output.clearOutside with a strided output layout"; return MI355FFT_ERR_UNSUPPORTED; }
      Step& z = b.push(ST_ZERO); z.p[0] = out; z.i[0] = out_n * d.batch * (elem / 4); z.grid = b.generic_grid(z.i[0]);
    }
    int64_t ext[8], ls[8], vs[8];
    if (view_region(d.io_output, lshape, rank, ext, ls, vs)) {
      mi355fft_side_layout lay = d.output;
      int64_t vstride = 1, poff = 0;
      for (int i = 0; i < rank; ++i) { poff += vs[i] * (lay.strided ? lay.strides[i] : vstride); vstride *= oshape[i]; }
      if (!lay.strided) { lay.strided = 1; int64_t st = 1; for (int i = 0; i < rank; ++i) { lay.strides[i] = st; st *= oshape[i]; } lay.offset_elements = 0; lay.batch_stride_elements = out_n; }
      else if (lay.batch_stride_elements <= 0) lay.batch_stride_elements = out_n;
      b.emit_strided(false, out, dst, lay, ext, rank, d.batch, lshape, ls, n, poff, real);
    }
    b.ir.route += "extract ";
  } else if (d.output.strided) {
    b.emit_strided(false, out, dst, d.output, lshape, rank, d.batch, lshape, nullptr, n, 0, real);
    b.ir.route += "scatter ";
  }
  return MI355FFT_OK;
}

int build_c2c(const mi355fft_plan_desc& d, Builder& b, std::string& err) {
  if (d.direction != MI355FFT_FORWARD && d.direction != MI355FFT_INVERSE) { err = "direction must be one of \"forward\", \"inverse\""; return MI355FFT_ERR_INVALID; }
  int rc = validate_views(d, err);
  if (rc) return rc;
  const bool inverse = d.direction == MI355FFT_INVERSE;
  const int rank = d.rank;
  const int64_t n = prodv(d.shape, rank);
  const float scale = (float)scale_factor(d.normalize, inverse, (double)n);
  const bool vin = d.io_input.enabled != 0, vout = d.io_output.enabled != 0;
  if (d.in_place && (vin || vout || d.input.strided || d.output.strided)) { err = "inPlace=true cannot be combined with ioView or strided layouts"; return MI355FFT_ERR_INVALID; }
  const int64_t* ishape = vin ? d.io_input.shape : d.shape;     // physical shapes of the two sides
  const int64_t* oshape = vout ? d.io_output.shape : d.shape;
  const int64_t in_n = prodv(ishape, rank), out_n = prodv(oshape, rank);
  PtrRef in(BUF_INPUT, 0), out(d.in_place ? BUF_INPUT : BUF_OUTPUT, 0);
  b.ir.in_bytes = d.input.strided ? strided_extent_elems(d.input, ishape, rank, d.batch, 0) * 8 : (uint64_t)in_n * d.batch * 8;
  b.ir.out_bytes = d.output.strided ? strided_extent_elems(d.output, oshape, rank, d.batch, 0) * 8 : (uint64_t)out_n * d.batch * 8;
  if (d.axes_mask >> rank) { err = "Invalid axis in axes for rank " + std::to_string(rank); return MI355FFT_ERR_INVALID; }
  if (vout && d.io_output.clear_outside && d.output.strided) { err = "Unsupported: ioView.output.clearOutside with a strided output layout"; return MI355FFT_ERR_UNSUPPORTED; }

  // ---- rank-1 lane layouts: unit stride along the line on both sides -> one launch, no staging ----
  if (rank == 1 && !vin && !vout && !d.zero_read.enabled && !d.zero_write.enabled && !d.in_place && (d.input.strided || d.output.strided) &&
      (!d.input.strided || d.input.strides[0] == 1) && (!d.output.strided || d.output.strides[0] == 1)) {
    const int64_t ip = d.input.strided && d.input.batch_stride_elements > 0 ? d.input.batch_stride_elements : n;
    const int64_t op = d.output.strided && d.output.batch_stride_elements > 0 ? d.output.batch_stride_elements : n;
    const int64_t ioff = d.input.strided ? d.input.offset_elements : 0, ooff = d.output.strided ? d.output.offset_elements : 0;
    // the line kernels address a tile with 32-bit element offsets: T lines * pitch must stay below 2^31 elements
    if (ip >= n && op >= n && ip < ((int64_t)1 << 24) && op < ((int64_t)1 << 24) &&
        b.emit_lines_pitched(in.plus(ioff * 8), out.plus(ooff * 8), n, d.batch, inverse, scale, ip, op))
      return MI355FFT_OK;
    // four-step sizes (r03): the fused kernels take the two pitches as they are — channel lanes and whdcn layouts of long lines need
    // no gather / scatter either.  Any other route of emit_axis would ignore the pitches: it is rolled back and the staging path taken.
    if (ip >= n && op >= n && n > b.opt.max_line && is_pow2(n) && b.opt.fuse_views && !b.opt.force_generic) {
      const size_t mark = b.ir.steps.size();
      const std::string route_mark = b.ir.route;
      const uint64_t work_mark = b.work_top, work_bytes_mark = b.ir.work_bytes;
      b.lane_in_pitch = ip; b.lane_out_pitch = op; b.lane_used = false;
      const int rcl = b.emit_axis(in.plus(ioff * 8), out.plus(ooff * 8), n, 1, d.batch, inverse, scale, err);
      b.lane_in_pitch = b.lane_out_pitch = 0;
      if (rcl == MI355FFT_OK && b.lane_used) { b.ir.route += "lanes[pitch=" + std::to_string(ip) + "/" + std::to_string(op) + "] "; return MI355FFT_OK; }
      b.ir.steps.resize(mark); b.ir.route = route_mark; b.work_top = work_mark; b.ir.work_bytes = work_bytes_mark; err.clear();
    }
  }

  // ---- rank-1 views of a four-step line with a VIEW instance (r03): ranges as predicates of the fused kernel's loads and stores ----
  if (rank == 1 && !d.in_place && (d.axes_mask == 0 || d.axes_mask == 1) && n > b.opt.max_line && (vin || vout || d.zero_read.enabled || d.zero_write.enabled)) {
    const SideMap im = input_side_map(d, d.shape, 1), om = output_side_map(d, d.shape, 1);
    if (im.stride[0] == 1 && om.stride[0] == 1 && n < ((int64_t)1 << 30)) {
      const size_t mark = b.ir.steps.size();
      if (vout && d.io_output.clear_outside) { Step& z = b.push(ST_ZERO); z.p[0] = out; z.i[0] = out_n * d.batch * 2; z.grid = b.generic_grid(z.i[0]); }
      if (b.emit_xcd_view(in, out, n, d.batch, inverse, scale, im, om)) return MI355FFT_OK;
      b.ir.steps.resize(mark);
    }
  }

  // ---- sides fused into the line kernels: where the first / last transformed axis runs as a line-kernel
  // launch, that side's strided layout, ioView and zero range become the launch's address map (kern_lines.hpp
  // fft_lines_mapped_kernel) — no gather / embed / zero / extract / scatter pass and no staging copy of the array
  int fa = -1, la = -1;
  Builder::nd_first_last(d.shape, rank, d.axes_mask, fa, la);
  const auto stride_below = [&](int a) { int64_t S = 1; for (int i = 0; i < a; ++i) S *= d.shape[i]; return S; };
  const bool need_in = d.input.strided || vin || d.zero_read.enabled;
  const bool need_out = d.output.strided || vout || d.zero_write.enabled;
  const bool small = n < ((int64_t)1 << 31);
  const bool fuse_in = need_in && small && fa >= 0 && b.axis_mappable(d.shape[fa], stride_below(fa), inverse);
  const bool fuse_out = need_out && small && la >= 0 && b.axis_mappable(d.shape[la], stride_below(la), inverse);
  SideMap imap, omap;
  if (fuse_in) imap = input_side_map(d, d.shape, rank);
  if (fuse_out) omap = output_side_map(d, d.shape, rank);

  // ---- input side: dense logical staging when anything but a plain dense read is asked for and the first pass cannot map it ----
  PtrRef src = in;
  const bool stage_in = !fuse_in && (d.input.strided || vin || (d.zero_read.enabled && !d.in_place));
  if (stage_in) {
    src = b.alloc_work((uint64_t)n * d.batch * 8);
    if (vin) {
      int64_t ext[8], ls[8], vs[8];
      const bool any = view_region(d.io_input, d.shape, rank, ext, ls, vs);
      bool covers = any;
      for (int i = 0; any && i < rank; ++i) covers = covers && ext[i] == d.shape[i];
      if (!covers) { Step& z = b.push(ST_ZERO); z.p[0] = src; z.i[0] = n * d.batch * 2; z.grid = b.generic_grid(z.i[0]); }
      if (any) {
        // physical offset of the region's first element inside the view
        mi355fft_side_layout lay = d.input;
        int64_t vstride = 1, poff = 0;
        for (int i = 0; i < rank; ++i) { poff += vs[i] * (lay.strided ? lay.strides[i] : vstride); vstride *= ishape[i]; }
        if (!lay.strided) { lay.strided = 1; int64_t st = 1; for (int i = 0; i < rank; ++i) { lay.strides[i] = st; st *= ishape[i]; } lay.offset_elements = 0; lay.batch_stride_elements = in_n; }
        else if (lay.batch_stride_elements <= 0) lay.batch_stride_elements = in_n;
        b.emit_strided(true, in, src, lay, ext, rank, d.batch, d.shape, ls, n, poff);
      }
      b.ir.route += "embed ";
    } else if (d.input.strided) {
      b.emit_strided(true, in, src, d.input, d.shape, rank, d.batch, d.shape, nullptr, n, 0);
      b.ir.route += "gather ";
    } else {
      Step& c = b.push(ST_COPY); c.p[0] = in; c.p[1] = src; c.i[0] = n * d.batch * 8;
    }
  }
  if (d.zero_read.enabled && !fuse_in) { emit_zero_outside(b, src, d.zero_read, d.shape, rank, d.batch); b.ir.route += "zero-read "; }

  // ---- transform ----
  // dst: the dense array the passes work on.  A staged or mapped output side needs one that is not the caller's output: the
  // input staging if there is one, else workspace — except when a single pass maps both sides (in -> out directly).
  const bool stage_out = !fuse_out && (d.output.strided || vout);
  PtrRef dst = out;
  if (stage_out || (fuse_out && (d.output.strided || vout) && !(fuse_in && fa == la))) dst = stage_in ? src : b.alloc_work((uint64_t)n * d.batch * 8);
  if (fuse_out && vout && d.io_output.clear_outside) {   // view elements outside the logical domain: zeroed before the store pass
    Step& z = b.push(ST_ZERO); z.p[0] = out; z.i[0] = out_n * d.batch * 2; z.grid = b.generic_grid(z.i[0]);
  }
  rc = b.emit_nd(src, dst, d.shape, rank, d.batch, inverse, scale, err, 0, d.axes_mask, fuse_in ? &imap : nullptr, fuse_out ? &omap : nullptr, out);
  if (rc) return rc;
  if (fuse_out) return MI355FFT_OK;
  if (d.zero_write.enabled) { emit_zero_outside(b, dst, d.zero_write, d.shape, rank, d.batch); b.ir.route += "zero-write "; }

  // ---- output side ----
  if (vout) {
    if (d.io_output.clear_outside) { Step& z = b.push(ST_ZERO); z.p[0] = out; z.i[0] = out_n * d.batch * 2; z.grid = b.generic_grid(z.i[0]); }
    int64_t ext[8], ls[8], vs[8];
    if (view_region(d.io_output, d.shape, rank, ext, ls, vs)) {
      mi355fft_side_layout lay = d.output;
      int64_t vstride = 1, poff = 0;
      for (int i = 0; i < rank; ++i) { poff += vs[i] * (lay.strided ? lay.strides[i] : vstride); vstride *= oshape[i]; }
      if (!lay.strided) { lay.strided = 1; int64_t st = 1; for (int i = 0; i < rank; ++i) { lay.strides[i] = st; st *= oshape[i]; } lay.offset_elements = 0; lay.batch_stride_elements = out_n; }
      else if (lay.batch_stride_elements <= 0) lay.batch_stride_elements = out_n;
      b.emit_strided(false, out, dst, lay, ext, rank, d.batch, d.shape, ls, n, poff);
    }
    b.ir.route += "extract ";
  } else if (d.output.strided) {
    b.emit_strided(false, out, dst, d.output, d.shape, rank, d.batch, d.shape, nullptr, n, 0);
    b.ir.route += "scatter ";
  }
  return MI355FFT_OK;
}

// r2c along axis 0 (packed P = N/2+1 bins), then c2c along the remaining axes of the packed array
int build_r2c(const mi355fft_plan_desc& d, Builder& b, std::string& err) {
  if (d.direction != MI355FFT_FORWARD) { err = "r2c supports direction:\"forward\" only"; return MI355FFT_ERR_INVALID; }
  if (d.in_place) { err = "inPlace=true is supported only on c2c"; return MI355FFT_ERR_INVALID; }
  const int64_t N = d.shape[0], P = N / 2 + 1;
  if (N < 2) { err = "r2c requires shape[0] >= 2"; return MI355FFT_ERR_INVALID; }
  const int64_t n = prodv(d.shape, d.rank), lines = d.batch * (n / N);
  const float scale = (float)scale_factor(d.normalize, false, (double)n);
  int64_t pshape[MI355FFT_MAX_RANK];
  for (int i = 0; i < d.rank; ++i) pshape[i] = d.shape[i];
  pshape[0] = P;
  // ioView.input / zeroPad.read live on the real logical domain, ioView.output / zeroPad.write on the packed one (r2c.js:72-123)
  if (int rv = validate_views(d, err, d.shape, pshape)) return rv;
  const PtrRef user_in(BUF_INPUT, 0), user_out(BUF_OUTPUT, 0);
  const int rank = d.rank;
  // ---- sides fused into the launches: the real side rides the r2c line kernel's first loads, the packed
  // side its store pass (rank 1) or the store pass of the last c2c axis (rank > 1); whatever cannot be mapped is staged as before
  const bool need_in = d.input.strided || d.io_input.enabled || d.zero_read.enabled;
  const bool need_out = d.output.strided || d.io_output.enabled || d.zero_write.enabled;
  const uint32_t upper = rank > 1 ? (((uint32_t)1 << rank) - 1u) & ~1u : 0u;
  int fa = -1, la = -1;
  if (rank > 1) Builder::nd_first_last(pshape, rank, upper, fa, la);
  const bool small = n * 2 < ((int64_t)1 << 31);
  const bool lines0 = small && b.lines_r2c_kernel(N, false, true) != nullptr;     // axis 0 can be the mapped line kernel
  const auto stride_below = [&](int a) { int64_t S = 1; for (int i = 0; i < a; ++i) S *= pshape[i]; return S; };
  const bool fuse_in = need_in && lines0;
  const bool fuse_out = need_out && small && (la >= 1 ? b.axis_mappable(pshape[la], stride_below(la), false) : lines0);
  if (fuse_out && d.io_output.enabled && d.io_output.clear_outside && d.output.strided) { err = "Unsupported: ioView.output.clearOutside with a strided output layout"; return MI355FFT_ERR_UNSUPPORTED; }
  const bool map0 = fuse_in || (fuse_out && la < 1);                              // the axis-0 launch carries at least one map
  b.ir.out_bytes = side_bytes(d.output, d.io_output, pshape, rank, d.batch, 8);
  PtrRef in = user_in;
  if (fuse_in) b.ir.in_bytes = side_bytes(d.input, d.io_input, d.shape, rank, d.batch, 4);
  else in = stage_side_input(d, b, user_in, d.shape, true, b.ir.in_bytes);
  // the dense packed array the passes work on: the caller's output unless a strided layout / ioView follows (staged or mapped)
  PtrRef out = (d.output.strided || d.io_output.enabled) ? b.alloc_work((uint64_t)lines * P * 8) : user_out;
  const SideMap omap = fuse_out ? output_side_map(d, pshape, rank) : SideMap();
  if (fuse_out && d.io_output.enabled && d.io_output.clear_outside) {   // view elements outside the logical domain: zeroed before the store pass
    Step& z = b.push(ST_ZERO); z.p[0] = user_out; z.i[0] = prodv(d.io_output.shape, rank) * d.batch * 2; z.grid = b.generic_grid(z.i[0]);
  }
  if (map0) {
    const SideMap im = fuse_in ? input_side_map(d, d.shape, rank) : Builder::dense_map(d.shape, rank);
    const bool store0 = fuse_out && la < 1;
    const SideMap om = store0 ? omap : Builder::dense_map(pshape, rank);
    if (!b.emit_lines_r2c(in, store0 ? user_out : out, N, lines, scale, false, 0, &im, &om)) { err = "no mapped r2c line kernel"; return MI355FFT_ERR_UNSUPPORTED; }
    if (store0) return MI355FFT_OK;
  } else if (N % 2 == 0) {
    const int rc = b.emit_r2c_even(in, out, N, lines, scale, err);
    if (rc) return rc;
  } else {
    PtrRef full = b.alloc_work((uint64_t)lines * N * 8);
    Step& e = b.push(ST_REAL_TO_COMPLEX);
    e.p[0] = in; e.p[1] = full; e.i[0] = lines * N; e.grid = b.generic_grid(lines * N);
    int rc = b.emit_axis(full, full, N, 1, lines, false, 1.0f, err);
    if (rc) return rc;
    Step& p = b.push(ST_PACK_HALF);
    p.p[0] = full; p.p[1] = out; p.i[0] = N; p.i[1] = P; p.i[2] = lines; p.i[3] = P; p.f[0] = scale;
    p.grid = b.generic_grid(lines * P);
    b.ir.route += "r2c-full ";
  }
  if (rank > 1) {
    const bool last_mapped = fuse_out && la >= 1;
    const int rc = last_mapped ? b.emit_nd(out, out, pshape, rank, d.batch, false, 1.0f, err, 1, upper, nullptr, &omap, user_out)
                               : b.emit_nd(out, out, pshape, rank, d.batch, false, 1.0f, err, 1);
    if (rc) return rc;
    if (last_mapped) return MI355FFT_OK;
  }
  return finish_side_output(d, b, user_out, out, pshape, false, err);
}

int build_c2r(const mi355fft_plan_desc& d, Builder& b, std::string& err) {
  if (d.direction != MI355FFT_INVERSE) { err = "c2r supports direction:\"inverse\" only"; return MI355FFT_ERR_INVALID; }
  if (d.in_place) { err = "inPlace=true is supported only on c2c"; return MI355FFT_ERR_INVALID; }
  const int64_t N = d.shape[0], P = N / 2 + 1;
  if (N < 2) { err = "c2r requires shape[0] >= 2"; return MI355FFT_ERR_INVALID; }
  const int64_t n = prodv(d.shape, d.rank), lines = d.batch * (n / N);
  const float scale = (float)scale_factor(d.normalize, true, (double)n);
  int64_t pshape[MI355FFT_MAX_RANK];
  for (int i = 0; i < d.rank; ++i) pshape[i] = d.shape[i];
  pshape[0] = P;
  // ioView.input / zeroPad.read live on the packed domain, ioView.output / zeroPad.write on the real one (c2r.js:168-220)
  if (int rv = validate_views(d, err, pshape, d.shape)) return rv;
  const PtrRef user_in(BUF_INPUT, 0), user_out(BUF_OUTPUT, 0);
  const int rank = d.rank;
  // ---- sides fused into the launches (as build_r2c): the packed side rides the first inverse c2c axis (rank > 1) or the c2r line
  // kernel's pre-split loads (rank 1), the real side the c2r line kernel's store pass
  const bool need_in = d.input.strided || d.io_input.enabled || d.zero_read.enabled;
  const bool need_out = d.output.strided || d.io_output.enabled || d.zero_write.enabled;
  const uint32_t upper = rank > 1 ? (((uint32_t)1 << rank) - 1u) & ~1u : 0u;
  int fa = -1, la = -1;
  if (rank > 1) Builder::nd_first_last(pshape, rank, upper, fa, la);
  const bool small = n * 2 < ((int64_t)1 << 31);
  const bool lines0 = small && b.lines_r2c_kernel(N, true, true) != nullptr;
  const auto stride_below = [&](int a) { int64_t S = 1; for (int i = 0; i < a; ++i) S *= pshape[i]; return S; };
  const bool fuse_in = need_in && small && (fa >= 1 ? b.axis_mappable(pshape[fa], stride_below(fa), true) : lines0);
  const bool fuse_out = need_out && lines0;
  if (fuse_out && d.io_output.enabled && d.io_output.clear_outside && d.output.strided) { err = "Unsupported: ioView.output.clearOutside with a strided output layout"; return MI355FFT_ERR_UNSUPPORTED; }
  const bool map0 = fuse_out || (fuse_in && fa < 1);
  b.ir.out_bytes = side_bytes(d.output, d.io_output, d.shape, rank, d.batch, 4);
  PtrRef in = user_in;
  if (fuse_in) b.ir.in_bytes = side_bytes(d.input, d.io_input, pshape, rank, d.batch, 8);
  else in = stage_side_input(d, b, user_in, pshape, false, b.ir.in_bytes);
  const SideMap imap = fuse_in ? input_side_map(d, pshape, rank) : SideMap();
  PtrRef out = user_out;
  if ((d.output.strided || d.io_output.enabled) && !fuse_out) out = b.alloc_work((uint64_t)n * d.batch * 4);
  PtrRef packed = in;
  bool in_mapped_done = false;
  if (fa >= 1) {
    // inverse c2c over axes 1.. of the packed spectrum, out of place into the workspace: the caller's input is not modified
    packed = b.alloc_work((uint64_t)lines * P * 8);
    const bool first_mapped = fuse_in;
    const int rc = first_mapped ? b.emit_nd(in, packed, pshape, rank, d.batch, true, 1.0f, err, 1, upper, &imap, nullptr)
                                : b.emit_nd(in, packed, pshape, rank, d.batch, true, 1.0f, err, 1);
    if (rc) return rc;
    in_mapped_done = first_mapped;
  }
  if (fuse_out && d.io_output.enabled && d.io_output.clear_outside) {
    Step& z = b.push(ST_ZERO); z.p[0] = user_out; z.i[0] = prodv(d.io_output.shape, rank) * d.batch; z.grid = b.generic_grid(z.i[0]);
  }
  if (map0) {
    const SideMap im = (fuse_in && !in_mapped_done) ? imap : Builder::dense_map(pshape, rank);
    const SideMap om = fuse_out ? output_side_map(d, d.shape, rank) : Builder::dense_map(d.shape, rank);
    if (!b.emit_lines_r2c(packed, out, N, lines, scale, true, 0, &im, &om)) { err = "no mapped c2r line kernel"; return MI355FFT_ERR_UNSUPPORTED; }
    if (fuse_out) return MI355FFT_OK;
  } else if (N % 2 == 0) {
    const int rc = b.emit_c2r_even(packed, out, N, lines, scale, err);
    if (rc) return rc;
  } else {
    PtrRef full = b.alloc_work((uint64_t)lines * N * 8);
    Step& u = b.push(ST_UNPACK_HERM);
    u.p[0] = packed; u.p[1] = full; u.i[0] = N; u.i[1] = P; u.i[2] = lines; u.i[3] = P;
    u.grid = b.generic_grid(lines * N);
    int rc = b.emit_axis(full, full, N, 1, lines, true, 1.0f, err);
    if (rc) return rc;
    Step& r = b.push(ST_COMPLEX_TO_REAL);
    r.p[0] = full; r.p[1] = out; r.i[0] = lines * N; r.f[0] = scale;
    r.grid = b.generic_grid(lines * N);
    b.ir.route += "c2r-full ";
  }
  if (int rv = finish_side_output(d, b, user_out, out, d.shape, true, err)) return rv;
  return MI355FFT_OK;
}

// y_k = IFFT( FFT(x) .* (conj?)FFT(h_k) ) / Nfft, cropped per boundary, written per output layout / lanes
// (runtime/plans/fftconv.js:308-709, exec :1415-1712; reference semantics: src/utils/math.js:469-603)
// DCT-I..IV / DST-I..IV over real buffers (dct_fft.js): per axis a pre-pass into complex lines of length L, the complex FFT
// of those lines, a post-pass back into the real array (kern_trig.hpp); one scale by normalizeScaleFactor(prod(shape)) folded
// into the last post-pass (dct_fft.js:882).  Layout strides, ioView and zeroPad ride the same side staging as r2c / c2r.
int build_trig(const mi355fft_plan_desc& d, Builder& b, std::string& err) {
  if (d.direction != MI355FFT_FORWARD && d.direction != MI355FFT_INVERSE) { err = "direction must be one of \"forward\", \"inverse\""; return MI355FFT_ERR_INVALID; }
  if (d.in_place) { err = "DCT/DST inPlace is not supported in current implementation"; return MI355FFT_ERR_INVALID; }
  const int rank = d.rank;
  for (int i = 0; i < rank; ++i) if (d.shape[i] < 2) { err = "All DCT/DST dimensions must be >= 2"; return MI355FFT_ERR_INVALID; }
  if (int rv = validate_views(d, err)) return rv;
  const bool fwd = d.direction == MI355FFT_FORWARD;
  int kind;
  switch (d.type) {                                     // dct_fft.js:48-57
    case MI355FFT_DCT1: kind = 0; break;
    case MI355FFT_DCT2: kind = fwd ? 1 : 2; break;
    case MI355FFT_DCT3: kind = fwd ? 2 : 1; break;
    case MI355FFT_DCT4: kind = 3; break;
    case MI355FFT_DST1: kind = 4; break;
    case MI355FFT_DST2: kind = fwd ? 5 : 6; break;
    case MI355FFT_DST3: kind = fwd ? 6 : 5; break;
    default: kind = 7; break;
  }
  const int64_t n = prodv(d.shape, rank);
  const float scale = (float)scale_factor(d.normalize, !fwd, (double)n);
  const PtrRef user_out(BUF_OUTPUT, 0);
  PtrRef cur = stage_side_input(d, b, PtrRef(BUF_INPUT, 0), d.shape, true, b.ir.in_bytes);
  const PtrRef dst = side_output_target(d, b, user_out, d.shape, true, b.ir.out_bytes);
  int64_t S = 1;
  int general_axes = 0;
  for (int a = 0; a < rank; ++a) {
    const int64_t N = d.shape[a], lines = d.batch * (n / N);
    const int64_t L = kind == 0 ? 2 * (N - 1) : (kind == 4 ? 2 * (N + 1) : 2 * N);
    const uint64_t mark = b.work_top;
    if (b.opt.trig_real && !b.opt.force_generic && N >= 4 && (N % 2 == 0 || kind == 0 || kind == 4)) {
      // dense lines: a real FFT of length N behind Makhoul's permutation (dct2/dst2 and their inverses), a complex FFT of
      // length N/2 (dct4/dst4), or the r2c of the real even/odd extension (dct1/dst1) -- kern_trig.hpp kinds 8..15
      const bool tfwd = kind == 1 || kind == 5, tinv = kind == 2 || kind == 6, quarter = kind == 3 || kind == 7;
      const int rkind = tfwd ? (kind == 1 ? 8 : 9) : tinv ? (kind == 2 ? 10 : 11) : quarter ? (kind == 3 ? 12 : 13) : (kind == 0 ? 14 : 15);
      // DCT-II / DST-II of dense lines whose half length is a line-kernel size: permutation, real FFT and phase in ONE launch
      if (tfwd && S == 1 && b.opt.trig_fused && b.emit_lines_r2c(cur, dst, N, lines, a == rank - 1 ? scale : 1.0f, false, kind == 1 ? 5 : 6)) {
        cur = dst;
        S *= N;
        continue;
      }
      // DCT-III / DST-III the same way on the c2r line kernel (bins formed from the real line in the pre-split, un-permutation
      // in the store); needs the LDS line buffer: half lengths of 64 and more
      if (tinv && S == 1 && N >= 128 && b.opt.trig_fused && b.emit_lines_r2c(cur, dst, N, lines, a == rank - 1 ? scale : 1.0f, true, kind == 2 ? 7 : 8)) {
        cur = dst;
        S *= N;
        continue;
      }
      const int64_t M = quarter ? N / 2 : (kind == 0 || kind == 4 ? L : N);     // length of the FFT in the middle
      const int64_t P = quarter ? M : M / 2 + 1;                                  // complex elements per line
      const PtrRef v = quarter ? PtrRef() : b.alloc_work((uint64_t)lines * M * 4), V = b.alloc_work((uint64_t)lines * P * 8);
      const float last = a == rank - 1 ? scale : 1.0f;
      Step& pre = b.push(ST_TRIG_PRE);
      pre.p[0] = cur; pre.p[1] = V; pre.p[2] = quarter ? dst : v;
      pre.i[0] = lines; pre.i[1] = N; pre.i[2] = P; pre.i[3] = M; pre.i[4] = rkind; pre.i[5] = S;
      // S > 1 (axes >= 1): 32 x 32 tiles through LDS, one workgroup per tile
      const auto tiled_grid = [&](int64_t per) { return b.oneshot_grid((lines / S) * ((S + 31) / 32) * ((per + 31) / 32)); };   // one workgroup per tile
      pre.grid = S > 1 ? tiled_grid(tinv || quarter ? P : M) : b.oneshot_grid((lines * (tinv || quarter ? P : M) + 255) / 256);   // streaming pass: one-shot grid (r02)
      b.ir.route += "trig-real[kind=" + std::to_string(kind) + "] ";
      const int rc = quarter ? b.emit_axis(V, V, M, 1, lines, false, 1.0f, err)
                   : tinv ? b.emit_c2r_even(V, v, M, lines, 1.0f, err) : b.emit_r2c_even(v, V, M, lines, 1.0f, err);
      if (rc) return rc;
      Step& post = b.push(ST_TRIG_POST);
      post.p[0] = quarter ? cur : v; post.p[1] = V; post.p[2] = dst;
      post.i[0] = lines; post.i[1] = N; post.i[2] = P; post.i[3] = M; post.i[4] = rkind; post.i[5] = S;
      post.f[0] = last;
      post.grid = S > 1 ? tiled_grid(tfwd || quarter ? P : N) : b.oneshot_grid((lines * (tfwd || quarter ? P : N) + 255) / 256);
      b.work_top = mark;
      cur = dst;
      S *= N;
      continue;
    }
    const PtrRef z = b.alloc_work((uint64_t)lines * L * 8);
    ++general_axes;
    Step& pre = b.push(ST_TRIG_PRE);
    pre.p[0] = cur; pre.p[1] = z; pre.p[2] = dst;
    pre.i[0] = lines; pre.i[1] = N; pre.i[2] = L; pre.i[3] = S; pre.i[4] = kind;
    pre.grid = b.generic_grid(lines * L);
    const int rc = b.emit_axis(z, z, L, 1, lines, kind == 2 || kind == 6, 1.0f, err);
    if (rc) return rc;
    Step& post = b.push(ST_TRIG_POST);
    post.p[0] = cur; post.p[1] = z; post.p[2] = dst;
    post.i[0] = lines; post.i[1] = N; post.i[2] = L; post.i[3] = S; post.i[4] = kind;
    post.f[0] = a == rank - 1 ? scale : 1.0f;
    post.grid = b.generic_grid(lines * N);
    b.work_top = mark;         // the complex lines are a per-axis temporary
    cur = dst;
    S *= N;
  }
  if (general_axes) b.ir.route += "trig[kind=" + std::to_string(kind) + "] ";
  return finish_side_output(d, b, user_out, dst, d.shape, true, err);
}

int build_fftconv(const mi355fft_plan_desc& d, Builder& b, std::string& err) {
  if (d.io_input.enabled || d.io_output.enabled) { err = "ioView is not an fftconv option"; return MI355FFT_ERR_INVALID; }
  if (d.in_place) { err = "fftconv inPlace=true is not supported in current implementation"; return MI355FFT_ERR_INVALID; }
  if (d.conv_mode != MI355FFT_CONVOLUTION && d.conv_mode != MI355FFT_CORRELATION) { err = "fftConv.mode must be one of \"convolution\", \"correlation\""; return MI355FFT_ERR_INVALID; }
  if (d.conv_boundary < 0 || d.conv_boundary > 3) { err = "fftConv.boundary must be one of \"circular\", \"linear-full\", \"linear-same\", \"linear-valid\""; return MI355FFT_ERR_INVALID; }
  if (d.conv_kernel_count <= 0) { err = "fftConv.kernelCount must be a positive integer; got " + std::to_string(d.conv_kernel_count); return MI355FFT_ERR_INVALID; }
  const int rank = d.rank;
  const int64_t K = d.conv_kernel_count, B = d.batch;
  int64_t ks[8], fs[8], os[8], ooff[8], zero[8] = {0};
  bool ks_given = false;
  for (int i = 0; i < rank; ++i) if (d.conv_kernel_shape[i] != 0) ks_given = true;
  for (int i = 0; i < rank; ++i) {
    ks[i] = ks_given ? d.conv_kernel_shape[i] : d.shape[i];
    if (ks[i] <= 0) { err = "fftConv.kernelShape must be an array of " + std::to_string(rank) + " positive ints"; return MI355FFT_ERR_INVALID; }
    if (d.conv_boundary == MI355FFT_CIRCULAR) {
      if (ks[i] > d.shape[i]) { err = "fftConv.kernelShape[" + std::to_string(i) + "] must be <= shape[" + std::to_string(i) + "] when fftConv.boundary=\"circular\""; return MI355FFT_ERR_INVALID; }
      fs[i] = d.shape[i]; os[i] = d.shape[i]; ooff[i] = 0;
    } else {
      fs[i] = d.shape[i] + ks[i] - 1;
      if (d.conv_boundary == MI355FFT_LINEAR_FULL) { os[i] = fs[i]; ooff[i] = 0; }
      else if (d.conv_boundary == MI355FFT_LINEAR_SAME) { os[i] = d.shape[i]; ooff[i] = (ks[i] - 1) / 2; }
      else {
        os[i] = d.shape[i] - ks[i] + 1; ooff[i] = ks[i] - 1;
        if (os[i] <= 0) { err = "fftConv.boundary=\"linear-valid\" requires kernelShape[" + std::to_string(i) + "] <= shape[" + std::to_string(i) + "]"; return MI355FFT_ERR_INVALID; }
      }
    }
  }
  // zeroPad ranges live on the FFT domain: read = the embedded data before the forward transform, write = the inverse
  // transform before the crop (fftconv.js:386,542,565)
  if (int rv = validate_views(d, err, fs, fs)) return rv;
  const bool zpad = d.zero_read.enabled || d.zero_write.enabled;
  const int64_t inN = prodv(d.shape, rank), kN = prodv(ks, rank), fN = prodv(fs, rank), oN = prodv(os, rank);
  PtrRef in(BUF_INPUT, 0), out(BUF_OUTPUT, 0), kern(BUF_KERNEL, 0);
  b.ir.kernel_bytes = (uint64_t)K * kN * 8;
  b.ir.in_bytes = d.input.strided ? strided_extent_elems(d.input, d.shape, rank, B, 0) * 8 : (uint64_t)inN * B * 8;
  const int64_t kstride = d.conv_output_kernel_stride_elements;
  if (d.output.strided) {
    if (K > 1 && kstride <= 0) { err = "multi-kernel strided output requires fftConv.channelPolicy.output or fftConv.outputKernelStrideElements"; return MI355FFT_ERR_INVALID; }
    b.ir.out_bytes = strided_extent_elems(d.output, os, rank, B, (K - 1) * kstride) * 8;
  } else b.ir.out_bytes = (uint64_t)K * B * oN * 8;

  // small 1-D circular problems: one launch, one workgroup per batch entry (kern_fftconv.hpp)
  // (throughput-sized problems do better on the line kernels below: measured 49 vs 150 GPoints/s at N=1024, batch 65536)
  const bool lines_mul_ok = b.opt.conv_lines && is_pow2(fN) && fN >= 64 && fN <= b.opt.max_line;
  if (!b.opt.force_generic && !zpad && rank == 1 && d.conv_boundary == MI355FFT_CIRCULAR && K <= 15 &&
      !(lines_mul_ok && B * fN * K > b.opt.conv_fused_max_points)) {
    const ConvKernelMeta* cm = nullptr;
    for (const auto& m : conv_kernel_registry())
      if (m.N == fN && m.TL >= K + 1 && (!cm || m.TL < cm->TL)) cm = &m;
    if (cm) {
      std::vector<float2h> tw;
      for (int q = 1; q < cm->R1; ++q) for (int k = 0; k < cm->R0; ++k) tw.push_back(root_of_unity((int64_t)q * k, fN));
      Step& st = b.push(ST_FFTCONV_FUSED);
      st.variant = cm->id;
      st.p[0] = in; st.p[1] = kern; st.p[2] = out; st.p[3] = b.add_table(tw);
      st.i[0] = B; st.i[1] = K; st.i[2] = ks[0]; st.i[3] = d.conv_mode == MI355FFT_CORRELATION ? 1 : 0;
      st.i[4] = d.input.strided ? d.input.offset_elements : 0;
      st.i[5] = (d.input.strided && d.input.batch_stride_elements > 0) ? d.input.batch_stride_elements : inN;
      st.i[6] = d.input.strided ? d.input.strides[0] : 1;
      if (d.output.strided) {
        st.i[7] = d.output.offset_elements; st.i[8] = kstride;
        st.i[9] = d.output.batch_stride_elements > 0 ? d.output.batch_stride_elements : oN;
        st.i[10] = d.output.strides[0];
      } else if (d.conv_output_layout == MI355FFT_KERNEL_MAJOR) { st.i[7] = 0; st.i[8] = B * oN; st.i[9] = oN; st.i[10] = 1; }
      else { st.i[7] = 0; st.i[8] = oN; st.i[9] = K * oN; st.i[10] = 1; }
      st.f[0] = (float)(1.0 / (double)fN);
      st.grid = (unsigned)std::min<int64_t>(B, (int64_t)b.opt.compute_units * 4);
      b.ir.route += "fftconv-fused[N=" + std::to_string(fN) + ",K=" + std::to_string(K) + ",TL=" + std::to_string(cm->TL) + "] ";
      return MI355FFT_OK;
    }
  }
  const bool embed = d.conv_boundary != MI355FFT_CIRCULAR;
  mi355fft_side_layout dense{};  // strided == 0
  // ---- sides fused into the launches (fftconv.js:353-373): the zero-padded embed of kernels and data, the
  // strided lanes and zeroPad.read ride the first forward axis' loads; zeroPad.write, the crop of the linear modes and the output
  // lanes ride the last inverse axis' store pass — where those axes are line-kernel launches
  int fa = -1, la = -1;
  Builder::nd_first_last(fs, rank, 0, fa, la);
  const auto stride_below = [&](int a) { int64_t S = 1; for (int i = 0; i < a; ++i) S *= fs[i]; return S; };
  const bool small = fN < ((int64_t)1 << 31);
  const bool map_fwd = small && fa >= 0 && b.axis_mappable(fs[fa], stride_below(fa), false);
  const bool map_inv = small && la >= 0 && b.axis_mappable(fs[la], stride_below(la), true);
  // 1. kernels: zero-padded into the FFT domain, transformed once per exec
  PtrRef kf = b.alloc_work((uint64_t)K * fN * 8);
  bool kernel_embed = false;
  for (int i = 0; i < rank; ++i) if (ks[i] != fs[i]) kernel_embed = true;
  int rc;
  if (kernel_embed && map_fwd) {
    SideMap km = Builder::dense_map(fs, rank);
    int64_t st = 1;
    for (int i = 0; i < rank; ++i) { km.stride[i] = st; st *= ks[i]; km.hi[i] = (int)ks[i]; }
    km.batch_stride = kN;
    rc = b.emit_nd(kern, kf, fs, rank, K, false, 1.0f, err, 0, 0, &km, nullptr);
  } else {
    if (kernel_embed) {
      Step& z = b.push(ST_ZERO); z.p[0] = kf; z.i[0] = K * fN * 2; z.grid = b.generic_grid(K * fN * 2);
      b.emit_strided(true, kern, kf, dense, ks, rank, K, fs, zero, fN, 0);
    }
    rc = b.emit_nd(kernel_embed ? kf : kern, kf, fs, rank, K, false, 1.0f, err);
  }
  if (rc) return rc;
  // 1b. 2^20-point circular lines, dense on both sides: forward transform, K products and K inverse transforms in ONE persistent launch
  // whose spectrum tiles never leave the registers (kern_regtile.hpp fft_xcd_conv1m_kernel): 56 + 40 (K - 1) B/point instead of 88 + 56 (K - 1)
  if (b.opt.conv_pipeline && !b.opt.force_generic && b.opt.xcd_fused == 1 && b.opt.xcd_shared && !b.opt.only_pass && rank == 1 && fN == ((int64_t)1 << 20) &&
      d.conv_boundary == MI355FFT_CIRCULAR && !zpad && !d.input.strided && !d.output.strided) {
    const XcdKernelMeta* xm = nullptr;
    for (const auto& m : xcd_kernel_registry()) if (m.real == 4) xm = &m;
    if (xm) {
      // groups per XCD x data lines per round (two slots per line, W and W2), same box, GPoints/s (profiles/r03_fftconv_pipeline_ab.log):
      // 2 x 1 101-104 (8 x 2 x 2 x 8 MiB = the Infinity Cache), 1 x 2 95-97, 2 x 2 92-96, 1 x 1 87-89, 4 x 1 68; three launches 70-73
      const int64_t split = b.opt.xcd_split > 0 ? b.opt.xcd_split : 2;
      const int64_t L = b.opt.xcd_slots > 0 ? b.opt.xcd_slots : 1;
      const PtrRef wslots = b.alloc_work((uint64_t)(16 * 2 * L * split) * fN * 8), ctl = b.alloc_work(40960);
      std::vector<float2h> lo(1024), hi((size_t)(fN >> 10));
      for (int64_t l = 0; l < 1024; ++l) lo[(size_t)l] = root_of_unity(l, fN);
      for (int64_t h = 0; h < (fN >> 10); ++h) hi[(size_t)h] = root_of_unity(h << 10, fN);
      const LineKernelMeta ml = make_meta(0, 1024, 32, 32, 1, 32, true, true, false, false, 0);
      const PtrRef ta = b.line_tables(ml), tlo = b.add_table(lo), thi = b.add_table(hi);
      { Step& z = b.push(ST_ZERO); z.p[0] = ctl; z.i[0] = 9216; z.grid = 1; }
      Step& st = b.push(ST_XCD_FUSED);
      st.variant = xm->id;
      st.p[0] = in; st.p[1] = out; st.p[2] = wslots; st.p[3] = ctl; st.p[4] = PtrRef(BUF_TABLE, 0);
      st.i[0] = B; st.i[1] = fN; st.i[2] = 10; st.i[3] = 1023; st.i[9] = fN;
      const bool kmajor = d.conv_output_layout == MI355FFT_KERNEL_MAJOR;
      st.i[10] = kmajor ? oN : K * oN;        // between data lines
      st.i[17] = kmajor ? B * oN : oN;        // between the kernels of one data line
      st.i[4] = ta.off; st.i[5] = ta.off; st.i[6] = tlo.off; st.i[7] = thi.off; st.i[8] = split; st.i[11] = L; st.i[12] = 0; st.i[13] = b.opt.xcd_spin_limit;
      st.i[14] = kf.off - wslots.off; st.i[15] = K; st.i[16] = d.conv_mode == MI355FFT_CORRELATION ? 1 : 0;
      st.f[0] = (float)(1.0 / (double)fN);
      st.grid = (unsigned)b.opt.compute_units;
      b.ir.route += "fftconv-pipeline[N=1024x1024,K=" + std::to_string(K) + "] ";
      return MI355FFT_OK;
    }
  }
  // 2. data: gather (strided lanes) / embed (linear modes) into the dense FFT domain, forward transform once
  PtrRef xf = b.alloc_work((uint64_t)B * fN * 8);
  const bool side_in = embed || d.input.strided || d.zero_read.enabled;
  const bool fuse_in = side_in && map_fwd;
  SideMap xmap;
  if (fuse_in) {
    xmap = Builder::dense_map(fs, rank);
    int64_t st = 1;
    for (int i = 0; i < rank; ++i) {
      xmap.stride[i] = d.input.strided ? d.input.strides[i] : st;
      st *= d.shape[i];
      int64_t lo = 0, hi = d.shape[i];
      if (d.zero_read.enabled) { lo = std::max(lo, d.zero_read.start[i]); hi = std::min(hi, d.zero_read.end[i]); }
      if (hi < lo) hi = lo;
      xmap.lo[i] = (int)lo; xmap.hi[i] = (int)hi;
    }
    xmap.offset = d.input.strided ? d.input.offset_elements : 0;
    xmap.batch_stride = d.input.strided && d.input.batch_stride_elements > 0 ? d.input.batch_stride_elements : inN;
  } else {
    if (embed) { Step& z = b.push(ST_ZERO); z.p[0] = xf; z.i[0] = B * fN * 2; z.grid = b.generic_grid(B * fN * 2); }
    if (embed || d.input.strided) b.emit_strided(true, in, xf, d.input, d.shape, rank, B, fs, zero, fN, 0);
    else if (d.zero_read.enabled) { Step& c = b.push(ST_COPY); c.p[0] = in; c.p[1] = xf; c.i[0] = B * fN * 8; }
    if (d.zero_read.enabled) { emit_zero_outside(b, xf, d.zero_read, fs, rank, B); b.ir.route += "zero-read "; }
  }
  const bool staged = side_in && !fuse_in;
  // 1-D, power-of-two FFT length with an LDS-resident line kernel: the product with kernel k's spectrum rides the forward FFT's
  // last stage (kern_lines.hpp fft_lines_mul_kernel), one launch per kernel instead of forward FFT + K pointwise passes
  const LineKernelMeta* mul_m = nullptr;
  if (b.opt.conv_lines && !b.opt.force_generic && rank == 1 && is_pow2(fN) && fN <= b.opt.max_line && !(b.opt.xcd_fused == 2 && fN == 4096)) {
    mul_m = find_line_kernel((int)fN, false, false, false, false, 0);
    if (mul_m && (mul_m->lds_bytes == 0 || mul_m->R1 <= 1)) mul_m = nullptr;
  }
  if (!mul_m) {
    rc = fuse_in ? b.emit_nd(in, xf, fs, rank, B, false, 1.0f, err, 0, 0, &xmap, nullptr)
                 : b.emit_nd(staged ? xf : in, xf, fs, rank, B, false, 1.0f, err);
    if (rc) return rc;
  }
  const PtrRef mul_tables = mul_m ? b.line_tables(*mul_m) : PtrRef();
  // 3. per kernel: product, inverse transform scaled by 1/Nfft, crop + place
  PtrRef y = b.alloc_work((uint64_t)B * fN * 8);
  const float inv_n = (float)(1.0 / (double)fN);
  const bool direct_out = !embed && !d.output.strided;   // the inverse FFT can land in the output itself
  const bool side_out = embed || d.output.strided || d.zero_write.enabled || d.conv_output_layout != MI355FFT_KERNEL_MAJOR;
  const bool fuse_out = side_out && map_inv;
  for (int64_t k = 0; k < K; ++k) {
    if (mul_m) {
      Step& st = b.push(ST_LINES);
      st.variant = mul_m->id;
      st.p[0] = staged ? xf : in; st.p[1] = y; st.p[2] = mul_tables; st.p[3] = kf.plus(k * fN * 8);
      const int64_t tiles = (B + mul_m->T - 1) / mul_m->T;
      st.i[0] = tiles; st.i[1] = B; st.i[2] = 1; st.i[3] = fN; st.i[4] = 1; st.i[5] = fN;
      st.i[6] = d.conv_mode == MI355FFT_CORRELATION ? 1 : 0; st.i[9] = 4;
      st.f[0] = 1.0f;
      st.grid = b.lines_grid(*mul_m, tiles);
      if (fuse_in) { st.i[10] = 1; st.imap = xmap; st.imap.ax = 0; st.omap = Builder::dense_map(fs, rank); }
      if (k == 0) b.ir.route += std::string(fuse_in ? "lines-mul-mapped[N=" : "lines-mul[N=") + std::to_string(fN) + "] ";
    } else {
      Step& pm = b.push(ST_POINTWISE);
      pm.p[0] = xf; pm.p[1] = y; pm.p[2] = kf.plus(k * fN * 8);
      pm.i[0] = fN; pm.i[1] = B * fN; pm.i[2] = d.conv_mode == MI355FFT_CORRELATION ? 1 : 0; pm.f[0] = 1.0f;
      pm.grid = b.generic_grid(B * fN);
    }
    if (fuse_out) {
      // output element (i - ooff) of the cropped result, lane of kernel k (fftconv.js:868-871), zeroPad.write on the FFT domain
      SideMap om = Builder::dense_map(fs, rank);
      int64_t st = 1;
      for (int i = 0; i < rank; ++i) {
        om.stride[i] = d.output.strided ? d.output.strides[i] : st;
        st *= os[i];
        om.lo[i] = (int)ooff[i]; om.hi[i] = (int)(ooff[i] + os[i]);
        if (d.zero_write.enabled) { om.zlo[i] = (int)d.zero_write.start[i]; om.zhi[i] = (int)d.zero_write.end[i]; }
      }
      if (d.output.strided) { om.offset = d.output.offset_elements + k * kstride; om.batch_stride = d.output.batch_stride_elements > 0 ? d.output.batch_stride_elements : oN; }
      else if (d.conv_output_layout == MI355FFT_KERNEL_MAJOR) { om.offset = k * B * oN; om.batch_stride = oN; }
      else { om.offset = k * oN; om.batch_stride = K * oN; }
      for (int i = 0; i < rank; ++i) om.offset -= ooff[i] * om.stride[i];
      rc = b.emit_nd(y, y, fs, rank, B, true, inv_n, err, 0, 0, nullptr, &om, out);
      if (rc) return rc;
      continue;
    }
    if (direct_out && d.conv_output_layout == MI355FFT_KERNEL_MAJOR) {
      rc = b.emit_nd(y, out.plus(k * B * oN * 8), fs, rank, B, true, inv_n, err);
      if (rc) return rc;
      if (d.zero_write.enabled) emit_zero_outside(b, out.plus(k * B * oN * 8), d.zero_write, fs, rank, B);
      continue;
    }
    rc = b.emit_nd(y, y, fs, rank, B, true, inv_n, err);
    if (rc) return rc;
    if (d.zero_write.enabled) emit_zero_outside(b, y, d.zero_write, fs, rank, B);
    if (d.output.strided) {
      // lane of kernel k: outputOffset + k*kernelStride + b*batchStride (fftconv.js:868-871)
      b.emit_strided(false, out, y, d.output, os, rank, B, fs, ooff, fN, k * kstride);
    } else {
      mi355fft_side_layout ol{};
      ol.strided = 1;
      int64_t st = 1;
      for (int i = 0; i < rank; ++i) { ol.strides[i] = st; st *= os[i]; }
      if (d.conv_output_layout == MI355FFT_KERNEL_MAJOR) { ol.offset_elements = k * B * oN; ol.batch_stride_elements = oN; }
      else { ol.offset_elements = k * oN; ol.batch_stride_elements = K * oN; }
      b.emit_strided(false, out, y, ol, os, rank, B, fs, ooff, fN, 0);
    }
  }
  if (d.zero_write.enabled && !fuse_out) b.ir.route += "zero-write ";
  b.ir.route += "fftconv[K=" + std::to_string(K) + "] ";
  return MI355FFT_OK;
}

}  // namespace

int build_plan(const mi355fft_plan_desc& desc, const PlannerOptions& opt, PlanIR& out, std::string& err) {
  out = PlanIR();
  out.desc = desc;
  int rc = validate_common(desc, err);
  if (rc) return rc;
  Builder b(out, opt);
  switch (desc.type) {
    case MI355FFT_C2C: rc = build_c2c(desc, b, err); break;
    case MI355FFT_R2C: rc = build_r2c(desc, b, err); break;
    case MI355FFT_C2R: rc = build_c2r(desc, b, err); break;
    case MI355FFT_DCT1: case MI355FFT_DCT2: case MI355FFT_DCT3: case MI355FFT_DCT4:
    case MI355FFT_DST1: case MI355FFT_DST2: case MI355FFT_DST3: case MI355FFT_DST4: rc = build_trig(desc, b, err); break;
    case MI355FFT_FFTCONV: rc = build_fftconv(desc, b, err); break;
    default: err = "type must be one of \"c2c\", \"r2c\", \"c2r\", \"fftconv\" (other createPlan types are outside the MI355X hot path)"; rc = MI355FFT_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  if (out.table.empty()) out.table.push_back(float2h{1, 0});
  return MI355FFT_OK;
}

}  // namespace mi355
