// plan.hpp — host-side planner: turns a createPlan option block into a list of kernel launches.
//
// Replaces the reference's runtime planners for the hot path:
//   src/runtime/plans/c2c.js  (C2CPlan ctor :533-1229, exec :3607-4211)   -> build_c2c
//   src/runtime/plans/r2c.js  (:52-512, core :1518-1557)                  -> build_r2c
//   src/runtime/plans/c2r.js  (:146-, core :1743-1763)                    -> build_c2r
//   src/runtime/plans/fftconv.js (:308-709, exec :1415-1712)              -> build_fftconv
//   src/plan.js factorizeRadices (:20-33) / createFftPlan (:1298-1512)    -> emit_axis + radix factoring
// The reference's ~75 % of planner code that works around WebGPU binding/buffer limits (large_policy,
// segmented_io, BufferView windows) has no counterpart: one HIP allocation spans 288 GB.
//
// This file has no HIP dependency: the same planner drives the HIP launcher (api.hip) and the host
// emulation used by the CPU tests (tests/emu).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi355fft.h"

namespace mi355 {

struct float2h { float x, y; };

struct LineKernelMeta {
  int id;
  int N, R0, R1, R2, T;
  bool in_col, out_col, swap_in, swap_out;
  int twid;
  int threads, lds_bytes, tw_elems;
};
const std::vector<LineKernelMeta>& line_kernel_registry();
const LineKernelMeta* find_line_kernel(int N, bool in_col, bool out_col, bool swap_in, bool swap_out, int twid);

// fused single-launch fftconv kernels (kern_fftconv.hpp): X(N, R0, R1, TL)
#define MI355_CONV_KERNEL_LIST(X) \
  X(64, 8, 8, 4) X(64, 8, 8, 16) X(128, 16, 8, 4) X(128, 16, 8, 16) X(256, 16, 16, 4) X(256, 16, 16, 16) \
  X(512, 32, 16, 4) X(512, 32, 16, 16) X(1024, 32, 32, 4) X(1024, 32, 32, 16)
struct ConvKernelMeta { int id, N, R0, R1, TL; };
// mixed-radix line kernels with compile-time plans (kern_mixed_ct.hpp): id = position in the list
// instances: X(N, lines per workgroup, threads, radices...).  Tile shapes keep every stage's butterfly count (T * N / R) at or
// above the thread count and two workgroups per CU inside the LDS.
#define MI355_MIXEDCT_LIST(X)                                                                                                   \
  X(96, 32, 256, 8, 4, 3) X(192, 16, 256, 8, 8, 3) X(384, 4, 128, 16, 8, 3) X(768, 4, 256, 8, 8, 4, 3) X(1536, 2, 256, 8, 8, 8, 3)       \
  X(3072, 2, 512, 16, 8, 8, 3) X(160, 16, 256, 8, 4, 5) X(320, 8, 256, 8, 8, 5) X(640, 4, 256, 8, 8, 2, 5) X(1280, 2, 256, 8, 8, 4, 5)  \
  X(2560, 1, 256, 8, 8, 8, 5) X(1000, 4, 256, 8, 5, 5, 5) X(2000, 4, 512, 16, 5, 5, 5) X(3000, 2, 512, 8, 5, 5, 5, 3)                   \
  X(105, 32, 256, 7, 5, 3) X(1001, 8, 512, 13, 11, 7) X(360, 8, 256, 8, 5, 3, 3) X(1920, 4, 512, 16, 8, 5, 3) X(2187, 1, 256, 3, 3, 3, 3, 3, 3, 3) \
  X(500, 8, 256, 4, 5, 5, 5) X(1500, 2, 256, 4, 5, 5, 5, 3) X(120, 32, 256, 8, 5, 3) X(240, 16, 256, 4, 4, 5, 3) X(480, 8, 256, 8, 4, 5, 3) X(720, 4, 256, 16, 5, 3, 3) X(1440, 2, 256, 8, 4, 5, 3, 3) \
  /* r03: 2187 = 9*9*9*3 (radix 9 by direct evaluation with literal roots: 4 LDS passes instead of 7): 180 -> 194 GPoints/s.  Listed after the plan it \
     replaces: the registry walk takes the last match (MI355FFT_MIXED_CT=2: the first).  3000 = 8*25*15 and 1500 = 4*25*15 measured SLOWER than their \
     five-stage plans (168 vs 192, 145 vs 203: profiles/r03_mixed_radix_composite.log) and are not instantiated */ \
  X(2187, 2, 512, 9, 9, 9, 3)

struct MixedCtMeta { int id, N, T, threads, lds_bytes; std::vector<int> radices; };
const std::vector<MixedCtMeta>& mixedct_registry();

// XCD-fused four-step kernels (kern_xcd.hpp): X(N1, R0a, R1a, R2a, Ta, N2, R0b, R1b, R2b, Tb); the tile widths are chosen so
// that both passes use the same workgroup size; each is built forward and inverse.  (64 x 64 exists for the CPU
// emulation tests.)  N = 2^15, 2^16, 2^17 (solo mode: one workgroup per transform), 2^18, 2^19, 2^20, 2^21 (the c2c half of
// r2c/c2r N = 2^22).  2^22 = 2048 x 2048 (8-wide tiles on both
// passes, 104 B of scratch) was built and measured slower than the two-launch route (104 vs 112 GPoints/s): not instantiated.
#define MI355_XCD_KERNEL_LIST(X) \
  X(64, 8, 8, 1, 16, 64, 8, 8, 1, 16) X(128, 16, 8, 1, 32, 256, 16, 16, 1, 16) X(256, 16, 16, 1, 16, 256, 16, 16, 1, 16) \
  X(256, 16, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 32, 1024, 32, 32, 1, 16) \
  X(1024, 32, 32, 1, 16, 1024, 32, 32, 1, 16) X(1024, 32, 32, 1, 16, 2048, 32, 32, 2, 8)
// (2^22 = 2048 x 2048 on LDS-resident 8-line tiles, re-measured in r03 with the one-slot grouping: 89-105 vs 159 GPoints/s for the register tiles: profiles/r03_regtile_ab.log)
// (2^20 with 1024 threads and 16-point register stages — 16 waves per CU instead of 8, 128 VGPRs, 56 B of scratch — measured
// 165-176 vs 186 GPoints/s: profiles/r02_xcd_2p21_orientation.log; not instantiated)
// (2^21 as 2048 x 1024 — 64-byte segments on the input side instead of the output side — measured 138 vs 167 GPoints/s for
// 1024 x 2048 and 189 vs 219 on config 5: profiles/r02_xcd_2p21_orientation.log; not instantiated)
// r2c and c2r variants (kern_xcd_real.hpp), same parameters: a real line of N1*N2 points; r2c forward, c2r inverse.  2^12 (test instance), 2^15 .. 2^17 (solo mode), 2^18 .. 2^21.
// (2^22 = 2048 x 2048 with 8-wide tiles on both passes was built and measured slower than the half-length route over the
// fused c2c kernel — 153-163 vs 183 G real points/s — and is not instantiated.)
#define MI355_XCD_R2C_KERNEL_LIST(X) \
  X(64, 8, 8, 1, 16, 64, 8, 8, 1, 16) X(128, 16, 8, 1, 32, 256, 16, 16, 1, 16) X(256, 16, 16, 1, 16, 256, 16, 16, 1, 16) \
  X(256, 16, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 32, 1024, 32, 32, 1, 16) \
  X(1024, 32, 32, 1, 16, 1024, 32, 32, 1, 16) X(1024, 32, 32, 1, 16, 2048, 32, 32, 2, 8)
// c2r: 2^21 (1024 x 2048, 396 B of scratch per lane) measured 140 vs 215 G real points/s for the half-length route: not instantiated
#define MI355_XCD_C2R_KERNEL_LIST(X) \
  X(64, 8, 8, 1, 16, 64, 8, 8, 1, 16) X(128, 16, 8, 1, 32, 256, 16, 16, 1, 16) X(256, 16, 16, 1, 16, 256, 16, 16, 1, 16) \
  X(256, 16, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 32, 1024, 32, 32, 1, 16) \
  X(1024, 32, 32, 1, 16, 1024, 32, 32, 1, 16)
// 2-D c2c of an [N1][N0] array (axis 0 = N0 fastest): X(N1, radices, Ta, N0, radices, Tb) — columns of N1 (pass A), barrier,
// rows of N0 with a ROW kernel (natural order out, no four-step roots); forward and inverse.  Tile widths chosen so that both
// passes use the same workgroup size (256 threads for the 256/512 planes, 512 with a 1024 side)
#define MI355_XCD_2D_KERNEL_LIST(X) \
  X(256, 16, 16, 1, 16, 256, 16, 16, 1, 16) X(512, 32, 16, 1, 16, 512, 32, 16, 1, 16) X(1024, 32, 32, 1, 16, 1024, 32, 32, 1, 16) \
  X(256, 16, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 16, 256, 16, 16, 1, 16) \
  X(1024, 32, 32, 1, 16, 512, 32, 16, 1, 32) X(512, 32, 16, 1, 32, 1024, 32, 32, 1, 16)
// register-tile instances (kern_regtile.hpp, r03): X(N1) — N = N1 x 2048 with the 2048-point side(s) on 16-line tiles held in
// registers (128-byte segments where the LDS-resident 8-line tiles above move 64-byte ones); forward and inverse.  Listed AFTER
// the LDS-resident instances of the same size so that a registry walk finds them last (PlannerOptions::xcd_rt).
#define MI355_XCD_RT_KERNEL_LIST(X) X(1024) X(2048)
// VIEW instantiations of the LDS-resident fused kernel (rank-1 ioView / zeroPad ranges as load / store predicates): 2^17, 2^18, 2^19, 2^21, forward and inverse
// (2^20 has its own in kern_regtile.hpp).  Same parameters as MI355_XCD_KERNEL_LIST.
#define MI355_XCD_VIEW_KERNEL_LIST(X) \
  X(256, 16, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 16, 512, 32, 16, 1, 16) X(512, 32, 16, 1, 32, 1024, 32, 32, 1, 16) X(1024, 32, 32, 1, 16, 2048, 32, 32, 2, 8)
struct XcdKernelMeta { int id, N1, N2, ra[3], rb[3], ta, tb; bool inverse; int threads, lds_bytes; int real; int rt; };   // real: 0 c2c, 1 r2c, 2 c2r, 3 two-dimensional c2c, 4 fftconv pipeline; rt: 1 register-tile instance (2048-point sides), 2 the two-workgroups-per-CU 1024 x 1024
const std::vector<XcdKernelMeta>& xcd_kernel_registry();
const std::vector<ConvKernelMeta>& conv_kernel_registry();

enum BufId : int { BUF_NONE = -1, BUF_INPUT = 0, BUF_OUTPUT = 1, BUF_WORK = 2, BUF_KERNEL = 3, BUF_TABLE = 4 };
struct PtrRef {
  int buf = BUF_NONE;
  int64_t off = 0;  // bytes
  PtrRef() {}
  PtrRef(int b, int64_t o) : buf(b), off(o) {}
  PtrRef plus(int64_t bytes) const { return PtrRef(buf, off + bytes); }
  bool same(const PtrRef& o) const { return buf == o.buf && off == o.off; }
};

// A side of a line-kernel launch as a map from logical coordinates to physical storage (layout strides,
// ioView, zeroPad fused into the first load / last store instead of separate gather / embed / zero / extract / scatter passes).
// Line G of the launch (all dims except `ax` in axis-0-fastest order, then the batch) and position idx along `ax`:
//   reads : inside [lo, hi) on every dim -> phys[offset + sum i_d * stride_d + b * batch_stride], else 0
//   writes: inside [lo, hi) on every dim -> stored, with the value replaced by 0 outside [zlo, zhi); else not stored
// a three-stage line kernel keeps its last stage table in LDS up to this many bytes (LineCfg::TW2_IN_LDS, plan.cpp make_meta)
#ifndef MI355_TW2_LDS_MAX
#define MI355_TW2_LDS_MAX (32 * 1024)
#endif

struct SideMap {
  int rank = 0, ax = 0;
  int dims[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  int lo[8] = {0}, hi[8] = {0}, zlo[8] = {0}, zhi[8] = {0};
  long long stride[8] = {0};
  long long offset = 0, batch_stride = 0;
};

enum StepKind : int {
  ST_LINES, ST_STAGE, ST_R2C_POST, ST_C2R_PRE, ST_REAL_TO_COMPLEX, ST_COMPLEX_TO_REAL, ST_PACK_HALF, ST_UNPACK_HERM,
  ST_POINTWISE, ST_GATHER, ST_SCATTER, ST_ZERO, ST_COPY, ST_SCALE, ST_FFTCONV_FUSED, ST_CHIRP_PRE, ST_CHIRP_POST, ST_ZERO_OUTSIDE, ST_XCD_FUSED, ST_LINES_MIXED, ST_TRIG_PRE, ST_TRIG_POST, ST_XCD_RES
};

// One recorded launch, pointers still symbolic.  Scalar fields are kind-specific (see dispatch.hpp).
struct Step {
  StepKind kind;
  int variant = 0;           // ST_LINES: registry id; ST_STAGE: radix
  PtrRef p[5];               // kind-specific pointer slots
  int64_t i[20] = {0};     // kind-specific integers
  float f[2] = {1.0f, 1.0f}; // kind-specific floats
  int64_t shape[8] = {0}, sa[8] = {0}, sb[8] = {0};  // ST_GATHER / ST_SCATTER
  SideMap imap, omap;        // ST_LINES with i[10] != 0: mapped sides (kern_lines.hpp fft_lines_mapped_kernel)
  unsigned grid = 1;
};

struct PlanIR {
  mi355fft_plan_desc desc;
  std::vector<Step> steps;
  std::vector<float2h> table;     // all twiddle tables, uploaded once at plan creation
  uint64_t work_bytes = 0;        // plan.getWorkspaceSizeBytes()
  uint64_t in_bytes = 0, out_bytes = 0, kernel_bytes = 0;  // minimum extents the exec buffers must cover
  std::string route;              // human-readable route description (plan_describe)
};

struct PlannerOptions {
  uint64_t chunk_bytes = 1ull << 30;   // two-pass: bytes of inter-pass intermediate per launch pair (measured: larger is faster)
  int compute_units = 256;
  int fuse_views = 1;                  // c2c: strided layouts / ioView / zeroPad ride the first load and last store of the line kernels where the axis route allows (0: staging passes)
  int lines_tiles_per_wg = 0;          // line kernels: 0 = resident grid (CUs x workgroups per CU) walking the tiles; k > 0 = one workgroup per k tiles
  int force_generic = 0;               // tests: route everything through the global-memory stage kernels
  int xcd_fused = 1;                   // N = N1*N2 with an XCD-fused kernel available: both passes in one persistent launch
  int xcd_shared = 1;                  // 0: no kernel whose workgroups wait for each other (shared-mode XCD kernels); solo instances stay
  unsigned xcd_spin_limit = 4000000u;  // polls before a bounded wait of the XCD kernels gives up (sticky error); tests force 1
  int line32k = 1;                     // N = 2^15 c2c lines in one workgroup (kern_line32k.hpp) instead of the solo four-step
  int trig_alt = 1;                    // one-launch DCT / DST: use the alternate ROW shapes where the registry has them (ROW_ALT_TRIG)
  int xcd_res = 0;                     // N = 2^20: XCD-resident kernel (kern_xcd_res.hpp): 1 on, 2 = its data-movement skeleton without arithmetic (measurement only)
  int xcd_res_depth = 4;               // exchange channels in flight per XCD (1, 2, 4): 1 MiB of L2-resident buffer each
  int xcd_split = 0;                   // groups per XCD in the fused kernels (1..8); 0 = chosen per plan from the workspace footprint
  int xcd_rt = 1;                      // 2048-point sides on register tiles (kern_regtile.hpp) where an instance exists (0: the LDS-resident 8-line tiles / two-pass route;
                                       // c2c 2^21 stays on the LDS-resident instance unless 2: 1024 x 2048 on register tiles, 3: 2048 x 1024 on register tiles)
  int xcd_hx = 2;                      // N = 2^20: 2 = 32-line register tiles (kern_regtile.hpp fft_xcd_rt1k_kernel; r03 default: 199 vs 194 GPoints/s), 1 = two workgroups per CU on 16-line
                                       // register tiles (fft_xcd_hx_kernel: 171), 0 = the LDS-resident fused kernel (kern_xcd.hpp)
  int xcd_2d = 1;                      // 2-D c2c planes with an instance: both axes in one fused launch
  int xcd_r2c = 1;                     // r2c: real four-step kernel where an instance exists (0: half-length c2c + split)
  int solo_cap_mb = 256;               // solo mode: all workgroups' workspace slots together (MiB) = the Infinity Cache (r02: 2^16 203 vs 189 GPoints/s with 1024; below 256 occupancy collapses)
  int solo_max_kb_2d = 1024;           // 2-D planes up to this size run in solo mode (unchanged from round 1)
  int solo_max_kb = 512;               // c2c transforms (and 2-D planes) up to this size run in solo mode; r2c up to half of it, c2r up to this many KB of REAL line
                                       // (r02, same box: c2c 2^17 shared 171 vs solo 147, r2c 2^17 281 vs 250, c2r 2^17 solo 318 vs 278, c2c 2^16 solo 206 vs 156)
  int xcd_slots = 0;                   // workspace slots per group: 0 = per route (r03: one slot and twice the groups; the 2048-point register-tile instances two), 1: two barriers per transform, 2: one barrier
  int mixed_ct = 1;                    // mixed-radix lengths with a compile-time-plan instance (kern_mixed_ct.hpp) use it (dense lines)
  int mixed_lds_kb = 0;                // experiments: LDS per workgroup of the mixed-radix line kernel (0 = per-length rule)
  int mixed_threads = 256;
  int64_t conv_fused_max_points = (int64_t)1 << 20;   // fftconv-fused (one launch, latency route) up to this many points B*N*K; above: forward-mul + inverse line launches
  int conv_pipeline = 1;               // fftconv of 2^20-point circular dense lines: forward, products and inverses in one persistent launch (kern_regtile.hpp fft_xcd_conv1m_kernel)
  int conv_lines = 1;                  // fftconv: kernel-spectrum product fused behind the forward line FFT (1-D, power-of-two FFT length <= max_line)
  int trig_fused = 1;                  // dct2 / dst2 of dense lines (half length a line-kernel size): permutation + real FFT + phase in one launch
  int trig_real = 1;                   // dct2/dst2/dct3/dst3 along a dense even axis through a real FFT of length N (kern_trig.hpp)
  int lines_c2r = 1;                   // c2r twin (pre-split from global into LDS before the first stage): half lengths <= 16384 (3: <= 8192, the round-1 choice)
                                       // (N = 256: 528 vs 133 G real points/s, 1024: 471 vs 243, 2^14: 312 vs 270); 2 forces it at 2^15 too
  int lines_r2c = 1;                   // r2c with a half length of 64..max_line: split fused into the line kernel
  int max_line = 16384;                // longest power-of-two line given to a single workgroup (4096: N = 8192, 16384 take the four-step routes)
  int mixed_lines = 1;                 // mixed-radix lengths <= 4096: one LDS line kernel instead of one global pass per radix
  int only_pass = 0;                   // measurement aid (bench.py per-kernel timing): 1 = emit pass A only, 2 = pass B only
};
PlannerOptions planner_options_from_env();

// returns MI355FFT_OK or an error status with `err` filled (message style follows the reference's throws)
int build_plan(const mi355fft_plan_desc& desc, const PlannerOptions& opt, PlanIR& out, std::string& err);

// radix factorisation of the generic route: greedy largest-first over {32,16,8,4,2,13,11,7,5,3}
// (superset of src/plan.js:20-33's {13,11,8,7,5,4,3,2}); empty when n has another prime factor
std::vector<int> factorize_radices(int64_t n, int max_radix = 32);

// e^{-2 pi i m / M} rounded to f32 from an 80-bit evaluation
float2h root_of_unity(int64_t m, int64_t M);

}  // namespace mi355
