// kern_lines.hpp — LDS-resident Stockham line FFTs for gfx950: the hot kernels.
//
// Replaces, for power-of-two line lengths N <= 4096:
//   * the per-stage dispatch loop of FftPlan.exec (src/plan.js:1233-1273): S full HBM round trips, one
//     compute pass per radix stage, become ONE kernel that reads each point once and writes it once;
//   * generateStockhamRadixStageWGSL (src/kernels/stockham_stage.js:76-103): same autosort index math
//       in[j + q*N/R]  ->  out[(j / Ns)*Ns*R + (j % Ns) + q*Ns],  twiddle e^{-2 pi i q (j % Ns)/(Ns*R)}
//     but with the R inputs of a butterfly held by one lane, roots from f64-built tables staged in LDS
//     (the reference calls cos/sin per thread per stage in f32 and chains w *= w1: ~1e-4 accuracy);
//   * generateSubgroupPow2FftWGSL (src/kernels/subgroup_pow2_fft.js:8-133): no bit-reverse pass exists
//     here at all — Stockham stages leave natural order;
//   * generateTransposeComplex2DWGSL (src/kernels/transpose.js:1-51) and the per-element copies of the
//     axis-0 two-step route (src/plan.js:375-384, 456-595; twiddle :114-153): the four-step transposes are
//     folded into the address maps of the two passes (column-tile loads/stores of >= 128 B segments), and
//     the e^{-2 pi i k1 n2/N} twiddle is fused into pass A's last stage.
//
// One workgroup owns a tile of T lines.  Stage 0 reads global memory, the last stage writes global
// memory, the stages between exchange through LDS.  Thread->(line, butterfly) maps:
//   ROW map  (lines contiguous in memory):      line = t / TPL, u = t % TPL  -> lanes walk along a line
//   COL map  (lines are columns, T adjacent):   line = t % T,   u = t / T    -> lanes walk across lines
// so that every global access is a run of consecutive lanes over consecutive addresses.
//
// LDS layouts (bank analysis in DESIGN.md):
//   COL in + COL out : idx-major  [idx][T]      conflict-free without padding for T >= 16
//   otherwise        : line-major [line][PITCH] idx padded by idx >> log2(R0); PITCH % 32 == 2 so the
//                      COL-mapped reads of a ROW-in/COL-out (transposing) kernel are conflict-free.
#pragma once
#include "platform.hpp"
#include "radix.hpp"
#include "plan.hpp"

// streamed-once global accesses of the line kernels: 0 = default cache policy, 1 = nontemporal loads + stores
#ifndef MI355_NT_GLOBAL
#define MI355_NT_GLOBAL 0
#endif
#ifndef MI355_LINES_PREFETCH
#define MI355_LINES_PREFETCH 1
#endif

namespace mi355 {

// NT: per-call-site override (the XCD-fused kernel streams x and the output past the L2 it wants to keep W in)
template <bool NT = false> MI_DEV cf ld_stream(const cf* p) {
  if constexpr (NT || MI355_NT_GLOBAL) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT = false> MI_DEV void st_stream(cf* p, cf v) {
  if constexpr (NT || MI355_NT_GLOBAL) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// four-step twiddle e^{-2 pi i (line index in its group) * (element index) / Ntot}:
//   TWID_FOURSTEP_OUT  multiplied into the last stage's outputs (LO table staged in LDS, HI from global)
//   TWID_FOURSTEP_IN   multiplied into the first stage's inputs from per-thread registers that are computed
//                      once per kernel launch when every tile of a workgroup has the same position inside its
//                      group (grid * T is a multiple of the group), else once per tile
enum : int { TWID_NONE = 0, TWID_FOURSTEP_OUT = 1, TWID_FOURSTEP_IN = 2,
             // no roots; COL sides whose group width S is NOT a multiple of T (e.g. the 513 packed bins of a 2-D r2c): tiles are
             // numbered per group (fs_group = tiles per group), the last tile of a group is ragged
             COL_RAGGED = 3,
             // registry tag only: an alternate shape of a plain ROW configuration, picked by the one-launch DCT / DST routes
             // (more, smaller workgroups per CU; the kernels see TWID_NONE)
             ROW_ALT_TRIG = 4 };

struct LineArgs {
  const cf* in;
  cf* out;
  const cf* tw;      // stage tables, concatenated: stage s>=1 is [R_s - 1][Ns_prev] (rows q=1.., k fastest)
  const cf* tw_lo;   // four-step: W_Ntot^l, l < 2^fs_shift
  const cf* tw_hi;   // four-step: W_Ntot^(h << fs_shift)
  long long num_tiles;
  long long num_lines;   // global lines with G >= num_lines are padding of the last tile: never loaded or stored
  // ROW side: element p of global line G at G*row_stride + p.
  // COL side: G -> (o = G / S, i = G % S); element p at o*outer_stride + i + p*S.
  long long in_S, in_outer_stride;
  long long out_S, out_outer_stride;
  float scale;
  int fs_shift;
  unsigned fs_lo_mask;
  int real_mode;         // 4: fft_lines_mul_kernel (tw_lo = kernel spectrum, fs_shift != 0: conjugate it);  1: fft_lines_r2c_kernel (real line read as complex pairs, split fused behind the last stage); 2: fft_lines_c2r_kernel
  long long fs_group;    // TWID_FOURSTEP_IN: lines per group (line index inside the group = G % fs_group); COL_RAGGED: tiles per group
  int v_in_lo, v_in_hi, v_out_lo, v_out_hi, v_zlo, v_zhi;   // VIEW instantiations of stage_read / stage_compute_write (kern_xcd.hpp fused kernels): rank-1 ranges of a four-step line
  int mapped;            // fft_lines_mapped_kernel: both sides go through imap / omap (in / out are the buffers' bases)
  SideMap imap, omap;
};

#ifndef MI355_NT_MAX_N
#define MI355_NT_MAX_N 16384
#endif
template <int N_, int R0_, int R1_, int R2_, int T_, bool IN_COL_, bool OUT_COL_, bool SWAP_IN_, bool SWAP_OUT_, int TWID_>
struct LineCfg {
  static constexpr int N = N_, R0 = R0_, R1 = R1_, R2 = R2_, T = T_;
  static constexpr bool IN_COL = IN_COL_, OUT_COL = OUT_COL_, SWAP_IN = SWAP_IN_, SWAP_OUT = SWAP_OUT_;
  static constexpr int TWID = TWID_ == ROW_ALT_TRIG ? (int)TWID_NONE : TWID_;
  static_assert(R0 * R1 * R2 == N, "radix product");
  static constexpr int NSTAGES = (R1 == 1) ? 1 : (R2 == 1 ? 2 : 3);
  static_assert(R1 > 1 || R2 == 1, "R2 needs R1");
  static constexpr int RMAX = R0 > R1 ? (R0 > R2 ? R0 : R2) : (R1 > R2 ? R1 : R2);
  static constexpr int E = RMAX;            // complex values per thread
  static constexpr int TPL = N / RMAX;      // threads per line
  static constexpr int THREADS = T * TPL;
  static constexpr bool IDX_MAJOR = IN_COL && OUT_COL;
  static constexpr int PADSH = ilog2(R0);
  static constexpr int PITCH_RAW = N + (N >> PADSH);
  // PITCH == PMOD (mod 32) complex: COL-mapped ds_read_b64 of a transposing kernel then spreads a 32-lane
  // group (T lines x 32/T butterflies) over all 64 banks: line step = 2*PMOD dwords, butterfly step = 2
  static constexpr int PMOD = (!IN_COL && OUT_COL && T <= 32) ? 32 / T : 2;
  static constexpr int PITCH = ((PITCH_RAW + 31) / 32) * 32 + PMOD;
  static constexpr int DATA_ELEMS = NSTAGES == 1 ? 0 : (IDX_MAJOR ? N * T : T * PITCH);
  // stage tables hold rows q = 1..R-1 only (row 0 is all ones): stage 1 [R1-1][R0], stage 2 [R2-1][R0*R1]
  static constexpr int TW1_ELEMS = NSTAGES >= 2 ? (R1 - 1) * R0 : 0;
  static constexpr int TW2_ELEMS = NSTAGES == 3 ? (R2 - 1) * R0 * R1 : 0;
  static constexpr int TW_ELEMS = TW1_ELEMS + TW2_ELEMS;
  // the last table of a three-stage plan has about N entries: beyond 32 KB (N = 8192, 16384) it stays in global memory,
  // where its reads are L1/L2 hits, so that the line itself still fits the LDS with room for a second workgroup
  // (the alternate DCT shapes trade the table's LDS residence for workgroups per CU already above 8 KB)
  static constexpr bool TW2_IN_LDS = TW2_ELEMS * 8 <= (TWID_ == ROW_ALT_TRIG ? 8 * 1024 : MI355_TW2_LDS_MAX);
  static constexpr int TW_LDS_ELEMS = TW1_ELEMS + (TW2_IN_LDS ? TW2_ELEMS : 0);
  static constexpr int LO_ELEMS = TWID == TWID_FOURSTEP_OUT ? 1024 : 0;
  static constexpr int LDS_BYTES = (DATA_ELEMS + TW_LDS_ELEMS + LO_ELEMS) * 8;
  // every line lives in one wave and only ROW maps are used: exchanges need no workgroup barrier
  static constexpr bool WAVE_LOCAL = !IN_COL && !OUT_COL && TPL <= 64 && (64 % TPL) == 0;
  // software pipelining of the resident workgroup's tile loop: the first-stage loads of the NEXT tile are issued before the
  // current tile is computed and stored, so the memory system always has this workgroup's reads in flight (E more complex
  // registers per thread).  ROW kernels of 64..2048 points: their register budget allows it at the occupancy the LDS permits.
  // plain ROW transforms of 256 points and more stream every byte once: nontemporal loads / stores measured +4...+6 % on the
  // one-shot and on the resident grids (N = 64: -10 %; profiles/r02_lines_nt_ab.log).  The XCD kernels pick their own policy per
  // call site (their intermediate must stay cached), the fused r2c / c2r / product variants keep the default.
  static constexpr bool STREAM_NT = !IN_COL && !OUT_COL && TWID == TWID_NONE && N >= 256 && N <= MI355_NT_MAX_N;
  static constexpr bool PREFETCH = MI355_LINES_PREFETCH && !IN_COL && !OUT_COL && TWID == TWID_NONE && NSTAGES == 2 && N >= 64 && N <= 2048;
  static_assert(THREADS <= 1024, "workgroup too large");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

// SWZ16 (three-stage ROW plans with a 16-point first stage: 2048 = 16*16*8, 4096 = 16*16*16): the padded form makes the 32-lane
// groups of the consecutive-index ds_read_b64 of stages 1 and 2 hit one bank pair twice (idx + idx/16 repeats mod 32 every 32
// elements: 17-23 % of the LDS cycles of these shapes were conflict cycles, profiles/r02_lds_bank_conflicts.log).  Rotating the
// low four index bits by the 16-block number instead keeps every aligned block of 16 a permutation of itself — consecutive
// indices stay conflict-free — and still spreads the stride-16 writes of stage 0 (fixed q, lanes over u: low bits q + u).
#ifndef MI355_LDS_SWZ16
#define MI355_LDS_SWZ16 0
#endif
template <class C> MI_DEV int lds_index(int line, int idx) {
  if constexpr (C::IDX_MAJOR) return idx * C::T + line;
  else if constexpr (MI355_LDS_SWZ16 && !C::IN_COL && !C::OUT_COL && C::NSTAGES == 3 && C::R0 == 16) return line * C::PITCH + ((idx & ~15) | ((idx + (idx >> 4)) & 15));
  else return line * C::PITCH + idx + (idx >> C::PADSH);
}

template <class C> MI_DEV void lines_sync() {
  if constexpr (C::WAVE_LOCAL) { MI_WAVE_SYNC(); } else { __syncthreads(); }
}

// radix / Ns_prev of stage S
template <class C, int S> struct StageInfo {
  static constexpr int R = S == 0 ? C::R0 : (S == 1 ? C::R1 : C::R2);
  static constexpr int NSP = S == 0 ? 1 : (S == 1 ? C::R0 : C::R0 * C::R1);
  static constexpr int NB = C::E / R;                      // butterflies per thread
  static constexpr int TW_OFF = S <= 1 ? 0 : C::TW1_ELEMS;  // offset of this stage's table in the LDS copy
  static constexpr bool FIRST = S == 0;
  static constexpr bool LAST = S == C::NSTAGES - 1;
  // thread map: stage 0 follows the input side, every later stage the output side
  static constexpr bool COLMAP = FIRST ? C::IN_COL : C::OUT_COL;
};

template <class C, int S> MI_DEV void thread_map(int t, int& line, int& u) {
  if constexpr (StageInfo<C, S>::COLMAP) { line = t % C::T; u = t / C::T; }
  else { line = t / C::TPL; u = t % C::TPL; }
}

// Addressing.  A tile never straddles an outer group (the planner guarantees S % T == 0 on COL sides), so
// everything tile-dependent is wave-uniform and lives in SGPRs; a lane adds only a 32-bit element offset
//   voff(line, idx) = line*ls + idx*es      ROW side: ls = row_stride, es = 1;   COL side: ls = 1, es = S
// and tiles span < 2^32 bytes, which lets the compiler use the saddr + 32-bit voffset load/store forms.
template <bool COL> MI_DEV long long tile_base(long long G0, long long S, long long outer_stride) {
  if constexpr (COL) { const long long o = G0 / S; return o * outer_stride + (G0 - o * S); }
  else return G0 * outer_stride;
}

// per-thread four-step roots for the first stage's inputs (TWID_FOURSTEP_IN)
template <class C>
MI_DEV void fourstep_in_roots(cf (&fsw)[C::E], const LineArgs& a, long long tile, int t) {
  using I = StageInfo<C, 0>;
  int line, u; thread_map<C, 0>(t, line, u);
  const unsigned gi = (unsigned)((tile * C::T + line) % a.fs_group);
#pragma unroll
  for (int b = 0; b < I::NB; ++b) {
#pragma unroll
    for (int q = 0; q < I::R; ++q) {
      const unsigned idx = (unsigned)(u + b * C::TPL + q * (C::N / I::R));
      const unsigned m = gi * idx;
      fsw[b * I::R + q] = cmul(a.tw_hi[m >> a.fs_shift], a.tw_lo[m & a.fs_lo_mask]);
    }
  }
}

// VIEW (column-mapped first stage of a fused four-step kernel): element idx * in_S + (column of the line) of the transform is read inside
// [v_in_lo, v_in_hi) and is 0 elsewhere (rank-1 ioView.input / zeroPad.read)
template <class C, int S, bool NT = false, bool VIEW = false>
MI_DEV void stage_read(cf (&v)[C::E], const LineArgs& a, long long tile, int t, const cf* lds) {
  using I = StageInfo<C, S>;
  int line, u; thread_map<C, S>(t, line, u);
  if constexpr (I::FIRST) {
    const long long G0 = tile * C::T;
    const cf* p;
    long long live_lines;
    if constexpr (C::TWID == COL_RAGGED) {
      const long long o = tile / a.fs_group, j0 = (tile - o * a.fs_group) * C::T;   // group, first line of the tile inside it
      p = a.in + o * a.in_outer_stride + j0;
      live_lines = a.in_S - j0;
    } else {
      p = a.in + tile_base<C::IN_COL>(G0, a.in_S, a.in_outer_stride);
      // padding lines of the last tile re-read its last live line (loads stay in bounds, stores are masked)
      live_lines = a.num_lines - G0;
    }
    const unsigned ls = C::IN_COL ? 1u : (unsigned)a.in_outer_stride;
    const unsigned es = C::IN_COL ? (unsigned)a.in_S : 1u;
    const int lclamp = (long long)line < live_lines ? line : (int)live_lines - 1;
    const unsigned voff = (unsigned)lclamp * ls + (unsigned)u * es;
#pragma unroll
    for (int b = 0; b < I::NB; ++b) {
#pragma unroll
      for (int q = 0; q < I::R; ++q) {
        const cf* pq = p + (unsigned)(b * C::TPL + q * (C::N / I::R)) * es;   // uniform
        if constexpr (VIEW) {
          static_assert(C::IN_COL, "views ride the column-mapped pass A");
          const int i = (int)((unsigned)(u + b * C::TPL + q * (C::N / I::R)) * es + (unsigned)(G0 % a.in_S) + (unsigned)line);
          cf xv = {0.0f, 0.0f};
          if (i >= a.v_in_lo && i < a.v_in_hi) xv = pq[voff];
          v[b * I::R + q] = cswap_if<C::SWAP_IN>(xv);
        } else
        v[b * I::R + q] = cswap_if<C::SWAP_IN>(ld_stream<NT>(pq + voff));
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < I::NB; ++b) {
      const int j = u + b * C::TPL;
#pragma unroll
      for (int q = 0; q < I::R; ++q) v[b * I::R + q] = lds[lds_index<C>(line, j + q * (C::N / I::R))];
    }
  }
}

// KEEP_IN_LDS: the last stage leaves the finished lines in LDS (same layout as the exchanges) instead of storing them to
// global memory — for kernels that post-process a whole line before it leaves the workgroup (kern_xcd_real.hpp)
// MUL: the finished outputs are multiplied by the spectrum a.tw_lo[k] (conjugated when a.fs_shift != 0) on their way out — the
// pointwise product of fftconv folded into the last stage's register store (fft_lines_mul_kernel)
template <class C, int S, bool NT = false, bool KEEP_IN_LDS = false, bool MUL = false, bool VIEW = false>
MI_DEV void stage_compute_write(cf (&v)[C::E], const LineArgs& a, long long tile, int t, cf* lds, const cf* tw_lds, const cf* lo_lds) {
  using I = StageInfo<C, S>;
  constexpr bool TO_GLOBAL = I::LAST && !KEEP_IN_LDS;
  int line, u; thread_map<C, S>(t, line, u);
  cf* po = nullptr;
  unsigned ls = 1, es = 1, gi = 0;
  bool live = true;
  if constexpr (TO_GLOBAL) {
    const long long G0 = tile * C::T;
    if constexpr (C::TWID == COL_RAGGED) {
      const long long o = tile / a.fs_group, j0 = (tile - o * a.fs_group) * C::T;
      po = a.out + o * a.out_outer_stride + j0;
      live = (long long)line < a.out_S - j0;
    } else {
      po = a.out + tile_base<C::OUT_COL>(G0, a.out_S, a.out_outer_stride);
      live = (long long)line < a.num_lines - G0;
    }
    ls = C::OUT_COL ? 1u : (unsigned)a.out_outer_stride;
    es = C::OUT_COL ? (unsigned)a.out_S : 1u;
    if constexpr (C::TWID == TWID_FOURSTEP_OUT) gi = (unsigned)(G0 % a.out_S) + (unsigned)line;   // column index n2 of this line
  }
#pragma unroll
  for (int b = 0; b < I::NB; ++b) {
    const int j = u + b * C::TPL;
    const int k = j % I::NSP;
    cf w[I::R];
#pragma unroll
    for (int q = 0; q < I::R; ++q) w[q] = v[b * I::R + q];
    if constexpr (!I::FIRST) {
#pragma unroll
      for (int q = 1; q < I::R; ++q) {
        if constexpr (S == 2 && !C::TW2_IN_LDS) w[q] = cmul(w[q], a.tw[I::TW_OFF + (q - 1) * I::NSP + k]);
        else w[q] = cmul(w[q], tw_lds[I::TW_OFF + (q - 1) * I::NSP + k]);
      }
    }
    fft_radix<I::R>(w);
    const int obase_idx = (j / I::NSP) * (I::NSP * I::R) + k;
#pragma unroll
    for (int q = 0; q < I::R; ++q) {
      const int oidx = obase_idx + q * I::NSP;
      if constexpr (TO_GLOBAL) {
        cf r = w[q];
        if constexpr (C::TWID == TWID_FOURSTEP_OUT) {
          // e^{-2 pi i (n2 * k1)/Ntot} = HI[m >> s] * LO[m & mask],  m = n2*k1 < Ntot
          const unsigned m = gi * (unsigned)oidx;
          const cf lo = lo_lds[m & a.fs_lo_mask];
          const cf hi = a.tw_hi[m >> a.fs_shift];
          r = cmul(r, cmul(hi, lo));
        }
        if constexpr (MUL) r = cmul(r, lo_lds[b * I::R + q]);   // this thread's slice of the spectrum, held in registers by the caller
        if (a.scale != 1.0f) r = r * a.scale;
        // last stage: Ns_prev = N/R, so oidx = j + q*(N/R): the q term is uniform
        cf* pq = po + (unsigned)(b * C::TPL + q * I::NSP) * es;
        const unsigned voff = (unsigned)line * ls + (unsigned)u * es;
        if constexpr (VIEW) {   // transposed store of pass B: element k = oidx * out_S + (row of the line): inside [v_out_lo, v_out_hi) only, 0 outside [v_zlo, v_zhi)
          static_assert(C::OUT_COL, "views ride the transposed store of pass B");
          const int kk = (int)((unsigned)oidx * es + (unsigned)((tile * C::T) % a.out_S) + (unsigned)line);
          if (kk < a.v_zlo || kk >= a.v_zhi) r = cf{0.0f, 0.0f};
          if (live && kk >= a.v_out_lo && kk < a.v_out_hi) pq[voff] = cswap_if<C::SWAP_OUT>(r);
        } else
        if (live) st_stream<NT>(pq + voff, cswap_if<C::SWAP_OUT>(r));
      } else {
        lds[lds_index<C>(line, oidx)] = w[q];
      }
    }
  }
}

template <class C>
__global__ void __launch_bounds__(C::THREADS) fft_lines_kernel(const LineArgs a) {
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_lds = lds + C::DATA_ELEMS;
  cf* lo_lds = tw_lds + C::TW_LDS_ELEMS;
  const int t = threadIdx.x;

  if constexpr (C::PREFETCH) {
    // (tables first, then the loads: issuing the first tile's loads ahead of the table staging makes every wave of the workgroup
    // wait at the staging barrier for the slowest wave's data — measured 292 vs 338 GPoints/s at N = 1024 with one-shot grids.
    // Skipping the LDS table on one-shot grids — each thread's stage-1 roots straight from the global table into registers,
    // requested behind the tile loads, no staging barrier — measured no better at 1024 (348-362 vs 347-354) and 3-8 % worse at
    // 64...512: profiles/r02_lines_regtw_ab.log)
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
    __syncthreads();
    cf v[C::E], vn[C::E];
    long long tile = blockIdx.x;
    if (tile < a.num_tiles) stage_read<C, 0, C::STREAM_NT>(v, a, tile, t, lds);
    for (; tile < a.num_tiles; tile += gridDim.x) {
      const long long next = tile + gridDim.x;
      if (next < a.num_tiles) stage_read<C, 0, C::STREAM_NT>(vn, a, next, t, lds);      // in flight while this tile is computed and stored
      stage_compute_write<C, 0>(v, a, tile, t, lds, tw_lds, lo_lds);
      lines_sync<C>();
      stage_read<C, 1>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 1, C::STREAM_NT>(v, a, tile, t, lds, tw_lds, lo_lds);
#pragma unroll
      for (int e = 0; e < C::E; ++e) v[e] = vn[e];
    }
    return;
  }
  // stage tables (and the four-step LO table) -> LDS once per workgroup
  if constexpr (C::TW_LDS_ELEMS > 0) {
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
  }
  if constexpr (C::LO_ELEMS > 0) {
    for (int i = t; i < C::LO_ELEMS; i += C::THREADS) lo_lds[i] = a.tw_lo[i & a.fs_lo_mask];
  }
  if constexpr (C::TW_LDS_ELEMS > 0 || C::LO_ELEMS > 0) __syncthreads();

  cf fsw[C::TWID == TWID_FOURSTEP_IN ? C::E : 1];
  // every tile of this workgroup sits at the same position inside its group <=> the tile stride is a
  // multiple of the group: the roots are then loop-invariant and computed once per launch
  const bool fs_hoist = C::TWID == TWID_FOURSTEP_IN && ((long long)gridDim.x * C::T) % a.fs_group == 0;
  if constexpr (C::TWID == TWID_FOURSTEP_IN) { if (fs_hoist) fourstep_in_roots<C>(fsw, a, blockIdx.x, t); }

  constexpr bool SNT = C::STREAM_NT;
  for (long long tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    cf v[C::E];
    stage_read<C, 0, SNT>(v, a, tile, t, lds);
    if constexpr (C::TWID == TWID_FOURSTEP_IN) {
      if (!fs_hoist) fourstep_in_roots<C>(fsw, a, tile, t);
#pragma unroll
      for (int e = 0; e < C::E; ++e) v[e] = cmul(v[e], fsw[e]);
    }
    stage_compute_write<C, 0, SNT>(v, a, tile, t, lds, tw_lds, lo_lds);     // (NT only matters where a stage stores to memory)
    if constexpr (C::NSTAGES >= 2) {
      lines_sync<C>();
      stage_read<C, 1>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 1, SNT>(v, a, tile, t, lds, tw_lds, lo_lds);
    }
    if constexpr (C::NSTAGES == 3) {
      lines_sync<C>();
      stage_read<C, 2>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 2, SNT>(v, a, tile, t, lds, tw_lds, lo_lds);
    }
  }
}

// ---- mapped sides (src/kernels/ioview.js:56-380, zero_pad.js:21-80, layout_semantics.js:178-232) ---------
// Line G of a launch -> physical base of its idx = 0 element; false when the line lies outside the side's box on some other dim
// (`zero`: inside the box but outside the value range on some other dim: a stored line is all zeros).
MI_DEV bool side_line(const SideMap& m, long long G, long long num_lines, long long& base, bool& zero) {
  zero = false;
  if (G >= num_lines) return false;
  long long rem = G, b = m.offset;
  bool ok = true;
  for (int d = 0; d < m.rank; ++d) {
    if (d == m.ax) continue;
    const long long q = rem / m.dims[d];
    const int c = (int)(rem - q * m.dims[d]);
    rem = q;
    ok = ok && c >= m.lo[d] && c < m.hi[d];
    zero = zero || c < m.zlo[d] || c >= m.zhi[d];
    b += (long long)c * m.stride[d];
  }
  base = b + rem * m.batch_stride;
  return ok;
}

// The line kernels with both sides mapped: first-stage loads read through imap (zeros outside its box), the stages run as
// usual with the finished lines KEPT in LDS, and the store pass writes through omap (crop / embed / zero ranges).  One launch
// does what gather / embed + zero-read + FFT + zero-write + extract / scatter did in up to five.  ROW and column
// configurations without four-step roots.
template <class C>
__global__ void __launch_bounds__(C::THREADS) fft_lines_mapped_kernel(const LineArgs a) {
  static_assert(C::IN_COL == C::OUT_COL && C::TWID == TWID_NONE, "ROW or column configuration");
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_lds = lds + C::DATA_ELEMS;
  const int t = threadIdx.x;
  if constexpr (C::TW_LDS_ELEMS > 0) {
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
    __syncthreads();
  }
  using I0 = StageInfo<C, 0>;
  const SideMap& im = a.imap;
  const SideMap& om = a.omap;
  for (long long tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    cf v[C::E];
    {
      int line, u; thread_map<C, 0>(t, line, u);
      long long base = 0; bool zero;
      const bool ok = side_line(im, tile * C::T + line, a.num_lines, base, zero);
      const long long sa = im.stride[im.ax];
      const int lo = im.lo[im.ax], hi = im.hi[im.ax];
#pragma unroll
      for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
        for (int q = 0; q < I0::R; ++q) {
          const int idx = u + b * C::TPL + q * (C::N / I0::R);
          cf x = {0.0f, 0.0f};
          if (ok && idx >= lo && idx < hi) x = a.in[base + (long long)idx * sa];
          v[b * I0::R + q] = cswap_if<C::SWAP_IN>(x);
        }
      }
    }
    const long long so = om.stride[om.ax];
    const int slo = om.lo[om.ax], shi = om.hi[om.ax], zlo = om.zlo[om.ax], zhi = om.zhi[om.ax];
    if constexpr (C::NSTAGES == 1) {
      // the whole line sits in one thread's registers (N <= 32): transform and store from there
      int line, u; thread_map<C, 0>(t, line, u);
      fft_radix<C::N>(v);
      long long base = 0; bool zero;
      if (side_line(om, tile * C::T + line, a.num_lines, base, zero)) {
#pragma unroll
        for (int q = 0; q < C::N; ++q) {
          if (q < slo || q >= shi) continue;
          cf r = cswap_if<C::SWAP_OUT>(v[q] * a.scale);
          if (a.fs_lo_mask & 1u) { r = cmul(r, a.tw_hi[q]); if (a.fs_lo_mask & 2u) r = r.yx; }   // Bluestein: chirp on the way out
          if (zero || q < zlo || q >= zhi) r = cf{0.0f, 0.0f};
          a.out[base + (long long)q * so] = r;
        }
      }
    } else {
      stage_compute_write<C, 0>(v, a, tile, t, lds, tw_lds, nullptr);
      __syncthreads();
      stage_read<C, 1>(v, a, tile, t, lds);
      __syncthreads();
      stage_compute_write<C, 1, false, true>(v, a, tile, t, lds, tw_lds, nullptr);
      if constexpr (C::NSTAGES == 3) {
        __syncthreads();
        stage_read<C, 2>(v, a, tile, t, lds);
        __syncthreads();
        stage_compute_write<C, 2, false, true>(v, a, tile, t, lds, tw_lds, nullptr);
      }
      __syncthreads();
      // store pass: a thread stays on one line (its box test and base are computed once), lanes run along the side that is
      // contiguous in LDS and, for dense targets, in memory
      int line, u;
      if constexpr (C::OUT_COL) { line = t % C::T; u = t / C::T; } else { line = t / C::TPL; u = t % C::TPL; }
      long long base = 0; bool zero;
      if (side_line(om, tile * C::T + line, a.num_lines, base, zero)) {
        for (int idx = u; idx < C::N; idx += C::TPL) {
          if (idx < slo || idx >= shi) continue;
          cf r = cswap_if<C::SWAP_OUT>(lds[lds_index<C>(line, idx)] * a.scale);
          if (a.fs_lo_mask & 1u) { r = cmul(r, a.tw_hi[idx]); if (a.fs_lo_mask & 2u) r = r.yx; }   // Bluestein: chirp on the way out
          if (zero || idx < zlo || idx >= zhi) r = cf{0.0f, 0.0f};
          a.out[base + (long long)idx * so] = r;
        }
      }
      __syncthreads();   // LDS is re-used by the next tile
    }
  }
}

// r2c of real lines of length N = 2H (H <= 16384 a power of two, H >= 64): the line is read as H complex numbers
// z[n] = x[2n] + i x[2n+1], transformed by the ROW stages with the finished lines KEPT in LDS, and the split of the
// half-length trick (kern_generic.hpp r2c_post_kernel: X[k] = E + wO, X[H-k] = conj(E - wO)) is applied from there: one launch,
// 4 B read + 4 B written per real point where the two-launch route moves 12.  a.out lines have H+1 bins (a.out_outer_stride);
// roots e^{-2 pi i k/N} = tw_hi[k >> fs_shift] * tw_lo[k & fs_lo_mask].
//
// TRIG (r02; dct_fft.js DCT-II / DST-II): the same kernel as a whole DCT-II (a.real_mode == 5) or DST-II (6) of real
// lines of length N = 2H — Makhoul's permutation v[n] = x[2n], v[N-1-n] = x[2n+1] (DST-II: odd samples negated) is applied while
// the line is staged into the LDS line buffer (pair loads from memory, float stores into LDS), the half-length FFT and the split
// give V = r2c(v), and each bin leaves as two real outputs y[k] = Re t, y[N-k] = -Im t, t = e^{-i pi k/2N} V[k] (DST-II: the
// output order reversed).  One launch and 8 B per point where the pre-pass + r2c + post-pass route moved 28.
//
// MAPPED (r02): the real side is read through a.imap (element = one float: strided layout, ioView.input box,
// zeroPad.read range — zeros outside) and the packed bins leave through a.omap (ioView.output / zeroPad.write / strided layout of
// the packed domain), as fft_lines_mapped_kernel does for c2c: no gather / embed / zero / extract / scatter launch around the r2c.
#ifndef MI355_R2C_POST_VEC
#define MI355_R2C_POST_VEC 1
#endif
#ifndef MI355_R2C_POST_ROOTS
#define MI355_R2C_POST_ROOTS 1
#endif
#ifndef MI355_R2C_POST_ROOTS_H
#define MI355_R2C_POST_ROOTS_H 2048
#endif
#ifndef MI355_R2C_POST_ROOTS_HMAX
#define MI355_R2C_POST_ROOTS_HMAX 4096
#endif
#ifndef MI355_R2C_POST_BATCH
#define MI355_R2C_POST_BATCH 1
#endif
template <class C, bool TRIG = false, bool MAPPED = false>
__global__ void __launch_bounds__(C::THREADS, C::THREADS == 256 && C::T == 1 && !MAPPED ? 2 : 1) fft_lines_r2c_kernel(const LineArgs a) {   // (N = 2^14: two workgroups per CU, see fft_lines_c2r_kernel)
  static_assert(!C::IN_COL && !C::OUT_COL && !C::SWAP_IN && !C::SWAP_OUT && C::TWID == TWID_NONE && C::NSTAGES >= 2, "forward ROW configuration with an LDS line buffer");
  static_assert(!(TRIG && MAPPED), "the fused DCT-II takes dense lines");
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_lds = lds + C::DATA_ELEMS;
  const int t = threadIdx.x;
  if constexpr (C::TW_LDS_ELEMS > 0) {
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
    __syncthreads();
  }
  constexpr int H = C::N, PER = H / 2 + 1, NREAL = 2 * H;
  const bool sine = TRIG && a.real_mode == 6;
  for (long long tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    cf v[C::E];
    if constexpr (TRIG) {
      const long long G0 = tile * C::T;
      const int live = (int)((a.num_lines - G0) < (long long)C::T ? (a.num_lines - G0) : (long long)C::T);
      float* ldsf = reinterpret_cast<float*>(lds);
      const float* xin = reinterpret_cast<const float*>(a.in);
      for (int p = t; p < live * H; p += C::THREADS) {
        const int l = p / H, n = p - l * H;
        const cf pr = *reinterpret_cast<const cf*>(xin + (G0 + l) * (long long)NREAL + 2 * n);   // (x[2n], x[2n+1])
        const int j1 = NREAL - 1 - n;
        ldsf[2 * lds_index<C>(l, n >> 1) + (n & 1)] = pr.x;                      // v[n]     = x[2n]
        ldsf[2 * lds_index<C>(l, j1 >> 1) + (j1 & 1)] = sine ? -pr.y : pr.y;     // v[N-1-n] = x[2n+1]
      }
      __syncthreads();
      {
        using I0 = StageInfo<C, 0>;
        int line, u; thread_map<C, 0>(t, line, u);
#pragma unroll
        for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
          for (int q = 0; q < I0::R; ++q) v[b * I0::R + q] = lds[lds_index<C>(line, u + b * C::TPL + q * (H / I0::R))];
        }
      }
      __syncthreads();   // everyone has its inputs before stage 0 re-uses the buffer
    } else if constexpr (MAPPED) {
      using I0 = StageInfo<C, 0>;
      const SideMap& im = a.imap;
      int line, u; thread_map<C, 0>(t, line, u);
      long long base = 0; bool zero;
      const bool ok = side_line(im, tile * C::T + line, a.num_lines, base, zero);
      const long long sa = im.stride[im.ax];
      const int lo = im.lo[im.ax], hi = im.hi[im.ax];
      const float* xin = reinterpret_cast<const float*>(a.in);
      const bool pairs = ok && sa == 1 && (base & 1) == 0;
#pragma unroll
      for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
        for (int q = 0; q < I0::R; ++q) {
          const int j = 2 * (u + b * C::TPL + q * (H / I0::R));     // z[n] = x[2n] + i x[2n+1]
          cf x = {0.0f, 0.0f};
          if (pairs && j >= lo && j + 1 < hi) x = *reinterpret_cast<const cf*>(xin + base + j);   // unit stride, even base: one 8-byte load
          else {
            if (ok && j >= lo && j < hi) x.x = xin[base + (long long)j * sa];
            if (ok && j + 1 >= lo && j + 1 < hi) x.y = xin[base + (long long)(j + 1) * sa];
          }
          v[b * I0::R + q] = x;
        }
      }
    } else {
      stage_read<C, 0>(v, a, tile, t, lds);
    }
    stage_compute_write<C, 0>(v, a, tile, t, lds, tw_lds, nullptr);
    __syncthreads();
    stage_read<C, 1>(v, a, tile, t, lds);
    __syncthreads();
    stage_compute_write<C, 1, false, true>(v, a, tile, t, lds, tw_lds, nullptr);
    if constexpr (C::NSTAGES == 3) {
      __syncthreads();
      stage_read<C, 2>(v, a, tile, t, lds);
      __syncthreads();
      stage_compute_write<C, 2, false, true>(v, a, tile, t, lds, tw_lds, nullptr);
    }
    __syncthreads();
    const long long G0 = tile * C::T;
    const int live = (int)((a.num_lines - G0) < (long long)C::T ? (a.num_lines - G0) : (long long)C::T);
    // MAPPED: a thread stays on one line (its box test and base are computed once); otherwise the pairs are dealt out flat
    [[maybe_unused]] long long obase = 0;
    [[maybe_unused]] bool ozero = false, oline = false;
    if constexpr (MAPPED) oline = side_line(a.omap, G0 + t / C::TPL, a.num_lines, obase, ozero);
    if constexpr (!TRIG && !MAPPED && MI355_R2C_POST_VEC && MI355_R2C_POST_ROOTS && H >= MI355_R2C_POST_ROOTS_H && H <= MI355_R2C_POST_ROOTS_HMAX && 1024 % (2 * C::TPL) == 0 && (H / 4) % C::TPL == 0) {
      // r03: the form below with the loop's table loads taken out of it (as in the c2r twin's raw-copy head): lane ru of a line takes the
      // items j = ru + TPL i, i.e. bins k = 2 ru + 1 + 2 TPL i and k + 1; with the 1024-entry LO table that is 1024 / (2 TPL) LO roots per lane and
      // bin, and the HI root is the same for the whole line (k + 1 crosses into the next HI block on the line's last lane only).
      // Same box (profiles/r03_r2c_post_roots.log): N = 2^12 585 -> 607, 2^13 557 -> 603 G real samples/s; the 8 / 16 trips of N = 2^14 / 2^15
      // unrolled this way lose (526 -> 430, 513 -> 353) and keep the loop below
      typedef float f4w __attribute__((ext_vector_type(4), aligned(8)));
      constexpr int QP = H / 4, S = 2 * C::TPL, WL = 1024 / S, NI = QP / C::TPL, NWH = (NI + WL - 1) / WL + 1;
      const int rl = C::T == 1 ? 0 : t / C::TPL, ru = C::T == 1 ? t : t % C::TPL;
      if (rl < live) {
        cf wl0[WL], wl1[WL], whs[NWH];
#pragma unroll
        for (int c = 0; c < WL; ++c) { wl0[c] = a.tw_lo[2 * ru + 1 + S * c]; wl1[c] = a.tw_lo[(2 * ru + 2 + S * c) & 1023]; }
#pragma unroll
        for (int c = 0; c < NWH; ++c) whs[c] = a.tw_hi[c];
        const bool last_lane = ru == C::TPL - 1;
        cf* x = a.out + (G0 + rl) * a.out_outer_stride;
        const auto split = [&](cf zk, cf zm0, cf w, cf& xk, cf& xm) {
          const cf zmc = {zm0.x, -zm0.y};
          const cf e = (zk + zmc) * 0.5f;
          const cf od = mul_neg_i((zk - zmc) * 0.5f);
          const cf wo = cmul(w, od);
          xk = (e + wo) * a.scale;
          xm = (e - wo) * a.scale;
          xm.y = -xm.y;
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int k = 2 * (ru + C::TPL * i) + 1;
          const cf h1 = (i % WL == WL - 1 && last_lane) ? whs[i / WL + 1] : whs[i / WL];
          cf xk0, xm0, xk1, xm1;
          split(lds[lds_index<C>(rl, k)], lds[lds_index<C>(rl, H - k)], cmul(whs[i / WL], wl0[i % WL]), xk0, xm0);
          split(lds[lds_index<C>(rl, k + 1)], lds[lds_index<C>(rl, H - k - 1)], cmul(h1, wl1[i % WL]), xk1, xm1);
          *reinterpret_cast<f4w*>(x + k) = f4w{xk0.x, xk0.y, xk1.x, xk1.y};
          if (k + 1 == H / 2) x[H - k] = xm0;                                   // X[H/2] is its own mirror: already stored
          else *reinterpret_cast<f4w*>(x + (H - k - 1)) = f4w{xm1.x, xm1.y, xm0.x, xm0.y};
        }
        if (ru == 0) {
          const cf z0 = lds[lds_index<C>(rl, 0)];
          x[0] = cf{(z0.x + z0.y) * a.scale, 0.0f};
          x[H] = cf{(z0.x - z0.y) * a.scale, 0.0f};
        }
      }
      __syncthreads();   // LDS is re-used by the next tile
      continue;
    }
    if constexpr (!TRIG && !MAPPED && MI355_R2C_POST_VEC) {
      // two adjacent bins per lane: X[k], X[k+1] leave as ONE 16-byte store and so do their mirrors X[H-k-1], X[H-k] (k odd, so that
      // the last item ends on the self-mirrored bin H/2); bin 0 / H is an item of its own.  Half the loop trips and store
      // instructions of the pair-per-lane form below (which the DCT and mapped variants keep).
      typedef float f4w __attribute__((ext_vector_type(4), aligned(8)));
      constexpr int QP = H / 4, PERV = QP + 1;
      const auto split = [&](cf zk, cf zm0, int k, cf& xk, cf& xm) {
        const cf w = cmul(a.tw_hi[(unsigned)k >> a.fs_shift], a.tw_lo[(unsigned)k & a.fs_lo_mask]);
        const cf zmc = {zm0.x, -zm0.y};
        const cf e = (zk + zmc) * 0.5f;
        const cf od = mul_neg_i((zk - zmc) * 0.5f);
        const cf wo = cmul(w, od);
        xk = (e + wo) * a.scale;
        xm = (e - wo) * a.scale;
        xm.y = -xm.y;
      };
      for (int p = t; p < live * PERV; p += C::THREADS) {
        const int line = p / PERV, j = p - line * PERV;
        cf* x = a.out + (G0 + line) * a.out_outer_stride;
        if (j == QP) {
          const cf z0 = lds[lds_index<C>(line, 0)];
          x[0] = cf{(z0.x + z0.y) * a.scale, 0.0f};
          x[H] = cf{(z0.x - z0.y) * a.scale, 0.0f};
          continue;
        }
        const int k = 2 * j + 1;
        cf xk0, xm0, xk1, xm1;
        split(lds[lds_index<C>(line, k)], lds[lds_index<C>(line, H - k)], k, xk0, xm0);
        split(lds[lds_index<C>(line, k + 1)], lds[lds_index<C>(line, H - k - 1)], k + 1, xk1, xm1);
        *reinterpret_cast<f4w*>(x + k) = f4w{xk0.x, xk0.y, xk1.x, xk1.y};
        if (k + 1 == H / 2) x[H - k] = xm0;                                   // X[H/2] is its own mirror: already stored
        else *reinterpret_cast<f4w*>(x + (H - k - 1)) = f4w{xm1.x, xm1.y, xm0.x, xm0.y};
      }
      __syncthreads();   // LDS is re-used by the next tile
      continue;
    }
    // pairs in batches of QU (the batch's root factors and LDS values requested before the first pair is finished).  Measured
    // (profiles/r02_split_load_batching.log): no gain for r2c at any size and a loss at N >= 2^14, so QU = 1 here; the c2r twin
    // below, whose batch covers global loads of the bins themselves, gains 20-40 % with 4
    constexpr int QU = MI355_R2C_POST_BATCH;
    const int pend = MAPPED ? PER : live * PER, pstep = MAPPED ? C::TPL : C::THREADS;
    for (int p0 = MAPPED ? t % C::TPL : t; p0 < pend; p0 += QU * pstep) {
     cf whs[QU], wls[QU], zks[QU], zms[QU];
#pragma unroll
     for (int j = 0; j < QU; ++j) {
      const int p = p0 + j * pstep;
      if (p >= pend) break;
      const int line = MAPPED ? t / C::TPL : p / PER, k = MAPPED ? p : p - line * PER;
      whs[j] = a.tw_hi[(unsigned)k >> a.fs_shift]; wls[j] = a.tw_lo[(unsigned)k & a.fs_lo_mask];
      zks[j] = lds[lds_index<C>(line, k)]; zms[j] = lds[lds_index<C>(line, k == 0 ? 0 : H - k)];
     }
#pragma unroll
     for (int j = 0; j < QU; ++j) {
      const int p = p0 + j * pstep;
      if (p >= pend) break;
      const int line = MAPPED ? t / C::TPL : p / PER, k = MAPPED ? p : p - line * PER;
      const int km = k == 0 ? 0 : H - k;
      const cf zk = zks[j], zm0 = zms[j];
      const cf w = cmul(whs[j], wls[j]);
      const cf zmc = {zm0.x, -zm0.y};
      const cf e = (zk + zmc) * 0.5f;
      const cf od = mul_neg_i((zk - zmc) * 0.5f);
      const cf wo = cmul(w, od);
      const cf xk = (e + wo) * a.scale;
      cf xm = (e - wo) * a.scale;
      xm.y = -xm.y;
      if constexpr (TRIG) {
        float* y = reinterpret_cast<float*>(a.out) + (G0 + line) * (long long)NREAL;
        const auto emit = [&](cf V, int m) {            // bin m of V = r2c(v) -> y[m] and y[N-m]
          // e^{-i pi m/2N}.  Evaluated (f32 sincospi of an exactly representable argument, good to ~1e-7) rather than read from the
          // f64-built table behind the LO roots that the DCT-III kernel uses: two more dependent loads per bin pair in this store
          // loop measured 205 vs 245 G points/s at N = 1024 (profiles/r02_trig_probes.log)
          float sn, cs;
#ifdef MI355_HOST_EMU   /* the shipped arithmetic's dominant error — the argument rounded to f32 — reproduced on the host (an exact evaluation of the f32 argument) */
          { const float arg = -(float)m / (float)(2 * NREAL); cs = (float)std::cos(3.14159265358979323846 * (double)arg); sn = (float)std::sin(3.14159265358979323846 * (double)arg); }
#else
          sincospif(-(float)m / (float)(2 * NREAL), &sn, &cs);
#endif
          const float re = V.x * cs - V.y * sn, im = -(V.x * sn + V.y * cs);
          y[sine ? NREAL - 1 - m : m] = re;
          if (m > 0 && 2 * m != NREAL) y[sine ? m - 1 : NREAL - m] = im;
        };
        emit(xk, k);
        if (k == 0) emit(xm, H);
        else if (km != k) emit(xm, km);
      } else if constexpr (MAPPED) {
        const SideMap& om = a.omap;
        const auto put = [&](int kk, cf val) {
          if (!oline || kk < om.lo[om.ax] || kk >= om.hi[om.ax]) return;
          if (ozero || kk < om.zlo[om.ax] || kk >= om.zhi[om.ax]) val = cf{0.0f, 0.0f};
          a.out[obase + (long long)kk * om.stride[om.ax]] = val;
        };
        put(k, xk);
        if (k == 0) put(H, xm);
        else if (km != k) put(km, xm);
      } else {
        cf* x = a.out + (G0 + line) * a.out_outer_stride;
        x[k] = xk;
        if (k == 0) x[H] = xm;
        else if (km != k) x[km] = xm;
      }
     }
    }
    __syncthreads();   // LDS is re-used by the next tile
  }
}

// fftconv, first half (src/kernels/fft_conv.js:3-66 pointwise product): forward FFT of complex lines whose
// outputs are multiplied by the kernel spectrum a.tw_lo[k] (its conjugate when a.fs_shift != 0: correlation) as they are stored —
// the separate pointwise pass (8 B read + 8 B written per point) disappears.  One spectrum for every line.
// MAPPED (r02): the data lines are read through a.imap (strided lanes, the zero-padded embed of the linear modes, zeroPad.read) —
// fft_lines_mapped_kernel's first-stage loads — so that no gather / embed pass runs ahead of the product.
template <class C, bool MAPPED = false>
__global__ void __launch_bounds__(C::THREADS) fft_lines_mul_kernel(const LineArgs a) {
  static_assert(!C::IN_COL && !C::OUT_COL && !C::SWAP_IN && !C::SWAP_OUT && C::TWID == TWID_NONE && C::NSTAGES >= 2, "forward ROW configuration with an LDS line buffer");
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_lds = lds + C::DATA_ELEMS;
  const int t = threadIdx.x;
  if constexpr (C::TW_LDS_ELEMS > 0) {
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
    __syncthreads();
  }
  // the product rides the last stage's register store (MUL): the line never goes back to LDS, and the spectrum values a thread
  // needs are the same for every tile (its output indices do not depend on the tile), so they are loaded ONCE per workgroup into
  // registers.  (Round 1 multiplied in a second sweep over the line kept in LDS; re-reading the spectrum per tile in the store
  // path costs the same 271 us against 190 us for the plain transform over the same bytes.)
  cf hk[C::E];
  {
    using IL = StageInfo<C, C::NSTAGES - 1>;
    int line, u; thread_map<C, C::NSTAGES - 1>(t, line, u);
#pragma unroll
    for (int b = 0; b < IL::NB; ++b) {
#pragma unroll
      for (int q = 0; q < IL::R; ++q) {
        cf h = a.tw_lo[u + b * C::TPL + q * IL::NSP];      // last stage: output index = j + q * Ns_prev, j = u + b * TPL
        if (a.fs_shift) h.y = -h.y;
        hk[b * IL::R + q] = h;
      }
    }
  }
  for (long long tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    cf v[C::E];
    if constexpr (MAPPED) {
      using I0 = StageInfo<C, 0>;
      const SideMap& im = a.imap;
      int line, u; thread_map<C, 0>(t, line, u);
      long long base = 0; bool zero;
      const bool ok = side_line(im, tile * C::T + line, a.num_lines, base, zero);
      const long long sa = im.stride[im.ax];
      const int lo = im.lo[im.ax], hi = im.hi[im.ax];
#pragma unroll
      for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
        for (int q = 0; q < I0::R; ++q) {
          const int idx = u + b * C::TPL + q * (C::N / I0::R);
          cf x = {0.0f, 0.0f};
          if (ok && idx >= lo && idx < hi) {
            x = a.in[base + (long long)idx * sa];
            // Bluestein (plan.cpp emit_bluestein): the chirp a.tw_hi[n] rides the load (bit 1: the swap that turns the route into an inverse)
            if (a.fs_lo_mask & 1u) { if (a.fs_lo_mask & 2u) x = x.yx; x = cmul(x, a.tw_hi[idx]); }
          }
          v[b * I0::R + q] = x;
        }
      }
    } else {
      stage_read<C, 0>(v, a, tile, t, lds);
    }
    stage_compute_write<C, 0>(v, a, tile, t, lds, tw_lds, nullptr);
    lines_sync<C>();
    stage_read<C, 1>(v, a, tile, t, lds);
    lines_sync<C>();
    if constexpr (C::NSTAGES == 2) {
      stage_compute_write<C, 1, false, false, true>(v, a, tile, t, lds, tw_lds, hk);
    } else {
      stage_compute_write<C, 1>(v, a, tile, t, lds, tw_lds, nullptr);
      lines_sync<C>();
      stage_read<C, 2>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 2, false, false, true>(v, a, tile, t, lds, tw_lds, hk);
    }
  }
}

// c2r of packed spectra (H+1 bins per line, N = 2H) into real lines: the pre-split of the half-length trick
// (kern_generic.hpp c2r_pre_kernel: Z[k] = E + iO, E = X[k] + conj X[H-k], O = (X[k] - conj X[H-k]) e^{+2 pi i k/N}; the
// imaginary parts of X[0] and X[H] are ignored, real_complex.js:147-155) is applied in LDS before the first stage (directly
// in the first-stage loads for single-stage lines), and the unnormalised inverse of length H then lands x[2n] + i x[2n+1], i.e.
// the real line, through the ordinary last-stage store.  C is the INVERSE ROW configuration (the swap trick of the c2c kernels).
//
// TRIG (r02; DCT-III / DST-III, the inverses of the kernel above; a.real_mode == 7 / 8): the bins V[k] = (X[k] - i X[N-k]) e^{+i pi k/2N}/2
// are formed from the REAL input line while the pre-split reads them (DST-III: the line reversed), and the finished real line is
// un-permuted on its way out of the LDS line buffer: y[2n] = v[n], y[2n+1] = v[N-1-n] (DST-III: odd samples negated).
//
// MAPPED (r02): the packed bins are read through a.imap (zeros outside its box) and the real line leaves the
// LDS line buffer through a.omap (element = one float), as in the r2c kernel above.
#ifndef MI355_C2R_RAW_COPY
#define MI355_C2R_RAW_COPY 1
#endif
#ifndef MI355_C2R_RAW_COPY_H
#define MI355_C2R_RAW_COPY_H 2048
#endif
#ifndef MI355_C2R_RAW_COPY_HMAX
#define MI355_C2R_RAW_COPY_HMAX 8192
#endif
#ifndef MI355_C2R_RAW_DIRECT_HMAX
#define MI355_C2R_RAW_DIRECT_HMAX 4096
#endif
#ifndef MI355_C2R_PV_8K
#define MI355_C2R_PV_8K 1
#endif
#ifndef MI355_C2R_PV_16K
#define MI355_C2R_PV_16K 2
#endif
#ifndef MI355_C2R_PRE_VEC
#define MI355_C2R_PRE_VEC 1
#endif
#ifndef MI355_C2R_PRE_BATCH
#define MI355_C2R_PRE_BATCH 4
#endif
// one-line 256-thread workgroups (N = 2^14, a 64 KB line each): capped at 256 registers a wave, two to a CU.  Uncapped, the raw-copy head
// takes 256 + 46..56 AGPRs and the code the compiler makes of that is either as fast (489) or much slower (289-300), depending on details
// of the source that should not matter (profiles/r03_c2r_raw_roots.log)
#ifndef MI355_LINES_C2R_MINW
#define MI355_LINES_C2R_MINW 2
#endif
template <class C, bool TRIG = false, bool MAPPED = false>
__global__ void __launch_bounds__(C::THREADS, C::THREADS == 256 && C::T == 1 ? MI355_LINES_C2R_MINW : 1) fft_lines_c2r_kernel(const LineArgs a) {
  static_assert(!C::IN_COL && !C::OUT_COL && C::SWAP_IN && C::SWAP_OUT && C::TWID == TWID_NONE, "inverse ROW configuration");
  static_assert(!(TRIG || MAPPED) || C::NSTAGES >= 2, "the fused DCT-III and the mapped sides need the LDS line buffer");
  static_assert(!(TRIG && MAPPED), "the fused DCT-III takes dense lines");
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_lds = lds + C::DATA_ELEMS;
  const int t = threadIdx.x;
  if constexpr (C::TW_LDS_ELEMS > 0) {
    for (int i = t; i < C::TW_LDS_ELEMS; i += C::THREADS) tw_lds[i] = a.tw[i];
    __syncthreads();
  }
  constexpr int H = C::N;
  [[maybe_unused]] constexpr int NREAL = 2 * H;
  [[maybe_unused]] const bool sine = TRIG && a.real_mode == 8;
  using I0 = StageInfo<C, 0>;
  // raw copy of the packed lines (2048 <= H <= 8192; see the head of the tile loop): TPL threads per line, lane ru of line rl takes bins ru + TPL i
  constexpr bool RAW = !TRIG && !MAPPED && MI355_C2R_RAW_COPY && C::NSTAGES >= 2 && H >= MI355_C2R_RAW_COPY_H && H <= MI355_C2R_RAW_COPY_HMAX;
  constexpr int RAWN = RAW ? H / C::TPL : 1, RAW_WL = RAW ? 1024 / C::TPL : 1, RAW_NK = (H / 2) / C::TPL, RAW_NWH = H / 2048 + 1;
  static_assert(!RAW || (1024 % C::TPL == 0 && (H / 2) % C::TPL == 0 && H >= 2048), "k = ru + TPL i walks the LO table in whole strides");
  [[maybe_unused]] const int rl = C::T == 1 ? 0 : t / C::TPL, ru = C::T == 1 ? t : t % C::TPL;
  for (long long tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    cf v[C::E];
    {
      int line, u; thread_map<C, 0>(t, line, u);
      const long long G0 = tile * C::T, live_lines = a.num_lines - G0;
      const int live = (int)(live_lines < (long long)C::T ? live_lines : (long long)C::T);
      const int lclamp = line < live ? line : live - 1;
      if constexpr (C::NSTAGES >= 2) {
        // through LDS: one lane per pair (k, H-k) reads the two bins (every bin read once, forward and mirrored runs), forms
        // Z[k] and Z[H-k] — the r2c kernel's post-pass in reverse, a few registers per lane — and drops them into the line slot;
        // the first stage then picks its inputs up
        static_assert(C::PITCH >= H + 1, "a packed line fits a line slot");
        constexpr int PER = H / 2 + 1;
        [[maybe_unused]] long long ibase = 0;
        [[maybe_unused]] bool izero = false, iline = false;
        if constexpr (MAPPED) iline = side_line(a.imap, G0 + t / C::TPL, a.num_lines, ibase, izero);
        if constexpr (RAW) {
          // r03 (first for N = 2^14, one line per 256-thread workgroup): the packed line is first copied into its LDS slot as it is — a plain strided
          // loop whose loads are all independent: 16-32 of them in flight per thread where the pair loop below has 2-4 — and the pre-split then
          // runs IN the slot (a pair (k, H-k) is read and rewritten by one lane; roots from the cache-resident tables).  Same box
          // (profiles/r03_c2r_raw_copy.log): N = 2^14 386 -> 422 G real samples/s; N = 2^15 (512 threads, one workgroup per CU) 372 -> 350 in that first form.
          // The roots of the in-slot loop leave with the copy's loads: k = t + THREADS i, so with the 1024-entry LO table (fs_shift = 10,
          // plan.cpp emit_lines_r2c) a lane needs 1024 / THREADS LO roots and the HI root is the same for the whole workgroup; the loop itself
          // is LDS and arithmetic only (with the table loads inside it every trip waited out a cache latency): N = 2^14 423 -> 488,
          // N = 2^13 515 -> 564, N = 2^12 (two lines per workgroup) 547 -> 574.
          // N = 2^15 (MI355_C2R_RAW_COPY_HMAX=16384; one 512-thread workgroup per CU, 20 registers spilled) 371 -> 362 / 378 -> 393 on two
          // boxes: not used; with the NEXT line's loads issued ahead into registers (64 more live through the stages) it spills 87 and
          // runs at 263 (profiles/r03_c2r_raw_roots.log).
          cf* xl = lds + rl * C::PITCH;
          const cf* x = a.in + (G0 + (rl < live ? rl : live - 1)) * a.in_outer_stride;    // (a ragged last tile: the spare lines repeat the last one)
          cf wl[RAW_WL], whs[RAW_NWH];       // (per tile: kept across the stages they would cost the stages their registers)
          const auto pre = [&](cf pk, cf m, int k, cf w) {
            if (k == 0) { pk.y = 0.0f; m.y = 0.0f; }
            const cf mc = {m.x, -m.y};
            const cf e = pk + mc;
            const cf o = cmul_conj(pk - mc, w);
            xl[k] = e + mul_pos_i(o);
            if (k != 0 && H - k != k) { const cf ec = {e.x, -e.y}, oc = {o.x, -o.y}; xl[H - k] = ec + mul_pos_i(oc); }
          };
          cf pks[RAW_NK], ms[RAW_NK], pmid;
          if constexpr (H <= MI355_C2R_RAW_DIRECT_HMAX) {
            // N <= 8192: a lane loads BOTH bins of its pairs (k ascending, H - k descending: each a contiguous run across the lanes), all of
            // them and the roots up front, pre-splits in registers and writes Z[k], Z[H-k] into the slot: no pass over the slot, no barrier.
            // Same box (profiles/r03_c2r_direct.log): 2^12 576 -> 592, 2^13 555 -> 588; 2^14 498 -> 490, which keeps the in-slot pass
#pragma unroll
            for (int i = 0; i < RAW_NK; ++i) { pks[i] = x[ru + i * C::TPL]; ms[i] = x[H - (ru + i * C::TPL)]; }
            pmid = x[H / 2];
#pragma unroll
            for (int c = 0; c < RAW_WL; ++c) wl[c] = a.tw_lo[ru + c * C::TPL];
#pragma unroll
            for (int c = 0; c < RAW_NWH; ++c) whs[c] = a.tw_hi[c];
          } else {
            cf raw[RAWN];
#pragma unroll
            for (int i = 0; i < RAWN; ++i) raw[i] = x[ru + i * C::TPL];
            const cf rawh = x[H];
#pragma unroll
            for (int c = 0; c < RAW_WL; ++c) wl[c] = a.tw_lo[ru + c * C::TPL];
#pragma unroll
            for (int c = 0; c < RAW_NWH; ++c) whs[c] = a.tw_hi[c];
#pragma unroll
            for (int i = 0; i < RAWN; ++i) xl[ru + i * C::TPL] = raw[i];
            if (ru == 0) xl[H] = rawh;
            __syncthreads();
            // all of a lane's pairs are read before the first is rewritten: interleaved, every read had to wait for the write before it
            // (the compiler cannot tell that k' and H - k' of a later trip differ from this trip's k and H - k): c2r 2^14 473 -> 499
            // (profiles/r03_c2r_2p15_bounds.log; the same log prices the whole in-slot pass at 13-15 % of the launch)
#pragma unroll
            for (int i = 0; i < RAW_NK; ++i) { pks[i] = xl[ru + i * C::TPL]; ms[i] = xl[H - (ru + i * C::TPL)]; }
            pmid = xl[H / 2];
          }
#pragma unroll
          for (int i = 0; i < RAW_NK; ++i) pre(pks[i], ms[i], ru + i * C::TPL, cmul(whs[i / RAW_WL], wl[i % RAW_WL]));
          if (ru == 0) pre(pmid, pmid, H / 2, cmul(whs[RAW_NWH - 1], wl[0]));      // the self-mirrored bin: (H/2) & 1023 = 0
        } else if constexpr (!TRIG && !MAPPED && MI355_C2R_PRE_VEC && H >= 2048) {
          // (N >= 4096: below that the pair-per-lane batches further down measured 5-10 % faster, profiles/r02_split_vec_ab.log)
          // two adjacent bins per lane: X[k], X[k+1] and their mirrors X[H-k-1], X[H-k] arrive as two 16-byte loads (k odd; the last
          // item ends on the self-mirrored bin H/2; bins 0 / H are an item of their own), PV items' loads in flight at a time
          typedef float f4w __attribute__((ext_vector_type(4), aligned(8)));
          constexpr int QP = H / 4, PERV = QP + 1, PV = H == 8192 ? MI355_C2R_PV_8K : (H == 16384 ? MI355_C2R_PV_16K : 2);   // (H = 8192: 403 vs 266 G real points/s with 1 vs 2 items in flight; H = 16384: 346 vs 383)
          const auto presplit = [&](cf pk, cf m, int k, cf wh, cf wl, cf* xl) {
            if (k == 0) { pk.y = 0.0f; m.y = 0.0f; }
            const cf w = cmul(wh, wl);
            const cf mc = {m.x, -m.y};
            const cf e = pk + mc;
            const cf o = cmul_conj(pk - mc, w);
            xl[k] = e + mul_pos_i(o);
            if (k != 0 && H - k != k) { const cf ec = {e.x, -e.y}, oc = {o.x, -o.y}; xl[H - k] = ec + mul_pos_i(oc); }
          };
          const int vend = live * PERV;
          for (int p0 = t; p0 < vend; p0 += PV * C::THREADS) {
            f4w up[PV], dn[PV];
            cf wh[PV][2], wl[PV][2];
#pragma unroll
            for (int i = 0; i < PV; ++i) {
              const int p = p0 + i * C::THREADS;
              if (p >= vend) break;
              const int l = p / PERV, j = p - l * PERV;
              const cf* x = a.in + (G0 + l) * a.in_outer_stride;
              if (j == QP) {
                const cf x0 = x[0], xh = x[H];
                up[i] = f4w{x0.x, x0.y, xh.x, xh.y};
                wh[i][0] = a.tw_hi[0]; wl[i][0] = a.tw_lo[0];
              } else {
                const int k = 2 * j + 1;
                up[i] = *reinterpret_cast<const f4w*>(x + k);
                dn[i] = *reinterpret_cast<const f4w*>(x + (H - k - 1));
#pragma unroll
                for (int c = 0; c < 2; ++c) { wh[i][c] = a.tw_hi[(unsigned)(k + c) >> a.fs_shift]; wl[i][c] = a.tw_lo[(unsigned)(k + c) & a.fs_lo_mask]; }
              }
            }
#pragma unroll
            for (int i = 0; i < PV; ++i) {
              const int p = p0 + i * C::THREADS;
              if (p >= vend) break;
              const int l = p / PERV, j = p - l * PERV;
              cf* xl = lds + l * C::PITCH;
              if (j == QP) presplit(cf{up[i].x, up[i].y}, cf{up[i].z, up[i].w}, 0, wh[i][0], wl[i][0], xl);
              else {
                const int k = 2 * j + 1;
                presplit(cf{up[i].x, up[i].y}, cf{dn[i].z, dn[i].w}, k, wh[i][0], wl[i][0], xl);          // X[k], X[H-k]
                presplit(cf{up[i].z, up[i].w}, cf{dn[i].x, dn[i].y}, k + 1, wh[i][1], wl[i][1], xl);      // X[k+1], X[H-k-1]
              }
            }
          }
        } else {
        // the pairs are taken in batches of PU: all of a batch's loads (two bins and two root factors per pair) are issued before
        // the first is used — one pair per iteration left the loop waiting out a full memory latency per pair
        constexpr int PU = H >= 8192 ? 1 : MI355_C2R_PRE_BATCH;   // one-line workgroups of 2^14 points and more lose with batches (310 -> 255)
        const int pend = MAPPED ? PER : live * PER, pstep = MAPPED ? C::TPL : C::THREADS;
        for (int p0 = MAPPED ? t % C::TPL : t; p0 < pend; p0 += PU * pstep) {
          cf pks[PU], ms[PU], whs[PU], wls[PU];
#pragma unroll
          for (int j = 0; j < PU; ++j) {
            const int p = p0 + j * pstep;
            if (p >= pend) break;
            const int l = MAPPED ? t / C::TPL : p / PER, k = MAPPED ? p : p - l * PER;
            cf pk, m;
            if constexpr (MAPPED) {
              const SideMap& im = a.imap;
              const auto bin = [&](int kk) {
                cf r = {0.0f, 0.0f};
                if (iline && kk >= im.lo[im.ax] && kk < im.hi[im.ax]) r = a.in[ibase + (long long)kk * im.stride[im.ax]];
                return r;
              };
              pk = bin(k); m = bin(H - k);
            } else if constexpr (TRIG) {
              const float* X = reinterpret_cast<const float*>(a.in) + (G0 + l) * (long long)NREAL;
              const auto bin = [&](int mm) {                 // V[mm] from the real line (trig_real_pre_kernel kinds 10 / 11)
                const float re = sine ? X[NREAL - 1 - mm] : X[mm];
                const float im = mm == 0 ? 0.0f : (sine ? X[mm - 1] : X[NREAL - mm]);
                const cf ph = a.tw_lo[1024 + mm];           // e^{-i pi m/2N} (table behind the LO roots); the phase wanted is its conjugate
                const float cs = ph.x, sn = -ph.y;
                cf r; r.x = 0.5f * (re * cs + im * sn); r.y = 0.5f * (re * sn - im * cs);
                return r;
              };
              pk = bin(k); m = bin(H - k);
            } else {
              const cf* x = a.in + (G0 + l) * a.in_outer_stride;
              pk = x[k]; m = x[H - k];
            }
            pks[j] = pk; ms[j] = m;
            whs[j] = a.tw_hi[(unsigned)k >> a.fs_shift]; wls[j] = a.tw_lo[(unsigned)k & a.fs_lo_mask];
          }
#pragma unroll
          for (int j = 0; j < PU; ++j) {
            const int p = p0 + j * pstep;
            if (p >= pend) break;
            const int l = MAPPED ? t / C::TPL : p / PER, k = MAPPED ? p : p - l * PER;
            cf* xl = lds + l * C::PITCH;
            cf pk = pks[j], m = ms[j];
            if (k == 0) { pk.y = 0.0f; m.y = 0.0f; }
            const cf w = cmul(whs[j], wls[j]);
            const cf mc = {m.x, -m.y};
            const cf e = pk + mc;
            const cf o = cmul_conj(pk - mc, w);
            xl[k] = e + mul_pos_i(o);
            if (k != 0 && H - k != k) { const cf ec = {e.x, -e.y}, oc = {o.x, -o.y}; xl[H - k] = ec + mul_pos_i(oc); }
          }
        }
        }
        __syncthreads();
        const cf* zl = lds + lclamp * C::PITCH;
#pragma unroll
        for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
          for (int q = 0; q < I0::R; ++q) v[b * I0::R + q] = cswap_if<true>(zl[u + b * C::TPL + q * (H / I0::R)]);
        }
        __syncthreads();   // everyone has its inputs before stage 0 re-uses the buffer
      } else {
        const cf* x = a.in + (G0 + lclamp) * a.in_outer_stride;
#pragma unroll
        for (int b = 0; b < I0::NB; ++b) {
#pragma unroll
          for (int q = 0; q < I0::R; ++q) {
            const int k = u + b * C::TPL + q * (H / I0::R);
            cf p = x[k], m = x[H - k];
            if (k == 0) { p.y = 0.0f; m.y = 0.0f; }
            const cf w = cmul(a.tw_hi[(unsigned)k >> a.fs_shift], a.tw_lo[(unsigned)k & a.fs_lo_mask]);
            const cf mc = {m.x, -m.y};
            const cf e = p + mc;
            const cf o = cmul_conj(p - mc, w);
            v[b * I0::R + q] = cswap_if<true>(e + mul_pos_i(o));
          }
        }
      }
    }
    stage_compute_write<C, 0>(v, a, tile, t, lds, tw_lds, nullptr);
    if constexpr (C::NSTAGES >= 2) {
      lines_sync<C>();
      stage_read<C, 1>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 1, false, (TRIG || MAPPED) && C::NSTAGES == 2>(v, a, tile, t, lds, tw_lds, nullptr);
    }
    if constexpr (C::NSTAGES == 3) {
      lines_sync<C>();
      stage_read<C, 2>(v, a, tile, t, lds);
      lines_sync<C>();
      stage_compute_write<C, 2, false, TRIG || MAPPED>(v, a, tile, t, lds, tw_lds, nullptr);
    }
    if constexpr (MAPPED) {
      // the finished line sits in LDS as swapped pairs (x[2n+1], x[2n]); a thread stays on one line and walks it through omap
      __syncthreads();
      const SideMap& om = a.omap;
      const int line = t / C::TPL, u = t % C::TPL;
      long long base = 0; bool zero;
      if (side_line(om, tile * C::T + line, a.num_lines, base, zero)) {
        float* y = reinterpret_cast<float*>(a.out);
        const long long so = om.stride[om.ax];
        const int slo = om.lo[om.ax], shi = om.hi[om.ax], zlo = om.zlo[om.ax], zhi = om.zhi[om.ax];
        const bool pairs = so == 1 && (base & 1) == 0;
        for (int idx = u; idx < H; idx += C::TPL) {
          const cf r = lds[lds_index<C>(line, idx)] * a.scale;
          const int j = 2 * idx;
          const float y0 = (zero || j < zlo || j >= zhi) ? 0.0f : r.y, y1 = (zero || j + 1 < zlo || j + 1 >= zhi) ? 0.0f : r.x;
          if (pairs && j >= slo && j + 1 < shi) *reinterpret_cast<cf*>(y + base + j) = cf{y0, y1};   // unit stride, even base: one 8-byte store
          else {
            if (j >= slo && j < shi) y[base + (long long)j * so] = y0;
            if (j + 1 >= slo && j + 1 < shi) y[base + (long long)(j + 1) * so] = y1;
          }
        }
      }
      __syncthreads();   // LDS is re-used by the next tile
    }
    if constexpr (TRIG) {
      // the finished line sits in LDS as swapped pairs (v[2n+1], v[2n]) (the inverse runs on re/im-swapped data and the swap back
      // belongs to the store): un-permute while storing — one float2 (y[2n], y[2n+1]) per lane
      __syncthreads();
      const long long G0 = tile * C::T;
      const int live = (int)((a.num_lines - G0) < (long long)C::T ? (a.num_lines - G0) : (long long)C::T);
      const float* ldsf = reinterpret_cast<const float*>(lds);
      float* yout = reinterpret_cast<float*>(a.out);
      for (int p = t; p < live * H; p += C::THREADS) {
        const int l = p / H, n = p - l * H, j1 = NREAL - 1 - n;
        const float v0 = ldsf[2 * lds_index<C>(l, n >> 1) + (1 - (n & 1))];
        const float v1 = ldsf[2 * lds_index<C>(l, j1 >> 1) + (1 - (j1 & 1))];
        cf o; o.x = v0 * a.scale; o.y = (sine ? -v1 : v1) * a.scale;
        *reinterpret_cast<cf*>(yout + (G0 + l) * (long long)NREAL + 2 * n) = o;
      }
      __syncthreads();   // LDS is re-used by the next tile
    }
  }
}

}  // namespace mi355
