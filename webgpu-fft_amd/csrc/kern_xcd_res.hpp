// kern_xcd_res.hpp — "XCD-resident" four-step for N = 1024 x 1024 (the headline size): the transform never goes back to HBM
// between its two passes.
//
// Why (profiles/r02_l2_writeback_probe.log): an XCD's 4 MiB L2 is write-back INSIDE a kernel — a buffer of <= 1-2 MiB per XCD
// that is re-written in place never reaches the fabric, and same-XCD readers hit it.  The fused kernel of kern_xcd.hpp parks
// the whole 8 MiB intermediate of a transform in a workspace slot, so it crosses the fabric twice (32 B per point measured,
// twice the algorithmic 16).  Here the 32 workgroups of one XCD (one per CU, 512 threads) hold the transform in their
// registers and LDS and hand it to each other through a small L2-resident exchange buffer that is re-used for every hand-off:
//
//   x (HBM) --loads--> pass A (column FFTs, registers + LDS) --> exchange through L2 --> pass B stage 0 (registers)
//           --> LDS transpose --> pass B stage 1 --> out (HBM, 256-byte segments)
//
// Replaces, like kern_xcd.hpp: the two-step axis route of src/plan.js:456-595 (FFT(n1), twiddle :114-153, per-element
// transposes :375-384, FFT(n2)).
//
// Index algebra.  n = n1*1024 + n2, k = k1 + 1024*k2.
//   pass A    Y[k1][n2]  = sum_n1 x[n1][n2] W1024^(n1 k1),   Y' = Y * W_N^(k1 n2)                  (four-step roots)
//   pass B    n2 = j + 32 q,  k2 = p + 32 s:
//     stage 0   Z[k1][j][p] = (sum_q Y'[k1][j + 32 q] W32^(q p)) * W1024^(j p)
//     stage 1   X[k1 + 1024 (p + 32 s)] = sum_j Z[k1][j][p] W32^(j s)
// Workgroup r of the group owns columns n2 in [32 r, 32 r + 32) in pass A — as two tiles h = 0, 1 of 16 columns — and rows
// k1 in [32 r, 32 r + 32) in pass B.  A stage-0 butterfly (k1, j) of pass B takes ONE value from every workgroup q (column
// j of its tile h = j / 16), so it can run as soon as tile set h has been exchanged, before the other tile has even been
// loaded: per tile set only half a transform (4 MiB per XCD) is in flight, and it moves in four channels of 4 columns (1 MiB
// each, one per pair of waves) that may be staggered in time (`depth` channels in flight, f.split) so that the exchange
// buffer really stays in the L2.
//
// Per transform and workgroup (registers per thread in brackets, data only):
//   1 tile 0: x -> registers (nontemporal loads), radix-32, LDS, radix-32, four-step roots             [64]
//   2 set 0 : push (4 channels) -> L2 -> butterfly (k1 = 32 r + row, j) in, radix-32, roots -> Z0       [64 + 64]
//   3 tile 1: as 1 (Z0 stays in registers: the LDS is busy)                                            [64 + 64]
//   4 park the p < 16 half of Z0 in its slots of the LDS transpose                                     [32 + 64]
//   5 set 1 : as 2 -> Z1                                                                               [32 + 64 + 64]
//   6 half A: Z1 (p < 16) -> LDS, barrier, rows of j out, radix-32, 256-byte stores                    [32 + 32 + 64]
//   7 half B: the p >= 16 halves of Z0, Z1 -> LDS, ...                                                 [64]
//
// Hand-off protocol (same XCD by construction: the group is formed from the XCC_ID register, kern_xcd.hpp xcd_register):
// producer waves store plainly (the lines stay in the shared L2), `s_waitcnt vmcnt(0)`, the two waves of a channel meet on an
// LDS counter and the later one bumps the channel's PUSH counter; consumer waves poll it with device-scope (sc1) loads and
// read the payload with sc1 loads, which bypass the CU's L1 — no L2 write-back, no L1 invalidate.  After its reads a wave
// pair bumps the READ counter that gates the next push into the same buffer.  Counters are monotonic (no reset inside a
// launch); every spin is bounded and a timeout raises the sticky error word (api.hip reports it) and stops the workgroup.
#pragma once
#include "kern_xcd.hpp"

namespace mi355 {

#ifndef MI355_HOST_EMU
#define MI_UNIFORM_U32(v) ((unsigned)__builtin_amdgcn_readfirstlane((int)(v)))   /* a value every lane agrees on -> SGPR */
#define MI_WAVE_ONLY_SYNC() MI_WAVE_SYNC()
// Pins a wave-uniform global address in an SGPR pair and hides where it came from, so that `base[lane offset]` becomes the
// "SGPR base + 32-bit VGPR offset" form of global_load / global_store.  Left to itself the compiler re-associates
// (uniform + constant) + lane offset into (pointer + lane offset) + constant: a 64-bit VGPR address per element, hoisted out of
// the transform loop — 230 spilt registers in this kernel.  The address goes through the asm as an integer and comes back as an
// explicit global (address space 1) pointer: a plain pointer laundered this way loses its address space and turns flat_*.
// a per-lane value the optimiser must treat as freshly defined here: loads indexed by it are not loop-invariant (the 31 + 31 stage
// roots of a thread are: hoisted out of the transform loop they would pin 124 registers for the whole kernel)
#define MI_OPAQUE_LANE_INT(i) asm volatile("" : "+v"(i))
#define MI_COMPILER_ACQUIRE() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); asm volatile("" ::: "memory"); } while (0)
template <class T> MI_DEV T* sgpr_base(T* p) {
  unsigned long long a = (unsigned long long)p;
  asm volatile("" : "+s"(a));
  return (T*)(__attribute__((address_space(1))) T*)a;
}
#define MI_LDS_ATOMIC_ADD_U32(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// device-scope (sc1) 8-byte load: served by the XCD's L2, never by this CU's L1
MI_DEV cf ld_sc1(const cf* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  cf r;
  r.x = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffull));
  r.y = __builtin_bit_cast(float, (unsigned)(v >> 32));
  return r;
}
#else
#define MI_UNIFORM_U32(v) ((unsigned)(v))
#define MI_OPAQUE_LANE_INT(i) do { } while (0)
#define MI_COMPILER_ACQUIRE() __atomic_thread_fence(__ATOMIC_ACQUIRE)
template <class T> MI_DEV T* sgpr_base(T* p) { return p; }
#define MI_WAVE_ONLY_SYNC() emu::sync_wave()
#define MI_LDS_ATOMIC_ADD_U32(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
MI_DEV cf ld_sc1(const cf* p) { __atomic_thread_fence(__ATOMIC_SEQ_CST); return *p; }
#endif

struct XcdResCfg {
  static constexpr int THREADS = 512;
  static constexpr int LPAD = 17;                               // pass A LDS image [1024 idx][17]: both thread maps conflict-free
  static constexpr int DATA_ELEMS = 1024 * LPAD;                // 136 KB; the pass B transpose [16 p][32 j][32 rows] (128 KB) re-uses it
  static constexpr int TW_ELEMS = 31 * 32;                      // W1024^(q k), rows q = 1..31
  static constexpr int WORDS = 48;                              // registration words, per-wave flags, channel meeting counters, [32..47] phase stamps (diagnostic build)
  static constexpr int LDS_BYTES = (DATA_ELEMS + TW_ELEMS) * 8 + WORDS * 4;
  static constexpr long long CHANNEL_ELEMS = 32ll * 32 * 4 * 32;   // one channel: [consumer 32][producer 32][4 columns][32 rows] = 1 MiB
  static constexpr long long XCD_W_ELEMS = 4 * CHANNEL_ELEMS;      // exchange buffer per XCD
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// e^{-2 pi i m/N} = HI[m >> shift] * LO[m & mask]
MI_DEV cf res_root(const XcdFusedArgs& f, unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); }

// A wave waits until *counter >= target: lane 0 polls (bounded), everybody reads the verdict from the wave's LDS flag.  An abort
// raised by any wave of the workgroup (s_abort) ends every later wait at once, so a lost co-resident never costs more than one
// spin budget.
MI_DEV bool res_wait(unsigned* counter, unsigned target, const XcdFusedArgs& f, unsigned* s_flag, unsigned* s_abort, int lane) {
  if (lane == 0) {
    unsigned ok = 0;
    for (unsigned it = 0; it < f.spin_limit; ++it) {
      if (MI_ATOMIC_LOAD_U32(counter) >= target) { ok = 1; break; }
      if (*(volatile unsigned*)s_abort) break;
      MI_SLEEP();
    }
    if (!ok) { MI_ATOMIC_OR_U32(f.sticky_error, 2u); *(volatile unsigned*)s_abort = 1u; }
    *(volatile unsigned*)s_flag = ok;
  }
  MI_WAVE_ONLY_SYNC();
  const bool ok = *(volatile unsigned*)s_flag != 0;
  MI_WAVE_ONLY_SYNC();   // the flag is re-used by this wave's next wait
  // Invariant: the payload behind `counter` is read only AFTER the poll has seen the producers' bumps.  The payload loads are
  // device-scope (sc1) loads served by the shared L2, so no cache maintenance is needed, but nothing above orders them for the
  // COMPILER (relaxed atomics, volatile LDS flag): this fence is the compiler-visible ordering point (workgroup scope: no L2
  // write-back, no L1 invalidate beyond what the sc1 loads do anyway).
  MI_COMPILER_ACQUIRE();
  return ok;
}
// this wave's vector-memory operations are complete; the later of the two waves of a channel bumps the XCD-wide counter
MI_DEV void res_signal(unsigned* counter, unsigned* s_meet, int lane) {
  MI_WAIT_VMEM();
  MI_WAVE_ONLY_SYNC();   // every lane of the wave is here (lock-step on the GPU; the emulation's lanes are free-running threads)
  if (lane == 0) {
    const unsigned old = MI_LDS_ATOMIC_ADD_U32(s_meet, 1u);
    if (old & 1u) MI_ATOMIC_ADD_U32(counter, 1u);
  }
}

// four-step roots W_N^(n2 (u2 + 32 q)) on a finished column: exact lookups every 8th element, recurrence with W_N^(32 n2) between
MI_DEV void res_fourstep(cf (&w)[32], const XcdFusedArgs& f, unsigned n2, unsigned u2) {
  const cf step = res_root(f, n2 * 32u);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cf a = res_root(f, n2 * (u2 + 256u * (unsigned)g));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      w[8 * g + i] = cmul(w[8 * g + i], a);
      if (i < 7) a = cmul(a, step);
    }
  }
}

// STAMP (diagnostic build only, never the product variant): thread 0 of every workgroup accumulates the wall time (s_memrealtime,
// 100 MHz) between phase boundaries into ctl->bar[256 + blockIdx][phase]; tools/res_stamps.py prints the breakdown.  The stamps go
// to memory nothing else reads.
#ifndef MI355_HOST_EMU
#define MI_REALTIME() __builtin_amdgcn_s_memrealtime()
#else
#define MI_REALTIME() 0ull
#endif
#define RES_STAMP(i) do { if constexpr (STAMP) { if (t == 0) { const unsigned long long now_ = MI_REALTIME(); s_words[32 + (i)] += (unsigned)(now_ - stamp_last); stamp_last = now_; } } } while (0)

// INV: inverse transform by the swap trick (re/im exchanged at the x loads and at the output stores).
// MATH = false: the data movement and synchronisation skeleton alone (butterflies and roots skipped) — tools/microbench/xcd3.hip
// times it to show what the FFT arithmetic costs on top.
template <bool INV, bool MATH = true, bool STAMP = false>
__global__ void __launch_bounds__(XcdResCfg::THREADS) fft_xcd_res_kernel(const XcdFusedArgs f) {
  using K = XcdResCfg;
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw = lds + K::DATA_ELEMS;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw + K::TW_ELEMS);   // [0..6] registration, [8..15] per-wave flags, [16..19] PUSH meeting, [20..23] READ meeting, [24] abort
  const int t = threadIdx.x, lane = t & 63, wave = (int)MI_UNIFORM_U32(t >> 6);
  for (int i = t; i < K::TW_ELEMS; i += K::THREADS) tw[i] = f.tw_a[i];
  if (t < 32) s_words[16 + t] = 0u;
  unsigned long long stamp_last = 0;
  if (!xcd_register(f.ctl, 1u, f.spin_limit, f.sticky_error, s_words)) return;
  // wave-uniform by construction: keeping them in SGPRs lets every global access below take the "uniform base + 32-bit lane
  // offset" form (one VGPR of address per access pattern instead of a 64-bit pointer per element)
  const unsigned gslot = MI_UNIFORM_U32(s_words[0]), r = MI_UNIFORM_U32(s_words[1]), gsize = MI_UNIFORM_U32(s_words[2]),
                 gidx = MI_UNIFORM_U32(s_words[4]), groups = MI_UNIFORM_U32(s_words[5]);
  if (gsize != 32u) {   // the data distribution below IS 32 workgroups per XCD: anything else is a planner error, reported
    if (t == 0) MI_ATOMIC_OR_U32(f.sticky_error, 4u);
    return;
  }
  unsigned* const s_flag = &s_words[8 + wave];
  unsigned* const s_abort = &s_words[24];

  // thread maps
  const int a_cc = t & 15, a_u = t >> 4;            // tile loads + pass A stage 0: lanes across the 16 columns
  const int b_c2 = t >> 5, b_u2 = t & 31;            // pass A stage 1 + push: a half-wave owns one column
  const int ch = wave >> 1, jj = b_c2 & 3;           // exchange channel (a pair of waves), column inside the channel
  const int row = t & 31;                            // consumer side: lanes across the 32 rows of this workgroup
  const unsigned depth = f.split ? f.split : 4u;     // channels in flight (1, 2 or 4); channel c uses buffer c % depth
  const unsigned buf = (unsigned)ch % depth, uses_per_set = 4u / depth;
  const unsigned long long w_off = (unsigned long long)gslot * (unsigned long long)K::XCD_W_ELEMS + (unsigned long long)buf * (unsigned long long)K::CHANNEL_ELEMS;
  unsigned* const c_push = &f.ctl->bar[gslot * 8u + buf][0];
  unsigned* const c_read = &f.ctl->bar[gslot * 8u + 4u + buf][0];
  unsigned* const m_push = &s_words[16 + ch];
  unsigned* const m_read = &s_words[20 + ch];

  unsigned k = 0;
  if constexpr (STAMP) stamp_last = MI_REALTIME();
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf zlo[16], zhi[16];                             // Z0: p < 16 half until it is parked, p >= 16 half until half B
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      cf v[32];
      // ---- tile h: 16 columns of x into registers, column FFT of length 1024 ----
      {
        const cf* const xt = f.in + ((unsigned long long)tr * (unsigned long long)f.in_pitch + (32u * r + 16u * (unsigned)h));   // uniform
        const unsigned vo = (unsigned)a_cc + ((unsigned)a_u << 10);
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = cswap_if<INV>(ld_stream<true>(sgpr_base(xt + ((unsigned)q << 15)) + vo));
        if constexpr (MATH) fft_radix<32>(v);
        RES_STAMP(0);                                // x loads issued, data arrived (first use), first radix-32
        if (h == 1 || k > 0) __syncthreads();        // the LDS image is free: everyone is past its last reads of it
#pragma unroll
        for (int p = 0; p < 32; ++p) lds[(32 * a_u + p) * K::LPAD + a_cc] = v[p];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = lds[(b_u2 + 32 * q) * K::LPAD + b_c2];
        if constexpr (MATH) {
          // eight roots at a time: all 31 table reads in flight at once would pin 62 registers next to Z0
          int ti = b_u2;
          MI_OPAQUE_LANE_INT(ti);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int q = 8 * g; q < 8 * g + 8; ++q) { if (q > 0) v[q] = cmul(v[q], tw[(q - 1) * 32 + ti]); }
            MI_SCHED_FENCE();
          }
          fft_radix<32>(v);
          res_fourstep(v, f, (unsigned)(32 * (int)r + 16 * h + b_c2), (unsigned)b_u2);
        }
        RES_STAMP(1);                                // LDS exchange of the column FFT, second radix-32, four-step roots
      }
      // ---- exchange of tile set h: v[q] = Y'[k1 = b_u2 + 32 q][this column] goes to workgroup q ----
      const unsigned use = (2u * k + (unsigned)h) * uses_per_set + (unsigned)ch / depth;   // how often this buffer has been used before
      if (use > 0 && !res_wait(c_read, 32u * use, f, s_flag, s_abort, lane)) return;        // the previous payload has been read by everyone
      RES_STAMP(2);                                  // wait: channel buffer free
      {
        cf* const pw = f.wslots + (w_off + r * 128u);                      // uniform
        const unsigned vo = (unsigned)(jj * 32 + b_u2);
#pragma unroll
        for (int q = 0; q < 32; ++q) *(sgpr_base(pw + (unsigned)q * 4096u) + vo) = v[q];             // [consumer q][producer r][jj][row]
      }
      res_signal(c_push, m_push, lane);
      RES_STAMP(3);                                  // push stores issued and complete (vmcnt(0): also drains older output stores)
      if (!res_wait(c_push, 32u * (use + 1u), f, s_flag, s_abort, lane)) return;
      RES_STAMP(4);                                  // wait: everyone has pushed
      {
        const cf* const pr = f.wslots + (w_off + r * 4096u);               // uniform
        const unsigned vo = (unsigned)(jj * 32 + row);
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = ld_sc1(sgpr_base(pr + (unsigned)q * 128u) + vo);          // [consumer r][producer q][jj][row]
      }
      res_signal(c_read, m_read, lane);
      RES_STAMP(5);                                  // payload read from the L2
      // ---- pass B stage 0 on the butterfly (k1 = 32 r + row, j = 16 h + b_c2) ----
      if constexpr (MATH) {
        fft_radix<32>(v);
        int j = 16 * h + b_c2;
        MI_OPAQUE_LANE_INT(j);
        if (j > 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int p = 8 * g; p < 8 * g + 8; ++p) { if (p > 0) v[p] = cmul(v[p], tw[(j - 1) * 32 + p]); }
            MI_SCHED_FENCE();
          }
        }
      }
      RES_STAMP(6);                                  // pass B stage 0
      if (h == 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p) { zlo[p] = v[p]; zhi[p] = v[16 + p]; }
      } else {
        // ---- transposes through LDS, pass B stage 1, output.  E(p', j, row) = (p' * 32 + j) * 32 + row ----
        __syncthreads();                             // pass A's image of tile 1 has been read by everyone
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          lds[(p * 32 + b_c2) * 32 + row] = zlo[p];
          lds[(p * 32 + 16 + b_c2) * 32 + row] = v[p];
        }
        __syncthreads();
        cf y[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = lds[(b_c2 * 32 + j) * 32 + row];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          lds[(p * 32 + b_c2) * 32 + row] = zhi[p];
          lds[(p * 32 + 16 + b_c2) * 32 + row] = v[16 + p];
        }
        cf* const ot = f.out + ((unsigned long long)tr * (unsigned long long)f.out_pitch + 32u * r);   // uniform
        const unsigned vo = (unsigned)row + ((unsigned)b_c2 << 10);
        if constexpr (MATH) fft_radix<32>(y);
#pragma unroll
        for (int s = 0; s < 32; ++s) st_stream<true>(sgpr_base(ot + ((unsigned)s << 15)) + vo, cswap_if<INV>(y[s] * f.scale));   // k2 = p + 32 s, p = b_c2
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = lds[(b_c2 * 32 + j) * 32 + row];
        if constexpr (MATH) fft_radix<32>(y);
#pragma unroll
        for (int s = 0; s < 32; ++s) st_stream<true>(sgpr_base(ot + (16u << 10) + ((unsigned)s << 15)) + vo, cswap_if<INV>(y[s] * f.scale));   // p = 16 + b_c2
        RES_STAMP(7);                                // LDS transposes, pass B stage 1, output stores issued
      }
    }
  }
  if constexpr (STAMP) {
    if (t == 0) {
      for (int i = 0; i < 8; ++i) f.ctl->bar[256 + blockIdx.x][i] = s_words[32 + i];
      f.ctl->bar[256 + blockIdx.x][8] = k;
      f.ctl->bar[256 + blockIdx.x][9] = gslot * 32u + r;
    }
  }
}

}  // namespace mi355
