// dispatch.hpp — turns one planned Step (plan.hpp) with resolved pointers into a kernel launch.
//
// Shared by the product launcher (api.hip: HipLauncher enqueues on the device stream or a capturing
// stream — the "hipGraph stage executor" that replaces the per-stage beginComputePass/dispatchWorkgroups
// loop of src/plan.js:1233-1273) and by the host emulation in tests/emu (EmuLauncher).
//
// A Launcher provides:
//   template <class... P, class... A> void launch(void (*kernel)(P...), unsigned grid, unsigned block, unsigned smem, A&&... args);
//   void copy(void* dst, const void* src, size_t bytes);
#pragma once
#include <utility>
#include "kern_fftconv.hpp"
#include "kern_generic.hpp"
#include "kern_lines.hpp"
#include "kern_mixed.hpp"
#include "kern_mixed_ct.hpp"
#include "kern_line_reg.hpp"
#include "kern_line32k.hpp"
#include "kern_trig.hpp"
#include "kern_xcd_real.hpp"
#include "kern_regtile.hpp"
#include "kern_xcd_res.hpp"
#include "plan.hpp"

namespace mi355 {

// line-kernel families: one translation unit each so the device code builds in parallel
enum : int { FAM_ROW_SMALL = 0, FAM_ROW_1K = 1, FAM_ROW_BIG = 2, FAM_PASS_A = 3, FAM_PASS_B = 4, FAM_COUNT = 5 };
constexpr int row_family(int N) { return N <= 256 ? FAM_ROW_SMALL : (N <= 1024 ? FAM_ROW_1K : FAM_ROW_BIG); }

template <int FAMILY, class L>
bool launch_lines_family(int id, const LineArgs& a, unsigned grid, L& l) {
  int cur = 0;
#define MI_LINE_CASE(FAM, N, R0, R1, R2, T, IC, OC, SI, SO, TW)                          \
  if (id == cur) {                                                                       \
    if constexpr ((FAM) == FAMILY) {                                                     \
      using C = LineCfg<N, R0, R1, R2, T, IC, OC, SI, SO, TW>;                           \
      if constexpr ((TW) == 4) {   /* ROW_ALT_TRIG: only the one-launch DCT / DST kernels exist for these shapes */ \
        if constexpr (!(SI)) {                                                           \
          if (a.real_mode == 5 || a.real_mode == 6) { l.launch(fft_lines_r2c_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); return true; } \
        } else {                                                                         \
          if (a.real_mode == 7 || a.real_mode == 8) { l.launch(fft_lines_c2r_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); return true; } \
        }                                                                                \
        return false;                                                                    \
      } else {                                                                           \
      if constexpr (!(IC) && !(OC) && !(SI) && !(SO) && (TW) == 0 && C::NSTAGES >= 2) {   \
        if (a.real_mode == 1) {                                                          \
          if (a.mapped) l.launch(fft_lines_r2c_kernel<C, false, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          else l.launch(fft_lines_r2c_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          return true;                                                                   \
        }                                                                                \
        if (a.real_mode == 5 || a.real_mode == 6) {                                      \
          l.launch(fft_lines_r2c_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          return true;                                                                   \
        }                                                                                \
        if (a.real_mode == 4) {                                                          \
          if (a.mapped) l.launch(fft_lines_mul_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          else l.launch(fft_lines_mul_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          return true;                                                                   \
        }                                                                                \
      }                                                                                  \
      if constexpr (!(IC) && !(OC) && (SI) && (SO) && (TW) == 0) {                        \
        if (a.real_mode == 2 && !a.mapped) {                                             \
          l.launch(fft_lines_c2r_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          return true;                                                                   \
        }                                                                                \
        if constexpr (C::NSTAGES >= 2) {                                                 \
          if (a.real_mode == 2) {                                                        \
            l.launch(fft_lines_c2r_kernel<C, false, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
            return true;                                                                 \
          }                                                                              \
          if (a.real_mode == 7 || a.real_mode == 8) {                                    \
            l.launch(fft_lines_c2r_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
            return true;                                                                 \
          }                                                                              \
        }                                                                                \
      }                                                                                  \
      if (a.real_mode != 0) return false;                                                \
      if constexpr ((IC) == (OC) && (TW) == 0) {                                         \
        if (a.mapped) {                                                                  \
          l.launch(fft_lines_mapped_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
          return true;                                                                   \
        }                                                                                \
      }                                                                                  \
      if (a.mapped) return false;                                                        \
      l.launch(fft_lines_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a); \
      return true;                                                                       \
      }                                                                                  \
    } else return false;                                                                 \
  }                                                                                      \
  ++cur;
#define LINE_ROW(N, R0, R1, R2, T)                                          \
  MI_LINE_CASE(row_family(N), N, R0, R1, R2, T, false, false, false, false, 0) \
  MI_LINE_CASE(row_family(N), N, R0, R1, R2, T, false, false, true, true, 0)
#define LINE_ROW_TRIG(N, R0, R1, R2, T) \
  MI_LINE_CASE(row_family(N), N, R0, R1, R2, T, false, false, false, false, 4)
#define LINE_PASS_A(N, R0, R1, R2, T)                               \
  MI_LINE_CASE(FAM_PASS_A, N, R0, R1, R2, T, true, true, false, false, 0) \
  MI_LINE_CASE(FAM_PASS_A, N, R0, R1, R2, T, true, true, true, false, 0)  \
  MI_LINE_CASE(FAM_PASS_A, N, R0, R1, R2, T, true, true, true, true, 0)
#define LINE_PASS_B(N, R0, R1, R2, T)                                \
  MI_LINE_CASE(FAM_PASS_B, N, R0, R1, R2, T, false, true, false, false, 2) \
  MI_LINE_CASE(FAM_PASS_B, N, R0, R1, R2, T, false, true, false, true, 2)
#define LINE_COL_RAGGED(N, R0, R1, R2, T)                             \
  MI_LINE_CASE(FAM_PASS_A, N, R0, R1, R2, T, true, true, false, false, 3) \
  MI_LINE_CASE(FAM_PASS_A, N, R0, R1, R2, T, true, true, true, true, 3)
#include "line_kernels.def"
#undef LINE_ROW
#undef LINE_ROW_TRIG
#undef LINE_PASS_A
#undef LINE_PASS_B
#undef LINE_COL_RAGGED
#undef MI_LINE_CASE
  (void)cur; (void)a; (void)grid; (void)l;
  return false;
}

inline int family_of_line_kernel(const LineKernelMeta& m) {
  if (m.in_col && m.out_col) return FAM_PASS_A;
  if (!m.in_col && m.out_col) return FAM_PASS_B;
  return row_family(m.N);
}

template <class L> bool launch_fftconv_fused(int id, const FusedConvArgs& a, unsigned grid, L& l) {
  int cur = 0;
#define X(N, R0, R1, TL)                                                                               \
  if (id == cur++) {                                                                                   \
    using C = ConvCfg<N, R0, R1, TL>;                                                                  \
    l.launch(fftconv_fused_kernel<C>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a);        \
    return true;                                                                                       \
  }
  MI355_CONV_KERNEL_LIST(X)
#undef X
  (void)cur;
  return false;
}

// XCD-fused kernels.  Instance ids follow the registry order of plan.cpp (c2c forward/inverse per entry, then r2c, then c2r).
// In the product build EVERY instance is compiled in its own translation unit (lines_xcd_one.hip with -DMI355_XCD_ID=k):
// these kernels sit at the edge of the 256-VGPR budget and the register allocation of one and the same kernel changed with
// the other kernels of its translation unit (e.g. 1024x2048 c2c: 247 VGPRs and no scratch alone, 256 VGPRs and 92 B of scratch
// per lane in a unit with eleven siblings; r2c 1024x2048: 88 B against 264 B).
#define MI_XCD_PLUS2(...) +2
#define MI_XCD_PLUS1(...) +1
constexpr int XCD_INSTANCE_COUNT = 0 MI355_XCD_KERNEL_LIST(MI_XCD_PLUS2) MI355_XCD_R2C_KERNEL_LIST(MI_XCD_PLUS1) MI355_XCD_C2R_KERNEL_LIST(MI_XCD_PLUS1)
                                     MI355_XCD_2D_KERNEL_LIST(MI_XCD_PLUS2) MI355_XCD_RT_KERNEL_LIST(MI_XCD_PLUS2) + 1 + 2 + 1 + 2 + 1 + 1 + 1 + 2 + 2 MI355_XCD_VIEW_KERNEL_LIST(MI_XCD_PLUS2) + 2;   // (+ 2048 x 1024 on register tiles, forward and inverse)   // (+ the VIEW instances of the LDS-resident kernels) (+ the VIEW instances of the 32-line 1024 x 1024 kernel) (+ 1024 x 1024 on 16-line register tiles, two 256-thread workgroups per CU: forward, inverse) (+ the register-tile r2c and c2r 1024 x 2048) (+ the fftconv pipeline for 2^20 points) + the register-tile r2c 2048 x 2048 + the two-workgroups-per-CU 1024 x 1024 (forward, inverse) + the register-tile c2r 2048 x 2048 + the 32-line register-tile 1024 x 1024 (forward, inverse)
#undef MI_XCD_PLUS2
#undef MI_XCD_PLUS1

// ONLY < 0: every instance (host emulation); ONLY = k: instance k alone is instantiated, the others fall through
template <int ONLY, class L> bool launch_xcd_sel(int id, const XcdFusedArgs& a, unsigned grid, L& l);

#if defined(MI355_XCD_DEFINE_INSTANCES) || defined(MI355_HOST_EMU)
enum { MI_XCD_COUNTER_BASE = __COUNTER__ + 1 };
#define MI_XCD_CASE(KERNEL, SWAP)                                                                  \
  {                                                                                               \
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;                                         \
    if constexpr (ONLY < 0 || ONLY == ME) {                                                       \
      if (id == ME) {                                                                             \
        using CA = LineCfg<N1_, A0_, A1_, A2_, TA_, true, true, SWAP, false, 0>;                   \
        using CB = LineCfg<N2_, B0_, B1_, B2_, TB_, false, MI_XCD_B_OUT_COL, false, SWAP, 0>;      \
        using F = XcdFusedCfg<CA, CB>;                                                            \
        l.launch_concurrent(KERNEL<CA, CB>, grid, (unsigned)F::THREADS, (unsigned)F::LDS_BYTES, a); \
        return true;                                                                              \
      }                                                                                           \
    }                                                                                             \
  }
#define MI_XCD_PARAMS(N1, A0, A1, A2, TA, N2, B0, B1, B2, TB) \
  constexpr int N1_ = N1, A0_ = A0, A1_ = A1, A2_ = A2, TA_ = TA, N2_ = N2, B0_ = B0, B1_ = B1, B2_ = B2, TB_ = TB;
template <int ONLY, class L> bool launch_xcd_sel(int id, const XcdFusedArgs& a, unsigned grid, L& l) {
#define MI_XCD_B_OUT_COL true
#define X(...) { MI_XCD_PARAMS(__VA_ARGS__) MI_XCD_CASE(fft_xcd_fused_kernel, false) MI_XCD_CASE(fft_xcd_fused_kernel, true) }
  MI355_XCD_KERNEL_LIST(X)
#undef X
#define X(...) { MI_XCD_PARAMS(__VA_ARGS__) MI_XCD_CASE(fft_xcd_r2c_kernel, false) }
  MI355_XCD_R2C_KERNEL_LIST(X)
#undef X
#define X(...) { MI_XCD_PARAMS(__VA_ARGS__) MI_XCD_CASE(fft_xcd_c2r_kernel, false) }
  MI355_XCD_C2R_KERNEL_LIST(X)
#undef X
#undef MI_XCD_B_OUT_COL
#define MI_XCD_B_OUT_COL false      /* 2-D: the second pass is a ROW kernel, natural order out */
#define X(...) { MI_XCD_PARAMS(__VA_ARGS__) MI_XCD_CASE(fft_xcd_fused_kernel, false) MI_XCD_CASE(fft_xcd_fused_kernel, true) }
  MI355_XCD_2D_KERNEL_LIST(X)
#undef X
#undef MI_XCD_B_OUT_COL
  // register-tile instances (kern_regtile.hpp): forward, then inverse, per entry
#define MI_XCD_RT_CASE(N1, INV)                                                                                   \
  {                                                                                                               \
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;                                                         \
    if constexpr (ONLY < 0 || ONLY == ME) {                                                                       \
      if (id == ME) {                                                                                             \
        using F = XcdRtCfg<N1, INV>;                                                                              \
        l.launch_concurrent(fft_xcd_rt_kernel<N1, INV>, grid, (unsigned)F::THREADS, (unsigned)F::LDS_BYTES, a);   \
        return true;                                                                                              \
      }                                                                                                           \
    }                                                                                                             \
  }
#define X(N1) MI_XCD_RT_CASE(N1, false) MI_XCD_RT_CASE(N1, true)
  MI355_XCD_RT_KERNEL_LIST(X)
#undef X
#undef MI_XCD_RT_CASE
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt_r2c_kernel<2048>, grid, (unsigned)XcdRtR2cCfg::THREADS, (unsigned)XcdRtR2cCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_hx_kernel<false>, grid, (unsigned)HxCfg::THREADS, (unsigned)HxCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_hx_kernel<true>, grid, (unsigned)HxCfg::THREADS, (unsigned)HxCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt_c2r_kernel<2048>, grid, (unsigned)XcdRtR2cCfg::THREADS, (unsigned)XcdRtR2cCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<false>, grid, (unsigned)Rt1kCfg::THREADS, (unsigned)Rt1kCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<true>, grid, (unsigned)Rt1kCfg::THREADS, (unsigned)Rt1kCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_conv1m_kernel<1024>, grid, (unsigned)Rt1kCfg::THREADS, (unsigned)Rt1kCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt_r2c_kernel<1024>, grid, (unsigned)XcdRtR2cCfgN<1024>::THREADS, (unsigned)XcdRtR2cCfgN<1024>::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt_c2r_kernel<1024>, grid, (unsigned)XcdRtR2cCfgN<1024>::THREADS, (unsigned)XcdRtR2cCfgN<1024>::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<false, 16>, grid, (unsigned)Rt1kCfgT<16>::THREADS, (unsigned)Rt1kCfgT<16>::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<true, 16>, grid, (unsigned)Rt1kCfgT<16>::THREADS, (unsigned)Rt1kCfgT<16>::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<false, 32, true>, grid, (unsigned)Rt1kCfg::THREADS, (unsigned)Rt1kCfg::LDS_BYTES, a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<true, 32, true>, grid, (unsigned)Rt1kCfg::THREADS, (unsigned)Rt1kCfg::LDS_BYTES, a); return true; }
    }
  }
#define MI_XCD_CASE_VIEW(SWAP)                                                                         \
  {                                                                                                   \
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;                                             \
    if constexpr (ONLY < 0 || ONLY == ME) {                                                           \
      if (id == ME) {                                                                                 \
        using CA = LineCfg<N1_, A0_, A1_, A2_, TA_, true, true, SWAP, false, 0>;                       \
        using CB = LineCfg<N2_, B0_, B1_, B2_, TB_, false, true, false, SWAP, 0>;                      \
        using F = XcdFusedCfg<CA, CB>;                                                                \
        l.launch_concurrent(fft_xcd_fused_kernel<CA, CB, true>, grid, (unsigned)F::THREADS, (unsigned)F::LDS_BYTES, a); \
        return true;                                                                                  \
      }                                                                                               \
    }                                                                                                 \
  }
#define X(...) { MI_XCD_PARAMS(__VA_ARGS__) MI_XCD_CASE_VIEW(false) MI_XCD_CASE_VIEW(true) }
  MI355_XCD_VIEW_KERNEL_LIST(X)
#undef X
#undef MI_XCD_CASE_VIEW
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<false, 32, false, 2048>, grid, 512u, (unsigned)(Rt1kCfg::LDS_BYTES + RtCfg::TW2_ELEMS * 8), a); return true; }
    }
  }
  {
    constexpr int ME = __COUNTER__ - MI_XCD_COUNTER_BASE;
    if constexpr (ONLY < 0 || ONLY == ME) {
      if (id == ME) { l.launch_concurrent(fft_xcd_rt1k_kernel<true, 32, false, 2048>, grid, 512u, (unsigned)(Rt1kCfg::LDS_BYTES + RtCfg::TW2_ELEMS * 8), a); return true; }
    }
  }
  static_assert(__COUNTER__ - MI_XCD_COUNTER_BASE == XCD_INSTANCE_COUNT, "instance ids out of step with the lists");
  return false;
}
#undef MI_XCD_CASE
#undef MI_XCD_PARAMS
#endif

template <class L, int... Is>
bool launch_xcd_fold(int id, const XcdFusedArgs& a, unsigned grid, L& l, std::integer_sequence<int, Is...>) {
  return ((id == Is && launch_xcd_sel<Is, L>(id, a, grid, l)) || ...);
}
template <class L> bool launch_xcd_fused(int id, const XcdFusedArgs& a, unsigned grid, L& l) {
#ifdef MI355_HOST_EMU
  return launch_xcd_sel<-1, L>(id, a, grid, l);
#else
  return launch_xcd_fold(id, a, grid, l, std::make_integer_sequence<int, XCD_INSTANCE_COUNT>{});
#endif
}

// XCD-resident kernel (kern_xcd_res.hpp): variant bit 0 = inverse, bit 1 = data-movement skeleton.  Own translation unit
// (xcd_res_kernel.hip) in the product build, like the fused instances.
template <class L> bool launch_xcd_res(int variant, const XcdFusedArgs& a, unsigned grid, L& l);
#if defined(MI355_XCD_RES_DEFINE_INSTANCES) || defined(MI355_HOST_EMU)
template <class L> bool launch_xcd_res(int variant, const XcdFusedArgs& a, unsigned grid, L& l) {
  constexpr unsigned T = (unsigned)XcdResCfg::THREADS, S = (unsigned)XcdResCfg::LDS_BYTES;
  switch (variant) {
    case 0: l.launch_concurrent(fft_xcd_res_kernel<false, true>, grid, T, S, a); return true;
    case 1: l.launch_concurrent(fft_xcd_res_kernel<true, true>, grid, T, S, a); return true;
    case 2: l.launch_concurrent(fft_xcd_res_kernel<false, false>, grid, T, S, a); return true;
    case 3: l.launch_concurrent(fft_xcd_res_kernel<true, false>, grid, T, S, a); return true;
    case 4: l.launch_concurrent(fft_xcd_res_kernel<false, true, true>, grid, T, S, a); return true;    // diagnostic: in-kernel phase stamps
    case 6: l.launch_concurrent(fft_xcd_res_kernel<false, false, true>, grid, T, S, a); return true;
  }
  return false;
}
#endif

// mixed-radix kernels with compile-time plans: own translation unit (mixed_ct_kernels.hip) in the product build
template <class L> bool launch_mixedct(int id, const MixedArgs& a, unsigned grid, L& l);
#if defined(MI355_MIXEDCT_DEFINE_INSTANCES) || defined(MI355_HOST_EMU)
template <class L> bool launch_mixedct(int id, const MixedArgs& a, unsigned grid, L& l) {
  int cur = 0;
#define X(N, T, TH, ...)                                                                                   \
  if (id == cur++) {                                                                                       \
    using M = MixedCt<N, T, TH, __VA_ARGS__>;                                                              \
    l.launch(fft_lines_mixedct_kernel<M>, grid, (unsigned)M::THREADS, (unsigned)M::LDS_BYTES, a);        \
    return true;                                                                                           \
  }
  MI355_MIXEDCT_LIST(X)
#undef X
  (void)cur;
  return false;
}
#endif
// lines of 2^13 / 2^14 / 2^15 points held in the registers of one workgroup (kern_line_reg.hpp); same translation unit as the
// mixed-radix instances.  lg = log2 N.
template <class L> bool launch_line_reg(int lg, const MixedArgs& a, unsigned grid, L& l);
#if defined(MI355_MIXEDCT_DEFINE_INSTANCES) || defined(MI355_HOST_EMU)
template <class C, class L> void launch_line_reg_cfg(const MixedArgs& a, unsigned grid, L& l) {
  if (a.swap_in) l.launch(fft_line_reg_kernel<C, true>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a);
  else l.launch(fft_line_reg_kernel<C, false>, grid, (unsigned)C::THREADS, (unsigned)C::LDS_BYTES, a);
}
template <class L> bool launch_line_reg(int lg, const MixedArgs& a, unsigned grid, L& l) {
  if (lg == 12) { launch_line_reg_cfg<LineRegCfg<4096, 16, 8>>(a, grid, l); return true; }
  if (lg == 13) { launch_line_reg_cfg<LineRegCfg<8192, 16, 16>>(a, grid, l); return true; }
  if (lg == 14) { launch_line_reg_cfg<LineRegCfg<16384, 32, 16>>(a, grid, l); return true; }
  if (lg == 15) {   // the dedicated form of the same scheme (kern_line32k.hpp): the generic template spills 344 B per lane at this size (177 vs 288 GPoints/s)
    if (a.swap_in) l.launch(fft_line32k_kernel<true>, grid, (unsigned)Line32kCfg::THREADS, (unsigned)Line32kCfg::LDS_BYTES, a);
    else l.launch(fft_line32k_kernel<false>, grid, (unsigned)Line32kCfg::THREADS, (unsigned)Line32kCfg::LDS_BYTES, a);
    return true;
  }
  return false;
}
#endif

template <class L> bool launch_stage(int radix, const StageArgs& a, unsigned grid, L& l) {
  switch (radix) {
#define MI_STAGE_CASE(R) case R: l.launch(stockham_stage_kernel<R>, grid, 256u, 0u, a); return true;
    MI_STAGE_CASE(2) MI_STAGE_CASE(3) MI_STAGE_CASE(4) MI_STAGE_CASE(5) MI_STAGE_CASE(7) MI_STAGE_CASE(8)
    MI_STAGE_CASE(11) MI_STAGE_CASE(13) MI_STAGE_CASE(16) MI_STAGE_CASE(32)
#undef MI_STAGE_CASE
  }
  return false;
}

// LinesFn: bool(int family, int id, const LineArgs&, unsigned grid) — supplied by the caller because the
// per-family instantiations live in different translation units in the product build.
template <class L, class LinesFn, class XcdFn>
bool dispatch_step(const Step& s, void* const ptr[5], L& l, LinesFn&& lines_fn, XcdFn&& xcd_fn) {
  switch (s.kind) {
    case ST_XCD_RES:
    case ST_XCD_FUSED: {
      XcdFusedArgs a{};
      a.in = (const cf*)ptr[0]; a.out = (cf*)ptr[1]; a.wslots = (cf*)ptr[2]; a.ctl = (XcdCtl*)ptr[3];
      const char* tb = (const char*)ptr[4];
      a.tw_a = (const cf*)(tb + s.i[4]); a.tw_b = (const cf*)(tb + s.i[5]); a.tw_lo = (const cf*)(tb + s.i[6]); a.tw_hi = (const cf*)(tb + s.i[7]);
      a.num_transforms = s.i[0]; a.N = s.i[1]; a.fs_shift = (int)s.i[2]; a.fs_lo_mask = (unsigned)s.i[3];
      a.in_pitch = s.i[9]; a.out_pitch = s.i[10];
      a.scale = s.f[0];
      a.sticky_error = l.sticky_error_word();
      a.spin_limit = s.i[13] > 0 ? (unsigned)s.i[13] : 4000000u;
      a.split = (unsigned)s.i[8]; a.slots = (unsigned)s.i[11]; a.solo = (unsigned)s.i[12];
      a.conv_k = (unsigned)s.i[15]; a.conv_conj = (unsigned)s.i[16]; a.out_kernel_pitch = s.i[17];
      a.mul = a.conv_k ? (const cf*)((const char*)ptr[2] + s.i[14]) : nullptr;   // the kernel spectra sit in the same workspace arena as the slots
      a.v_in_lo = (int)s.imap.lo[0]; a.v_in_hi = (int)s.imap.hi[0]; a.v_out_lo = (int)s.omap.lo[0]; a.v_out_hi = (int)s.omap.hi[0];
      a.v_zlo = (int)s.omap.zlo[0]; a.v_zhi = (int)s.omap.zhi[0];   // (VIEW instances only; the planner fills the two maps)
      if (s.kind == ST_XCD_RES) return launch_xcd_res(s.variant, a, s.grid, l);
      return xcd_fn(s.variant, a, s.grid);
    }
    case ST_LINES: {
      LineArgs a{};
      a.in = (const cf*)ptr[0]; a.out = (cf*)ptr[1]; a.tw = (const cf*)ptr[2]; a.tw_lo = (const cf*)ptr[3]; a.tw_hi = (const cf*)ptr[4];
      a.num_tiles = s.i[0]; a.num_lines = s.i[1];
      a.in_S = s.i[2]; a.in_outer_stride = s.i[3]; a.out_S = s.i[4]; a.out_outer_stride = s.i[5];
      a.fs_shift = (int)s.i[6]; a.fs_lo_mask = (unsigned)s.i[7]; a.fs_group = s.i[8] ? s.i[8] : 1; a.real_mode = (int)s.i[9];
      a.mapped = (int)s.i[10];
      if (a.mapped) { a.imap = s.imap; a.omap = s.omap; }
      a.scale = s.f[0];
      const LineKernelMeta& m = line_kernel_registry()[(size_t)s.variant];
      return lines_fn(family_of_line_kernel(m), s.variant, a, s.grid);
    }
    case ST_TRIG_PRE:
    case ST_TRIG_POST: {
      TrigArgs a{};
      a.x = (const float*)ptr[0]; a.z = (cf*)ptr[1]; a.y = (float*)ptr[2];
      a.lines = s.i[0]; a.N = s.i[1]; a.L = s.i[2]; a.S = s.i[3]; a.kind = (int)s.i[4]; a.scale = s.f[0]; a.stride = s.i[5] ? s.i[5] : 1;
      if (a.kind >= 8 && a.stride > 1) {
        if (s.kind == ST_TRIG_PRE) l.launch(trig_real_pre_tiled_kernel, s.grid, 256u, 0u, a); else l.launch(trig_real_post_tiled_kernel, s.grid, 256u, 0u, a);
      } else if (a.kind >= 8) {
        if (s.kind == ST_TRIG_PRE) l.launch(trig_real_pre_kernel, s.grid, 256u, 0u, a); else l.launch(trig_real_post_kernel, s.grid, 256u, 0u, a);
      } else if (s.kind == ST_TRIG_PRE) l.launch(trig_pre_kernel, s.grid, 256u, 0u, a); else l.launch(trig_post_kernel, s.grid, 256u, 0u, a);
      return true;
    }
    case ST_LINES_MIXED: {
      MixedArgs a{};
      a.in = (const cf*)ptr[0]; a.out = (cf*)ptr[1]; a.tw = (const cf*)ptr[2];
      a.lines = s.i[0]; a.N = (int)s.i[1]; a.S = s.i[2]; a.T = (int)s.i[3]; a.nst = (int)s.i[4];
      a.swap_in = a.swap_out = (int)s.i[5];
      a.scale = s.f[0];
      if (s.variant >= 1000) return launch_line_reg(s.variant - 1000, a, s.grid, l);   // 2^13 .. 2^15 in one workgroup's registers (kern_line_reg.hpp)
      if (s.variant > 0) return launch_mixedct(s.variant - 1, a, s.grid, l);   // compile-time plan: nothing else to pass
      {
        const auto rcp = [](unsigned d) { return d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u; };
        unsigned nsp = 1;
        for (int k = 0; k < a.nst; ++k) {
          a.radix[k] = (int)(s.i[8 + k] >> 32); a.tw_off[k] = (int)(s.i[8 + k] & 0xffffffff);
          a.rcp_nb[k] = rcp((unsigned)(a.N / a.radix[k])); a.rcp_nsp[k] = rcp(nsp);
          nsp *= (unsigned)a.radix[k];
        }
      }
      a.lds_bytes = (int)s.i[6]; a.tw_total = (int)s.i[19];
      l.launch(fft_lines_mixed_kernel, s.grid, (unsigned)s.i[7], (unsigned)(a.lds_bytes + MIXED_MAX_T * 8 + a.tw_total * 8), a);
      return true;
    }
    case ST_STAGE: {
      StageArgs a{};
      a.in = (const cf*)ptr[0]; a.out = (cf*)ptr[1]; a.tw = (const cf*)ptr[2];
      a.total = s.i[0]; a.N = s.i[1]; a.S = s.i[2]; a.Nsp = s.i[3]; a.swap_in = (int)s.i[4]; a.swap_out = (int)s.i[5];
      a.scale = s.f[0];
      return launch_stage(s.variant, a, s.grid, l);
    }
    case ST_R2C_POST: {
      R2cPostArgs a{};
      a.z = (const cf*)ptr[0]; a.x = (cf*)ptr[1]; a.tw_lo = (const cf*)ptr[2]; a.tw_hi = (const cf*)ptr[3];
      a.H = s.i[0]; a.batch = s.i[1]; a.x_line_stride = s.i[2]; a.scale = s.f[0]; a.shift = (int)s.i[3]; a.mask = (unsigned)s.i[4];
      l.launch(r2c_post_kernel, s.grid, 256u, 0u, a);
      return true;
    }
    case ST_C2R_PRE: {
      C2rPreArgs a{};
      a.x = (const cf*)ptr[0]; a.z = (cf*)ptr[1]; a.tw_lo = (const cf*)ptr[2]; a.tw_hi = (const cf*)ptr[3];
      a.H = s.i[0]; a.batch = s.i[1]; a.x_line_stride = s.i[2]; a.shift = (int)s.i[3]; a.mask = (unsigned)s.i[4];
      l.launch(c2r_pre_kernel, s.grid, 256u, 0u, a);
      return true;
    }
    case ST_REAL_TO_COMPLEX:
      l.launch(real_to_complex_kernel, s.grid, 256u, 0u, (const float*)ptr[0], (cf*)ptr[1], (long long)s.i[0]);
      return true;
    case ST_COMPLEX_TO_REAL:
      l.launch(complex_to_real_kernel, s.grid, 256u, 0u, (const cf*)ptr[0], (float*)ptr[1], (long long)s.i[0], s.f[0]);
      return true;
    case ST_PACK_HALF:
      l.launch(pack_half_kernel, s.grid, 256u, 0u, (const cf*)ptr[0], (cf*)ptr[1], (long long)s.i[0], (long long)s.i[1], (long long)s.i[2],
               (long long)s.i[3], s.f[0]);
      return true;
    case ST_UNPACK_HERM:
      l.launch(unpack_hermitian_kernel, s.grid, 256u, 0u, (const cf*)ptr[0], (cf*)ptr[1], (long long)s.i[0], (long long)s.i[1],
               (long long)s.i[2], (long long)s.i[3]);
      return true;
    case ST_POINTWISE:
      l.launch(pointwise_mul_kernel, s.grid, 256u, 0u, (const cf*)ptr[0], (cf*)ptr[1], (const cf*)ptr[2], (long long)s.i[0], (long long)s.i[1],
               (int)s.i[2], s.f[0]);
      return true;
    case ST_GATHER:
    case ST_SCATTER: {
      StridedArgs a{};
      a.src = ptr[0]; a.dst = ptr[1];
      a.total = s.i[0]; a.per = s.i[1]; a.rank = (int)s.i[2];
      a.phys_offset = s.i[3]; a.phys_batch_stride = s.i[4]; a.dense_offset = s.i[5]; a.dense_batch_stride = s.i[6];
      for (int d = 0; d < 8; ++d) { a.shape[d] = s.shape[d] ? s.shape[d] : 1; a.phys_stride[d] = s.sa[d]; a.dense_stride[d] = s.sb[d]; }
      const bool real = s.i[7] != 0;     // element type: 0 complex, 1 real (r2c input / c2r output sides)
      if (s.kind == ST_GATHER) { if (real) l.launch(strided_copy_kernel<true, float>, s.grid, 256u, 0u, a); else l.launch(strided_copy_kernel<true, cf>, s.grid, 256u, 0u, a); }
      else { if (real) l.launch(strided_copy_kernel<false, float>, s.grid, 256u, 0u, a); else l.launch(strided_copy_kernel<false, cf>, s.grid, 256u, 0u, a); }
      return true;
    }
    case ST_FFTCONV_FUSED: {
      FusedConvArgs a{};
      a.in = (const cf*)ptr[0]; a.kern = (const cf*)ptr[1]; a.out = (cf*)ptr[2]; a.tw = (const cf*)ptr[3];
      a.batch = s.i[0]; a.K = (int)s.i[1]; a.kern_len = (int)s.i[2]; a.conj_kernel = (int)s.i[3];
      a.in_offset = s.i[4]; a.in_batch_stride = s.i[5]; a.in_stride = s.i[6];
      a.out_offset = s.i[7]; a.out_kernel_stride = s.i[8]; a.out_batch_stride = s.i[9]; a.out_stride = s.i[10];
      a.scale = s.f[0];
      return launch_fftconv_fused(s.variant, a, s.grid, l);
    }
    case ST_CHIRP_PRE:
    case ST_CHIRP_POST: {
      ChirpArgs a{};
      a.in = (const cf*)ptr[0]; a.out = (cf*)ptr[1]; a.chirp = (const cf*)ptr[2];
      a.N = s.i[0]; a.M = s.i[1]; a.lines = s.i[2]; a.swap_in = (int)s.i[3]; a.swap_out = (int)s.i[4]; a.scale = s.f[0];
      if (s.kind == ST_CHIRP_PRE) l.launch(bluestein_pre_kernel, s.grid, 256u, 0u, a);
      else l.launch(bluestein_post_kernel, s.grid, 256u, 0u, a);
      return true;
    }
    case ST_ZERO_OUTSIDE: {
      ZeroOutsideArgs a{};
      a.data = ptr[0]; a.total = s.i[0]; a.per = s.i[1]; a.rank = (int)s.i[2];
      for (int d = 0; d < 8; ++d) { a.shape[d] = s.shape[d] ? s.shape[d] : 1; a.start[d] = s.sa[d]; a.end[d] = d < a.rank ? s.sb[d] : 1; }
      if (s.i[3]) l.launch(zero_outside_kernel<float>, s.grid, 256u, 0u, a); else l.launch(zero_outside_kernel<cf>, s.grid, 256u, 0u, a);
      return true;
    }
    case ST_ZERO:
      l.launch(zero_kernel, s.grid, 256u, 0u, (float*)ptr[0], (long long)s.i[0]);
      return true;
    case ST_SCALE:
      l.launch(scale_kernel, s.grid, 256u, 0u, (float*)ptr[0], (long long)s.i[0], s.f[0]);
      return true;
    case ST_COPY:
      if (ptr[0] != ptr[1]) l.copy(ptr[1], ptr[0], (size_t)s.i[0]);
      return true;
  }
  return false;
}

}  // namespace mi355
