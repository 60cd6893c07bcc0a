// platform.hpp — the one include every kernel header uses.
//
// Product build (hipcc, --offload-arch=gfx950): pulls in the HIP runtime.
//
// MI355_HOST_EMU build (tests only, tests/emu/): the SAME kernel sources are compiled as plain C++ for the
// host so that index math, LDS layouts and barrier placement can be unit-tested, and run under
// ASan/UBSan, in the GPU-less build container.  One std::thread per GPU thread, a pthread barrier for
// __syncthreads().  It is a test harness: nothing in libmi355fft.so or the addon is built from it and
// there is no runtime fallback to it (api.hip fails loudly when no HIP device is present).
#pragma once

#ifndef MI355_HOST_EMU
#include <hip/hip_runtime.h>
#define MI_SMEM_DECL(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define MI_SMEM_DECL_STATIC(type, name, n) __shared__ type name[n]
#define MI_WAVE_SYNC()                                       \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)
// inter-workgroup primitives of the XCD-fused kernel (kern_xcd.hpp)
#define MI_XCC_ID() (__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 0xf)
#define MI_ATOMIC_ADD_U32(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MI_ATOMIC_LOAD_U32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MI_ATOMIC_ADD_U64(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MI_ATOMIC_LOAD_U64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MI_ATOMIC_OR_U32(p, v) __hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MI_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MI_ACQUIRE_AGENT() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define MI_SLEEP() __builtin_amdgcn_s_sleep(1)
#else
#include <cmath>
#include <cstdint>
#include <cstring>
namespace emu {
struct dim3_t { unsigned x = 1, y = 1, z = 1; };
extern thread_local dim3_t t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local char* t_smem;
extern unsigned g_xcds;   // emulated XCD count: block b reports XCC id b % g_xcds
void sync_threads();
void sync_wave();    // the 64 emulated threads of one wave (kern_xcd_res.hpp: waves of a workgroup wait on different counters)
void yield_thread();
}  // namespace emu
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
#define __syncthreads() emu::sync_threads()
#define MI_SMEM_DECL(name) char* name = emu::t_smem
#define MI_SMEM_DECL_STATIC(type, name, n) type* name = reinterpret_cast<type*>(emu::t_smem)
#define MI_XCC_ID() (emu::t_blockIdx.x % emu::g_xcds)
#define MI_ATOMIC_ADD_U32(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define MI_ATOMIC_LOAD_U32(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define MI_ATOMIC_ADD_U64(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define MI_ATOMIC_LOAD_U64(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define MI_ATOMIC_OR_U32(p, v) __atomic_fetch_or((p), (v), __ATOMIC_SEQ_CST)
#define MI_WAIT_VMEM() do { } while (0)
#define MI_ACQUIRE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define MI_SLEEP() emu::yield_thread()   /* bounded spins must outlast thread start-up of the other emulated blocks */
#define MI_WAVE_SYNC() emu::sync_threads() /* emulated waves are not lock-step: use the block barrier */
#endif

// high 32 bits of a 32x32-bit product (v_mul_hi_u32): division by an invariant through a precomputed reciprocal
#ifdef MI355_HOST_EMU
#define MI_UMULHI(a, b) ((unsigned)(((unsigned long long)(a) * (unsigned long long)(b)) >> 32))
#else
#define MI_UMULHI(a, b) __umulhi((a), (b))
#endif

// keeps the instruction scheduler from moving anything across this point (caps the loads in flight, and with them the
// registers they pin, in long unrolled load/compute sequences)
#ifdef MI355_HOST_EMU
#define MI_SCHED_FENCE() do { } while (0)
#else
#define MI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
