// gmx_jit.h — device side of a specialised site program.
//
// gmx_program_specialize() emits a translation unit of the form
//
//   #include "gmx_jit.h"
//   __device__ static constexpr uint32_t GMX_JIT_CONST[] = { c0, c1, ... };
//   GMX_JIT_BEGIN(N_REGS, FULL, N_DYN)
//     GMX_JIT_OP(w0, w1)          // one line per instruction
//     ...
//   GMX_JIT_END
//
// i.e. the SAME gmx_vm_step template (gmx_vm.h) the interpreter loops over,
// instantiated once per instruction with compile-time instruction words
// (gmx_cword): the dispatch switch, the operand selection and the register
// indices are constant expressions, so what is compiled is straight-line code
// calling the hand-written samplers / log-densities — bit-identical to the
// interpreter by construction (same source, same -ffp-contract=off).
#pragma once
#include "gmx_block.h"
#include "gmx_vm.h"

template <int NDYN>
struct gmx_jit_ctx {
  const uint32_t* consts;   // constexpr constants (pool entries NDYN..)
  const gmx_run_args* A;
  float* lds4;
  __device__ __forceinline__ uint32_t pool(uint32_t i) const {
    return i < (uint32_t)NDYN ? A->uni[i] : consts[i - (uint32_t)NDYN];
  }
  __device__ __forceinline__ const void* in_ptr(uint32_t s) const { return A->in_d[s]; }
  __device__ __forceinline__ void* out_ptr(uint32_t s) const { return A->out_d[s]; }
  __device__ __forceinline__ const void* tab_ptr(uint32_t s) const { return A->tab_d[s]; }
  __device__ __forceinline__ void red_max(float x, bool active) { gmx_red_max(A->red_out_d, lds4, x, active); }
  __device__ __forceinline__ void red_lse(float x, bool active) { gmx_red_lse(A->red_out_d, lds4, x, active); }
};

#define GMX_JIT_BEGIN(NREGS, FULLV, NDYN)                                                        \
  extern "C" __global__ void __launch_bounds__(GMX_BLOCK) gmx_jit_kernel(int64_t n, const gmx_run_args A) { \
    __shared__ float lds4[4];                                                                    \
    const int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;                             \
    const bool active = i < n;                                                                   \
    typedef gmx_regs_vgpr<NREGS> regs_t;                                                         \
    typedef gmx_jit_ctx<NDYN> ctx_t;                                                             \
    constexpr bool full_v = FULLV;                                                               \
    ctx_t ctx;                                                                                   \
    ctx.consts = GMX_JIT_CONST; ctx.A = &A; ctx.lds4 = lds4;                                     \
    regs_t R;                                                                                    \
    R.init();

#define GMX_JIT_OP(W0, W1) \
    gmx_vm_step<regs_t, full_v, gmx_cword<W0, W1>, ctx_t>(R, gmx_cword<W0, W1>(), i, active, A, ctx);

#define GMX_JIT_END }
