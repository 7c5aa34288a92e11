// gmx_jit.h — device side of a specialised site program.
//
// gmx_program_specialize() emits a translation unit of the form
//
//   #include "gmx_jit.h"
//   GMX_JIT_PROGRAM(N_INSTR, N_REGS, FULL, N_CONST_BASE) = { w0, w1, ... };
//   GMX_JIT_CONSTS = { c0, c1, ... };
//   GMX_JIT_KERNEL(N_INSTR, N_REGS, FULL, N_UNI_DYN)
//
// i.e. the SAME interpreter template (gmx_vm.h) instantiated with a context
// whose instruction fetch and constant pool are compile-time constants and a
// fully unrolled op loop.  hipcc folds the dispatch switch, the operand
// selection and the register indexing: what remains is straight-line code
// calling the hand-written samplers / log-densities — bit-identical to the
// interpreter by construction (same source, same -ffp-contract=off).
#pragma once
#include "gmx_block.h"
#include "gmx_vm.h"

template <int NI, int NDYN>
struct gmx_jit_ctx {
  const uint32_t* prog;     // constexpr instruction words
  const uint32_t* consts;   // constexpr constants (pool entries NDYN..)
  const gmx_run_args* A;
  float* lds4;
  __device__ __forceinline__ void fetch(uint32_t pc, uint32_t* w0, uint32_t* w1) const {
    *w0 = prog[2u * pc]; *w1 = prog[2u * pc + 1u];
  }
  __device__ __forceinline__ uint32_t pool(uint32_t i) const {
    return i < (uint32_t)NDYN ? A->uni[i] : consts[i - (uint32_t)NDYN];
  }
  __device__ __forceinline__ const void* in_ptr(uint32_t s) const { return A->in_d[s]; }
  __device__ __forceinline__ void* out_ptr(uint32_t s) const { return A->out_d[s]; }
  __device__ __forceinline__ const void* tab_ptr(uint32_t s) const { return A->tab_d[s]; }
  __device__ __forceinline__ void red_max(float x, bool active) { gmx_red_max(A->red_out_d, lds4, x, active); }
  __device__ __forceinline__ void red_lse(float x, bool active) { gmx_red_lse(A->red_out_d, lds4, x, active); }
};

#define GMX_JIT_KERNEL(NI, NREGS, FULL, NDYN)                                                   \
  extern "C" __global__ void __launch_bounds__(GMX_BLOCK) gmx_jit_kernel(int64_t n, const gmx_run_args A) { \
    __shared__ float lds4[4];                                                                    \
    int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;                                   \
    gmx_jit_ctx<NI, NDYN> ctx;                                                                   \
    ctx.prog = GMX_JIT_PROG; ctx.consts = GMX_JIT_CONST; ctx.A = &A; ctx.lds4 = lds4;            \
    gmx_vm_run<gmx_regs_vgpr<NREGS>, FULL, NI, gmx_jit_ctx<NI, NDYN>>((uint32_t)NI, i, i < n, A, ctx); \
  }
