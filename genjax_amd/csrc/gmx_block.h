// gmx_block.h — wave / block reductions shared by the AOT kernels and the
// hiprtc-specialised site programs (device code only).
//
// Fixed butterfly order: every lane ends with the same bits and the result
// does not depend on scheduling, so block partials are reproducible.
#pragma once
#include "gmx_math.h"

#define GMX_BLOCK 256
#define GMX_WAVE 64

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = gmx_fmax(v, __shfl_xor(v, m, GMX_WAVE));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, GMX_WAVE);
  return v;
}
// block of 256 threads = 4 waves; result broadcast to all threads
__device__ __forceinline__ float block_max(float v, float* lds4) {
  v = wave_max(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  float r = gmx_fmax(gmx_fmax(lds4[0], lds4[1]), gmx_fmax(lds4[2], lds4[3]));
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* lds4) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  float r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  return r;
}

// order-preserving map float -> uint32 (larger float <=> larger key; every key > 0)
__device__ __forceinline__ uint32_t gmx_max_key(float x) {
  uint32_t b = gmx_f2u(x);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float gmx_max_unkey(uint32_t k) {
  return gmx_u2f((k >> 31) ? (k & 0x7fffffffu) : ~k);
}

// OP_REDMAX / OP_REDLSE epilogues.  `row` = index of the 256-particle group.
__device__ __forceinline__ void gmx_red_max(float* red_out, uint32_t* bins, float* lds4, uint32_t row, float x,
                                            bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  if (threadIdx.x == 0) {
    if (red_out) red_out[2 * (size_t)row] = m;
    if (bins && !gmx_isnan(m)) atomicMax(&bins[row & 31u], gmx_max_key(m));
  }
}
__device__ __forceinline__ void gmx_red_lse(float* red_out, float* lds4, uint32_t row, float x, bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  float e = active ? gmx_expf(x - m) : 0.0f;
  if (!(m > -gmx_inf())) e = 0.0f;  // empty / all -inf block
  float s = block_sum(e, lds4);
  if (threadIdx.x == 0 && red_out) {
    red_out[2 * (size_t)row] = m;
    red_out[2 * (size_t)row + 1] = s;
  }
}
