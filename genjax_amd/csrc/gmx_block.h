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

// OP_REDMAX / OP_REDLSE epilogues.  `row` = index of the 256-particle group,
// `rows` = number of groups; partials are planar: max[rows] then sumexp[rows].
__device__ __forceinline__ void gmx_red_max(float* red_out, float* lds4, uint32_t row, float x, bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  if (threadIdx.x == 0 && red_out) red_out[row] = m;
}
__device__ __forceinline__ void gmx_red_lse(float* red_out, float* lds4, uint32_t row, uint32_t rows, float x,
                                            bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  float e = active ? gmx_expf(x - m) : 0.0f;
  if (!(m > -gmx_inf())) e = 0.0f;  // empty / all -inf block
  float s = block_sum(e, lds4);
  if (threadIdx.x == 0 && red_out) {
    red_out[row] = m;
    red_out[(size_t)rows + row] = s;
  }
}

// ---- 64-bit wave primitives and the fixed-point weight of the two-level CDF (include/genmi.h) ----
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_up(lo, d, GMX_WAVE);
  hi = __shfl_up(hi, d, GMX_WAVE);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, GMX_WAVE); hi = __shfl_xor(hi, m, GMX_WAVE);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// q = floor(exp(lw - ref) * 2^shift) as u64; NaN / negative / not below 2^63 (lw = +inf) -> 0
__device__ __forceinline__ uint64_t weight_fixed(float lw, float ref, float scale) {
  float w = gmx_expf(lw - ref);        // NaN if lw - ref is NaN
  float q = w * scale;                 // exact: scale is a power of two
  if (!(q >= 0.0f) || !(q < 0x1p63f)) return 0ull;
  return (uint64_t)q;                  // truncation
}
