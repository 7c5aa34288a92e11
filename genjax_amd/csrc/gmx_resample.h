// gmx_resample.h — device code of the ordered (systematic / stratified) resamplers, shared by the AOT kernels
// (gmx_kernels.hip: k_offspring, k_offspring_tile, k_shard_*).  Device only.
//
// ancestor(j) = first i with cdf_i * D > P_j (128-bit integers), D = n_out * 2^23, P_j = (j * 2^23 + u_j) * total;
// P_j increases with j, so source i owns the slot range [f(cdf_{i-1}), f(cdf_i)), f(c) = #{ j : P_j < c * D }.
#pragma once
#include "gmx_block.h"
#include "gmx_rng.h"
#include "genmi.h"

struct u128 { uint64_t hi, lo; };
__device__ __forceinline__ u128 mul64(uint64_t a, uint64_t b) {
  u128 r; r.lo = a * b; r.hi = __umul64hi(a, b); return r;
}
__device__ __forceinline__ bool gt128(u128 a, u128 b) {
  return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo);
}

// P_j for the ordered kinds
__device__ __forceinline__ u128 slot_threshold(int kind, gmx_key key, uint64_t u0, int64_t j, uint64_t total) {
  uint64_t u = (kind == GMX_RESAMPLE_SYSTEMATIC) ? u0 : (uint64_t)(gmx_bits32(key, (uint64_t)j) >> 9);
  return mul64(((uint64_t)j << 23) + u, total);
}

// The exact predicate (runs for a handful of sources per launch).  Kept inline — a real call gives the kernel a
// stack (scratch), which costs more at wave launch than the code size does — so callers on the hot path take
// `kind` as a template constant: the systematic instantiation then carries no Threefry at all.
__device__ __forceinline__ int64_t slots_below_exact(int kind, gmx_key key, uint64_t u0, uint64_t c, uint64_t D,
                                                  uint64_t total, int64_t j, int64_t n_out) {
  u128 X = mul64(c, D);
  while (j > 0 && !gt128(X, slot_threshold(kind, key, u0, j - 1, total))) --j;
  while (j < n_out && gt128(X, slot_threshold(kind, key, u0, j, total))) ++j;
  return j;
}

// ---- slot ranges, fast path without branches ----
// slots_below() above decides one CDF value with nested branches and carries the exact 128-bit predicate inline;
// k_offspring_tile evaluates it five times per thread, so here the f64 estimate is straight-line code for every
// evaluation and the (rare: ~1e-7 per evaluation) "within eps of a boundary" cases are collected in a flag and
// settled afterwards by ONE rolled loop over the exact predicate.  Same answer as slots_below() in every case.
struct sb_est { int32_t j; bool near; };
template <int kind>
// uslot (stratified only, optional): slot t's 23-bit uniform read from memory (gmx_slot_uniforms wrote
// bits32(key, t) >> 9 there — on a background stream, ahead of the chain) instead of one Threefry block per evaluation.
__device__ __forceinline__ sb_est slots_below_est(gmx_key key, uint32_t u0, uint64_t c, uint64_t total,
                                                  double n_over_total, double eps, int32_t n_out,
                                                  const uint32_t* __restrict__ uslot = nullptr) {
  const double cd = __builtin_fma((double)(uint32_t)(c >> 32), 4294967296.0, (double)(uint32_t)c);   // exact product, one rounding
  sb_est r;
  if (kind == GMX_RESAMPLE_SYSTEMATIC) {
    // slot j is below iff j + du < v  <=>  j < v - du: the count is floor(v - du) + 1 (v - du not an integer;
    // within eps of one, the exact predicate decides).  y = v - du + 1 > 0 comes out of ONE fma.
    const double y = __builtin_fma(cd, n_over_total, 1.0 - (double)u0 * (1.0 / 8388608.0));
    const int32_t t = (int32_t)y;                // floor (y > 0), saturating
    const double frac = y - (double)t;
    r.j = t < n_out ? t : n_out;
    r.near = (frac < eps) || (frac > 1.0 - eps);
    return r;
  }
  const double v = cd * n_over_total;
  int32_t t = (int32_t)v;                        // floor (v >= 0), saturating
  t = t < n_out - 1 ? t : n_out - 1;
  const double frac = v - (double)t;
  const uint32_t u = uslot ? uslot[t] : (gmx_bits32(key, (uint64_t)(uint32_t)t) >> 9);
  const double diff = frac - (double)u * (1.0 / 8388608.0);
  r.j = t + (diff > 0.0 ? 1 : 0);
  r.near = (frac < eps) || (frac > 1.0 - eps) || !(__builtin_fabs(diff) > eps);
  if (c == 0ull) { r.j = 0; r.near = false; }
  if (c >= total) { r.j = n_out; r.near = false; }
  return r;
}

// inclusive max-scan of u32 over the wave (the LDS slot fill of k_offspring_tile / k_shard_step_fill)
__device__ __forceinline__ uint32_t gmx_wave_umax_scan(uint32_t v) {
#define GMX_OP(CTRL) "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " CTRL "\n\t"
  asm(GMX_DPP_ASM_ENTER GMX_DPP_ASM_STEPS(GMX_OP) : "+v"(v));
#undef GMX_OP
  return v;
}
