// gmx_math.h — f32 elementary functions with a FIXED operation sequence.
//
// Every function here is a straight-line sequence of IEEE-754 binary32
// add / mul / fma / div / sqrt / rint plus integer bit manipulation.  All of
// those are correctly rounded on gfx950 (hipcc keeps IEEE div/sqrt by default)
// and on x86-64, so the same input gives the same bits on the GPU and in the
// CPU oracle's independent restatement (oracle/orc_core.c) as long as both are
// built with -ffp-contract=off (fusion happens only where fmaf is written).
//
// Why not ocml/libm: resampling indices must be bit-exact between MI355X and
// the CPU oracle (BASELINE.json north_star), and the fixed-point CDF is built
// from exp(lw - max); vendor expf/logf differ in the last ulp between targets.
//
// Accuracy targets (checked against float64 in tests/test_oracle_pins.py::test_elementary_function_accuracy
// for the oracle's side and, bit for bit against it, in tests/test_gpu_parity.py for this one):
//   expf, logf, log1pf  <= 2 ulp;  lgammaf <= 4e-6 relative (x >= 1e-3);
//   erfinvf = the XLA/Giles f32 polynomial (reference: SURVEY.md App. A.2).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#else
#include "genmi.h"
#endif

#if defined(__HIPCC__)
#define GMX_HD __host__ __device__ __forceinline__
#define GMX_HDM __host__ __device__ __forceinline__   /* member functions */
#else
#define GMX_HD static inline
#define GMX_HDM inline
#endif

#define GMX_INF_BITS 0x7f800000u
#define GMX_NAN_BITS 0x7fc00000u

GMX_HD float gmx_u2f(uint32_t u) {
  union { uint32_t u; float f; } c; c.u = u; return c.f;
}
GMX_HD uint32_t gmx_f2u(float f) {
  union { uint32_t u; float f; } c; c.f = f; return c.u;
}
GMX_HD float gmx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
GMX_HD float gmx_inf() { return gmx_u2f(GMX_INF_BITS); }
GMX_HD float gmx_nan() { return gmx_u2f(GMX_NAN_BITS); }
GMX_HD int gmx_isnan(float x) { return (gmx_f2u(x) & 0x7fffffffu) > GMX_INF_BITS; }
GMX_HD float gmx_fabs(float x) { return gmx_u2f(gmx_f2u(x) & 0x7fffffffu); }
GMX_HD float gmx_fmax(float a, float b) { return (a > b || gmx_isnan(b)) ? a : b; }
GMX_HD float gmx_fmin(float a, float b) { return (a < b || gmx_isnan(b)) ? a : b; }
// the max of REDUCTIONS (block / tile / global maxima of log-weights): a NaN operand is ignored, max(-0, +0) = +0 —
// IEEE maxNum as v_max_f32 implements it, so the result does not depend on the order of the reduction (gmx_fmax
// above returns its SECOND operand for -0 vs +0).  One instruction on the device.
#if defined(__HIP_DEVICE_COMPILE__)
GMX_HD float gmx_rmax(float a, float b) { return __builtin_fmaxf(a, b); }
#else
GMX_HD float gmx_rmax(float a, float b) {
  if (gmx_isnan(a)) return b;
  if (gmx_isnan(b)) return a;
  if (a == b) return gmx_u2f(gmx_f2u(a) & gmx_f2u(b));      // +-0: + wins; otherwise the same bits
  return a > b ? a : b;
}
#endif

// 2^k as a float for k in [-126, 127].
GMX_HD float gmx_pow2i(int k) { return gmx_u2f((uint32_t)(k + 127) << 23); }

// Block floating point for the two-level integer CDF (include/genmi.h "Resampling"): a tile whose
// largest log-weight is m gets the exponent k = ceil(m / ln 2) (clamped to +-2^29; -inf / NaN -> -2^29)
// and its weights are taken relative to k * ln 2, so tiles combine by integer shifts 2^(k - K).
#define GMX_TILE_EXP_LIM (1 << 29)
GMX_HD int32_t gmx_tile_exp(float m) {               // straight-line: ceil(clamp(m / ln 2)), NaN -> -LIM
  const float t = m * gmx_u2f(0x3FB8AA3Bu);        // 1 / ln 2
  float tc = t > -(float)GMX_TILE_EXP_LIM ? t : -(float)GMX_TILE_EXP_LIM;
  tc = tc < (float)GMX_TILE_EXP_LIM ? tc : (float)GMX_TILE_EXP_LIM;
  return (int32_t)__builtin_ceilf(tc);
}
GMX_HD float gmx_tile_ref(int32_t k) { return (float)k * gmx_u2f(0x3F317218u); }   // k * ln 2
// v * 2^(k - K) for k <= K and v < 2^63 (every tile sum is: shift + log2 n <= 62): a shift by 63 or more leaves 0
GMX_HD uint64_t gmx_tile_scale(uint64_t v, int32_t k, int32_t K) {
  const int32_t d = K - k;                         // |k|, |K| <= 2^29: no overflow
  return v >> (d < 63 ? d : 63);
}

// exp(x).  Results below the smallest normal are flushed to +0 so that the
// answer never depends on a target's denormal mode.
// Straight-line: the range checks are selects applied to the main path's result (computed on a
// clamped argument), so on the GPU the function is one basic block — no exec-mask branches in the
// middle of a particle's instruction stream.  The selected values are the ones the branching form
// returned, bit for bit.
GMX_HD float gmx_expf(float x) {
  const float hi = 88.72283935546875f, lo = -87.33654022216797f;
  const int nan = gmx_isnan(x);
  float xc = x > hi ? hi : x;
  xc = xc < lo ? lo : xc;
  xc = nan ? 0.0f : xc;
  float kf = __builtin_rintf(xc * 1.44269502162933349609375f);
  // Cody–Waite: ln2 = hi + lo, hi has 9 trailing zero bits (kf*hi is exact).
  float r = gmx_fma(kf, -0.693359375f, xc);
  r = gmx_fma(kf, 2.12194440e-4f, r);
  // e^r on [-ln2/2, ln2/2], degree-6 minimax (cephes expf coefficients).
  float p = 1.9875691500e-4f;
  p = gmx_fma(p, r, 1.3981999507e-3f);
  p = gmx_fma(p, r, 8.3334519073e-3f);
  p = gmx_fma(p, r, 4.1665795894e-2f);
  p = gmx_fma(p, r, 1.6666665459e-1f);
  p = gmx_fma(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  p = gmx_fma(p, r2, r);
  p = p + 1.0f;
  int k = (int)kf;
  int k1 = k >> 1;          // split so both scale factors stay normal
  int k2 = k - k1;
  float y = (p * gmx_pow2i(k1)) * gmx_pow2i(k2);
  y = y < 1.17549435e-38f ? 0.0f : y;
  y = x < lo ? 0.0f : y;
  y = x > hi ? gmx_inf() : y;
  return nan ? x : y;
}

// exp(x) for x <= 0 (or a hair above: anything <= 88) and NaN: gmx_expf without the overflow clamp and without the
// explicit NaN select (a NaN argument reaches the result through the polynomial on its own).  Same bits as
// gmx_expf on that domain; used for the CDF weights exp(lw - ref), ref >= the tile's largest log-weight.
GMX_HD float gmx_expf_nonpos(float x) {
  const float lo = -87.33654022216797f;
  float xc = x < lo ? lo : x;                       // NaN stays NaN
  float kf = __builtin_rintf(xc * 1.44269502162933349609375f);
  float r = gmx_fma(kf, -0.693359375f, xc);
  r = gmx_fma(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = gmx_fma(p, r, 1.3981999507e-3f);
  p = gmx_fma(p, r, 8.3334519073e-3f);
  p = gmx_fma(p, r, 4.1665795894e-2f);
  p = gmx_fma(p, r, 1.6666665459e-1f);
  p = gmx_fma(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  p = gmx_fma(p, r2, r);
  p = p + 1.0f;
  int k = gmx_isnan(x) ? 0 : (int)kf;               // (int)NaN is not portable: pin it
  int k1 = k >> 1;
  int k2 = k - k1;
  float y = (p * gmx_pow2i(k1)) * gmx_pow2i(k2);
  y = y < 1.17549435e-38f ? 0.0f : y;               // false for NaN
  return x < lo ? 0.0f : y;
}

// floor(gmx_expf(d) * 2^shift) as u64, 0 when that is NaN, negative or not below 2^63 — the fixed-point weight of the
// integer CDF (include/genmi.h "Resampling": l_i), 1 <= shift <= 62 — without going through a float result:
// gmx_expf's value is p * 2^k with p = its polynomial (in [0.70, 1.42]) and the power-of-two scalings exact, so the
// floor is p's 24-bit significand shifted by (exponent of p) - 23 + k + shift.  Same integer as the expression above
// for EVERY float d and every shift (tests/test_host_logic.py::test_weight_fixed_matches_its_definition sweeps the
// bit patterns), in about half the instructions: no 2^k1 * 2^k2 scaling, no denormal flush, no f32 -> u64 conversion.
// packed: significand (24 bits, 0 when the value is not below 2^63) | shift amount (6 bits) << 24; the value is
// (significand << 39) >> amount.  gmx_exp_fixed_packed / gmx_fixed_unpack split gmx_exp_fixed so that a kernel can hand
// the weight to another in 4 bytes (gmx_run_args.tile_q_d).
GMX_HD uint32_t gmx_exp_fixed_packed(float d, int shift) {
  const float lo = -87.33654022216797f;
  float dc = d > lo ? d : lo;                        // NaN -> lo: contributes 0 below, like every d < -(shift + 1) ln 2
  dc = dc < 100.0f ? dc : 100.0f;                    // anything above 63 ln 2 is "not below 2^63"
  const float kf = __builtin_rintf(dc * 1.44269502162933349609375f);
  float r = gmx_fma(kf, -0.693359375f, dc);
  r = gmx_fma(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = gmx_fma(p, r, 1.3981999507e-3f);
  p = gmx_fma(p, r, 8.3334519073e-3f);
  p = gmx_fma(p, r, 4.1665795894e-2f);
  p = gmx_fma(p, r, 1.6666665459e-1f);
  p = gmx_fma(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  p = gmx_fma(p, r2, r);
  p = p + 1.0f;
  const uint32_t pb = gmx_f2u(p);
  uint32_t m = (pb & 0x007fffffu) | 0x00800000u;     // p = m * 2^(e - 150), e = pb >> 23 (126 or 127)
  // value = m * 2^s, s = e - 150 + k + shift; as (m << 39) >> (39 - s): zero for s <= -24, and s >= 40 is >= 2^63
  int32_t amt = 189 - (int32_t)(pb >> 23) - (int32_t)kf - shift;
  m = amt < 0 ? 0u : m;
  amt = amt > 63 ? 63 : amt;
  amt = amt < 0 ? 0 : amt;
  return m | ((uint32_t)amt << 24);
}
GMX_HD uint64_t gmx_fixed_unpack(uint32_t pk) { return ((uint64_t)(pk & 0x00ffffffu) << 39) >> (pk >> 24); }
GMX_HD uint64_t gmx_exp_fixed(float d, int shift) { return gmx_fixed_unpack(gmx_exp_fixed_packed(d, shift)); }

// log(x), natural.  cephes logf polynomial on [sqrt(1/2), sqrt(2)).
// Straight-line like gmx_expf: special cases are selects over the main path's result.
GMX_HD float gmx_logf(float x) {
  const uint32_t ux = gmx_f2u(x);
  const int den = ux < 0x00800000u;          // denormal (or +0: overridden below): scale up by 2^23 (exact)
  const float xs = den ? x * 8388608.0f : x;
  const uint32_t us = gmx_f2u(xs);
  // frexp: x = m * 2^e with m in [0.5, 1)
  int e = (den ? -23 : 0) + (int)(us >> 23) - 126;
  const float m = gmx_u2f((us & 0x007fffffu) | 0x3f000000u);
  const int low = m < 0.707106781186547524f;
  e -= low;
  const float f = low ? (m + m) - 1.0f : m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = gmx_fma(p, f, -1.1514610310e-1f);
  p = gmx_fma(p, f, 1.1676998740e-1f);
  p = gmx_fma(p, f, -1.2420140846e-1f);
  p = gmx_fma(p, f, 1.4249322787e-1f);
  p = gmx_fma(p, f, -1.6668057665e-1f);
  p = gmx_fma(p, f, 2.0000714765e-1f);
  p = gmx_fma(p, f, -2.4999993993e-1f);
  p = gmx_fma(p, f, 3.3333331174e-1f);
  float y = (p * f) * z;
  float ef = (float)e;
  y = gmx_fma(ef, -2.12194440e-4f, y);
  y = gmx_fma(-0.5f, z, y);
  float r = f + y;
  r = gmx_fma(ef, 0.693359375f, r);
  // special cases, lowest priority first
  r = (ux == GMX_INF_BITS) ? x : r;
  r = (ux >> 31) ? gmx_nan() : r;
  r = (ux == 0u || ux == 0x80000000u) ? -gmx_inf() : r;
  return gmx_isnan(x) ? x : r;
}

// log(1 + x) by Kahan's correction of log(u), u = fl(1 + x).  Straight-line (see gmx_expf).
GMX_HD float gmx_log1pf(float x) {
  float u = 1.0f + x;
  float l = gmx_logf(u);
  float d = u - 1.0f;
  float r = l * (x / d);                 // d == 0 only when u == 1: overridden
  r = (gmx_f2u(u) == GMX_INF_BITS) ? u : r;
  r = (u == 1.0f) ? x : r;
  return gmx_isnan(x) ? x : r;
}

GMX_HD float gmx_sqrtf(float x) { return __builtin_sqrtf(x); }

// softplus(x) = log(1 + e^x), the stable form max(x,0) + log1p(exp(-|x|)).
GMX_HD float gmx_softplusf(float x) {
  float ax = gmx_fabs(x);
  float t = gmx_log1pf(gmx_expf(-ax));
  return (x > 0.0f ? x : 0.0f) + t;
}

GMX_HD float gmx_sigmoidf(float x) {
  // 1 / (1 + e^-x), evaluated on the side that cannot overflow.
  if (x >= 0.0f) {
    float e = gmx_expf(-x);
    return 1.0f / (1.0f + e);
  }
  float e = gmx_expf(x);
  return e / (1.0f + e);
}

GMX_HD float gmx_tanhf(float x) {
  // tanh(x) = sign(x) * (1 - e^{-2|x|}) / (1 + e^{-2|x|})
  float ax = gmx_fabs(x);
  if (ax < 1e-4f) return x;
  float e = gmx_expf(-2.0f * ax);
  float t = (1.0f - e) / (1.0f + e);
  return (gmx_f2u(x) >> 31) ? -t : t;
}

// -log1p(-x*x) for |x| <= 1, i.e. gmx_log1pf(-(x*x)) negated, with only the cases that argument can reach:
// u = 1 - x*x lies in [0, 1] (never NaN / negative / infinite, and never denormal: it is 0 or >= 2^-24), so of
// gmx_logf's and gmx_log1pf's special handling only "u == 1" (return the argument) and "u == 0" (log = -inf) are
// live.  Same operation sequence on the main path, hence the same bits as the general functions for every such x
// (tests: the oracle keeps the general form) — 14 vector instructions fewer per normal draw.
GMX_HD float gmx_neg_log1m_sq(float x) {
  const float t = -(x * x);
  const float u = 1.0f + t;
  const uint32_t us = gmx_f2u(u);
  int e = (int)(us >> 23) - 126;
  const float m = gmx_u2f((us & 0x007fffffu) | 0x3f000000u);
  const int low = m < 0.707106781186547524f;
  e -= low;
  const float f = low ? (m + m) - 1.0f : m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = gmx_fma(p, f, -1.1514610310e-1f);
  p = gmx_fma(p, f, 1.1676998740e-1f);
  p = gmx_fma(p, f, -1.2420140846e-1f);
  p = gmx_fma(p, f, 1.4249322787e-1f);
  p = gmx_fma(p, f, -1.6668057665e-1f);
  p = gmx_fma(p, f, 2.0000714765e-1f);
  p = gmx_fma(p, f, -2.4999993993e-1f);
  p = gmx_fma(p, f, 3.3333331174e-1f);
  float y = (p * f) * z;
  float ef = (float)e;
  y = gmx_fma(ef, -2.12194440e-4f, y);
  y = gmx_fma(-0.5f, z, y);
  float l = f + y;
  l = gmx_fma(ef, 0.693359375f, l);
  l = (us == 0u) ? -gmx_inf() : l;          // log(0)
  const float d = u - 1.0f;
  float r = l * (t / d);                     // d == 0 only when u == 1: overridden
  r = (u == 1.0f) ? t : r;
  return -r;
}

// Inverse error function, XLA's f32 lowering of Giles' polynomial
// (SURVEY.md App. A.2), given w = -log1p(-x*x).  |x| == 1 -> +-inf.
GMX_HD float gmx_erfinv_central(float w) {      // w < 5
  w = w - 2.5f;
  float p = 2.81022636e-08f;
  p = gmx_fma(p, w, 3.43273939e-07f);
  p = gmx_fma(p, w, -3.5233877e-06f);
  p = gmx_fma(p, w, -4.39150654e-06f);
  p = gmx_fma(p, w, 0.00021858087f);
  p = gmx_fma(p, w, -0.00125372503f);
  p = gmx_fma(p, w, -0.00417768164f);
  p = gmx_fma(p, w, 0.246640727f);
  p = gmx_fma(p, w, 1.50140941f);
  return p;
}
GMX_HD float gmx_erfinv_tail(float w) {         // w >= 5 (and NaN)
  w = gmx_sqrtf(w) - 3.0f;
  float p = -0.000200214257f;
  p = gmx_fma(p, w, 0.000100950558f);
  p = gmx_fma(p, w, 0.00134934322f);
  p = gmx_fma(p, w, -0.00367342844f);
  p = gmx_fma(p, w, 0.00573950773f);
  p = gmx_fma(p, w, -0.0076224613f);
  p = gmx_fma(p, w, 0.00943887047f);
  p = gmx_fma(p, w, 1.00167406f);
  p = gmx_fma(p, w, 2.83297682f);
  return p;
}
GMX_HD float gmx_erfinvf_from_w(float x, float w) {
  float p;
  if (w < 5.0f) p = gmx_erfinv_central(w);
  else p = gmx_erfinv_tail(w);
  if (gmx_fabs(x) == 1.0f) return x * gmx_inf();
  return p * x;
}
GMX_HD float gmx_erfinvf(float x) { return gmx_erfinvf_from_w(x, -gmx_log1pf(-(x * x))); }
// the same for |x| <= 1 (what a uniform draw feeds it): identical bits, fewer instructions
GMX_HD float gmx_erfinvf_unit(float x) { return gmx_erfinvf_from_w(x, gmx_neg_log1m_sq(x)); }

// log Gamma(x) for x > 0: shift x up to >= 8 with the recurrence, then the
// Stirling series.  The product of shifts is accumulated in one log.
GMX_HD float gmx_lgammaf(float x) {
  if (gmx_isnan(x)) return x;
  if (x <= 0.0f) return gmx_inf();
  if (gmx_f2u(x) == GMX_INF_BITS) return x;
  float shift = 0.0f;
  if (x < 8.0f) {
    float prod = 1.0f;
    // at most 8 steps; the count depends only on x
    while (x < 8.0f) {
      prod = prod * x;
      x = x + 1.0f;
    }
    shift = gmx_logf(prod);
  }
  float inv = 1.0f / x;
  float inv2 = inv * inv;
  // 1/12 - 1/360 t + 1/1260 t^2 - 1/1680 t^3, t = 1/x^2
  float s = -5.9523809523809529e-4f;
  s = gmx_fma(s, inv2, 7.9365079365079365e-4f);
  s = gmx_fma(s, inv2, -2.7777777777777778e-3f);
  s = gmx_fma(s, inv2, 8.3333333333333329e-2f);
  float lx = gmx_logf(x);
  float r = (x - 0.5f) * lx;
  r = r - x;
  r = r + 0.91893853320467274f;
  r = gmx_fma(s, inv, r);
  return r - shift;
}

// x * log(y) with 0 * log(0) = 0 (tf.math.xlogy).
GMX_HD float gmx_xlogyf(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return x * gmx_logf(y);
}
// x * log1p(y) with 0 * anything = 0 (tf.math.xlog1py).
GMX_HD float gmx_xlog1pyf(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return x * gmx_log1pf(y);
}

// sin/cos with a 3-term Cody–Waite reduction by pi/2; accurate to ~2 ulp for
// |x| <= 8192 (the range model code uses; beyond it accuracy degrades
// gracefully, results stay deterministic).
GMX_HD void gmx_sincosf(float x, float* s_out, float* c_out) {
  float q = __builtin_rintf(x * 0.636619746685028076171875f);
  float r = gmx_fma(q, -1.5703125f, x);
  r = gmx_fma(q, -4.837512969970703125e-4f, r);
  r = gmx_fma(q, -7.54978995489188e-8f, r);
  float z = r * r;
  // sin(r), cos(r) on [-pi/4, pi/4] (cephes sinf/cosf coefficients)
  float ps = -1.9515295891e-4f;
  ps = gmx_fma(ps, z, 8.3321608736e-3f);
  ps = gmx_fma(ps, z, -1.6666654611e-1f);
  float sr = gmx_fma(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = gmx_fma(pc, z, -1.388731625493765e-3f);
  pc = gmx_fma(pc, z, 4.166664568298827e-2f);
  float cr = gmx_fma(pc * z, z, gmx_fma(-0.5f, z, 1.0f));
  int n = ((int)q) & 3;
  float s, c;
  if (n == 0) { s = sr; c = cr; }
  else if (n == 1) { s = cr; c = -sr; }
  else if (n == 2) { s = -sr; c = -cr; }
  else { s = -cr; c = sr; }
  *s_out = s; *c_out = c;
}
GMX_HD float gmx_sinf(float x) { float s, c; gmx_sincosf(x, &s, &c); return s; }
GMX_HD float gmx_cosf(float x) { float s, c; gmx_sincosf(x, &s, &c); return c; }

// pow(x, y) for x > 0 via exp(y log x); exact cases handled first.
GMX_HD float gmx_powf(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (y == 1.0f) return x;
  if (y == 2.0f) return x * x;
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : gmx_inf();
  if (x < 0.0f) {
    float yi = __builtin_rintf(y);
    if (yi != y) return gmx_nan();
    float r = gmx_expf(y * gmx_logf(-x));
    int odd = ((int)yi) & 1;
    return odd ? -r : r;
  }
  return gmx_expf(y * gmx_logf(x));
}
