// gmx_math2.h — the standard-normal sampler for TWO particles at a time (device only, specialised kernels).
//
// OPT-IN EXPERIMENT (GENMI_JIT_PAIR_NORMALS=1; measured slower on MI355X — see gmx_kernels.hip: jit_pair_normals).
// A bootstrap step is bound by vector-instruction issue, and gfx950 has packed f32 arithmetic: v_pk_mul_f32,
// v_pk_add_f32, v_pk_fma_f32 do two IEEE operations per instruction.  hipcc does not pair the four particles of a
// thread by itself (the sampler has a branch; seeding its SLP vectoriser with a build-vector does not help), so the
// float pipeline of gmx_std_normal_from_bits (gmx_rng.h) — bits -> uniform -> -log1p(-u^2) -> erf_inv polynomial —
// is written here on 2-vectors.  EVERY operation is the scalar function's, in its order, element by element (a packed
// instruction rounds each half exactly as the scalar one does; fma only where the scalar code has an fma; integer
// and compare / select steps stay per element), so the two results carry the bits gmx_std_normal_from_bits gives
// for each input — which is what the parity tests check, specialised kernel against interpreter and oracle.
#pragma once
#include "gmx_rng.h"

typedef float gmx_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gmx_f2 gmx_fma2(gmx_f2 a, gmx_f2 b, gmx_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ gmx_f2 gmx_splat2(float v) { gmx_f2 r; r.x = v; r.y = v; return r; }

// gmx_neg_log1m_sq (gmx_math.h) on two values
__device__ __forceinline__ gmx_f2 gmx_neg_log1m_sq2(gmx_f2 x) {
  const gmx_f2 t = -(x * x);
  const gmx_f2 u = gmx_splat2(1.0f) + t;
  const uint32_t us0 = gmx_f2u(u.x), us1 = gmx_f2u(u.y);
  int e0 = (int)(us0 >> 23) - 126, e1 = (int)(us1 >> 23) - 126;
  gmx_f2 m;
  m.x = gmx_u2f((us0 & 0x007fffffu) | 0x3f000000u);
  m.y = gmx_u2f((us1 & 0x007fffffu) | 0x3f000000u);
  const int low0 = m.x < 0.707106781186547524f, low1 = m.y < 0.707106781186547524f;
  e0 -= low0; e1 -= low1;
  const gmx_f2 mm = m + m;
  gmx_f2 g;                                  // low ? (m + m) - 1 : m - 1
  g.x = low0 ? mm.x : m.x;
  g.y = low1 ? mm.y : m.y;
  const gmx_f2 f = g - gmx_splat2(1.0f);
  const gmx_f2 z = f * f;
  gmx_f2 p = gmx_splat2(7.0376836292e-2f);
  p = gmx_fma2(p, f, gmx_splat2(-1.1514610310e-1f));
  p = gmx_fma2(p, f, gmx_splat2(1.1676998740e-1f));
  p = gmx_fma2(p, f, gmx_splat2(-1.2420140846e-1f));
  p = gmx_fma2(p, f, gmx_splat2(1.4249322787e-1f));
  p = gmx_fma2(p, f, gmx_splat2(-1.6668057665e-1f));
  p = gmx_fma2(p, f, gmx_splat2(2.0000714765e-1f));
  p = gmx_fma2(p, f, gmx_splat2(-2.4999993993e-1f));
  p = gmx_fma2(p, f, gmx_splat2(3.3333331174e-1f));
  gmx_f2 y = (p * f) * z;
  gmx_f2 ef;
  ef.x = (float)e0; ef.y = (float)e1;
  y = gmx_fma2(ef, gmx_splat2(-2.12194440e-4f), y);
  y = gmx_fma2(gmx_splat2(-0.5f), z, y);
  gmx_f2 l = f + y;
  l = gmx_fma2(ef, gmx_splat2(0.693359375f), l);
  l.x = (us0 == 0u) ? -gmx_inf() : l.x;      // log(0)
  l.y = (us1 == 0u) ? -gmx_inf() : l.y;
  const gmx_f2 d = u - gmx_splat2(1.0f);
  gmx_f2 q;                                  // t / d: IEEE division, per element (d == 0 only when u == 1: overridden)
  q.x = t.x / d.x;
  q.y = t.y / d.y;
  gmx_f2 r = l * q;
  r.x = (u.x == 1.0f) ? t.x : r.x;
  r.y = (u.y == 1.0f) ? t.y : r.y;
  return -r;
}

// gmx_erfinv_central / gmx_erfinv_tail (gmx_math.h) on two values
__device__ __forceinline__ gmx_f2 gmx_erfinv_central2(gmx_f2 w) {
  w = w - gmx_splat2(2.5f);
  gmx_f2 p = gmx_splat2(2.81022636e-08f);
  p = gmx_fma2(p, w, gmx_splat2(3.43273939e-07f));
  p = gmx_fma2(p, w, gmx_splat2(-3.5233877e-06f));
  p = gmx_fma2(p, w, gmx_splat2(-4.39150654e-06f));
  p = gmx_fma2(p, w, gmx_splat2(0.00021858087f));
  p = gmx_fma2(p, w, gmx_splat2(-0.00125372503f));
  p = gmx_fma2(p, w, gmx_splat2(-0.00417768164f));
  p = gmx_fma2(p, w, gmx_splat2(0.246640727f));
  p = gmx_fma2(p, w, gmx_splat2(1.50140941f));
  return p;
}
__device__ __forceinline__ gmx_f2 gmx_erfinv_tail2(gmx_f2 w) {
  gmx_f2 s;
  s.x = gmx_sqrtf(w.x); s.y = gmx_sqrtf(w.y);
  w = s - gmx_splat2(3.0f);
  gmx_f2 p = gmx_splat2(-0.000200214257f);
  p = gmx_fma2(p, w, gmx_splat2(0.000100950558f));
  p = gmx_fma2(p, w, gmx_splat2(0.00134934322f));
  p = gmx_fma2(p, w, gmx_splat2(-0.00367342844f));
  p = gmx_fma2(p, w, gmx_splat2(0.00573950773f));
  p = gmx_fma2(p, w, gmx_splat2(-0.0076224613f));
  p = gmx_fma2(p, w, gmx_splat2(0.00943887047f));
  p = gmx_fma2(p, w, gmx_splat2(1.00167406f));
  p = gmx_fma2(p, w, gmx_splat2(2.83297682f));
  return p;
}

// gmx_std_normal_from_bits (gmx_rng.h) for two draws: sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1))
__device__ __forceinline__ gmx_f2 gmx_std_normal_from_bits2(uint32_t b0, uint32_t b1) {
  const float lo = -0.99999994f;             // nextafter(-1, 0) in f32
  // gmx_uniform_from_bits(bits, lo, 1): u = unit(bits); v = u * (hi - lo); v = v + lo; v > lo ? v : lo
  gmx_f2 u;
  u.x = gmx_u2f((b0 >> 9) | 0x3f800000u);
  u.y = gmx_u2f((b1 >> 9) | 0x3f800000u);
  u = u - gmx_splat2(1.0f);
  gmx_f2 v = u * gmx_splat2(1.0f - lo);
  v = v + gmx_splat2(lo);
  gmx_f2 x;
  x.x = v.x > lo ? v.x : lo;
  x.y = v.y > lo ? v.y : lo;
  // gmx_erfinvf_unit(x) = gmx_erfinvf_from_w(x, gmx_neg_log1m_sq(x))
  const gmx_f2 w = gmx_neg_log1m_sq2(x);
  gmx_f2 p = gmx_erfinv_central2(w);
  if (!(w.x < 5.0f) || !(w.y < 5.0f)) {       // w >= 5 (and NaN): the tail polynomial for that element
    const gmx_f2 pt = gmx_erfinv_tail2(w);
    p.x = (w.x < 5.0f) ? p.x : pt.x;
    p.y = (w.y < 5.0f) ? p.y : pt.y;
  }
  gmx_f2 r = p * x;
  r.x = (gmx_fabs(x.x) == 1.0f) ? x.x * gmx_inf() : r.x;
  r.y = (gmx_fabs(x.y) == 1.0f) ? x.y * gmx_inf() : r.y;
  return gmx_splat2(1.41421354f) * r;
}
