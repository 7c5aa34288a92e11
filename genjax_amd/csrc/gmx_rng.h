// gmx_rng.h — Threefry-2x32 (20 rounds) and the jax.random key algebra the
// reference relies on (jax 0.5.2, `jax_threefry_partitionable=True`, the
// default for that pin).  Everything lives in
// registers: a key is two uint32, nothing here touches memory.
//
// Reference call sites this replaces:
//   jax.random.split   — src/genjax/_src/inference/smc.py:154,171,299-300,386
//   jax.random.fold_in — src/genjax/_src/generative_functions/static.py:261,350,420,525,634
//   sampler bits       — distributions/tensorflow_probability/__init__.py:52-55
#pragma once
#include "gmx_math.h"

struct gmx_key { uint32_t k0, k1; };

GMX_HD uint32_t gmx_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define GMX_TF_ROUND(r)            \
  x0 += x1;                        \
  x1 = gmx_rotl(x1, r);            \
  x1 ^= x0;

// Threefry-2x32-20.  Known-answer vectors (Random123) are checked in
// tests/test_oracle_pins.py::test_threefry_random123_kat and tests/test_abi.py through the C-ABI symbol
// gmx_threefry2x32_host.
GMX_HD void gmx_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                              uint32_t* o0, uint32_t* o1) {
  uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
  GMX_TF_ROUND(13) GMX_TF_ROUND(15) GMX_TF_ROUND(26) GMX_TF_ROUND(6)
  x0 += ks1; x1 += ks2 + 1u;
  GMX_TF_ROUND(17) GMX_TF_ROUND(29) GMX_TF_ROUND(16) GMX_TF_ROUND(24)
  x0 += ks2; x1 += ks0 + 2u;
  GMX_TF_ROUND(13) GMX_TF_ROUND(15) GMX_TF_ROUND(26) GMX_TF_ROUND(6)
  x0 += ks0; x1 += ks1 + 3u;
  GMX_TF_ROUND(17) GMX_TF_ROUND(29) GMX_TF_ROUND(16) GMX_TF_ROUND(24)
  x0 += ks1; x1 += ks2 + 4u;
  GMX_TF_ROUND(13) GMX_TF_ROUND(15) GMX_TF_ROUND(26) GMX_TF_ROUND(6)
  x0 += ks2; x1 += ks0 + 5u;
  *o0 = x0; *o1 = x1;
}

// split(key, n)[i] in partitionable mode == threefry(key, ctr = (hi32(i), lo32(i))).
GMX_HD gmx_key gmx_split_child(gmx_key k, uint64_t i) {
  gmx_key o;
  gmx_threefry2x32(k.k0, k.k1, (uint32_t)(i >> 32), (uint32_t)i, &o.k0, &o.k1);
  return o;
}
// fold_in(key, d) == threefry(key, ctr = (0, d)): the same block as split child d.
GMX_HD gmx_key gmx_fold_in(gmx_key k, uint32_t d) {
  gmx_key o;
  gmx_threefry2x32(k.k0, k.k1, 0u, d, &o.k0, &o.k1);
  return o;
}
// random_bits(key, 32, shape)[j] == hi ^ lo of threefry(key, ctr = j).
GMX_HD uint32_t gmx_bits32(gmx_key k, uint64_t j) {
  uint32_t a, b;
  gmx_threefry2x32(k.k0, k.k1, (uint32_t)(j >> 32), (uint32_t)j, &a, &b);
  return a ^ b;
}

// jax.random.uniform's mantissa trick: [0, 1) with 23 random bits.
GMX_HD float gmx_bits_to_unit(uint32_t bits) {
  return gmx_u2f((bits >> 9) | 0x3f800000u) - 1.0f;
}
// uniform(key, minval, maxval): max(minval, u * (maxval - minval) + minval),
// mul and add rounded separately (no contraction).
GMX_HD float gmx_uniform_from_bits(uint32_t bits, float lo, float hi) {
  float u = gmx_bits_to_unit(bits);
  float v = u * (hi - lo);
  v = v + lo;
  return v > lo ? v : lo;
}
// jax.random.normal: sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1)).
GMX_HD float gmx_std_normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f;   // nextafter(-1, 0) in f32
  float u = gmx_uniform_from_bits(bits, lo, 1.0f);
  return 1.41421354f * gmx_erfinvf_unit(u);     // |u| < 1 by construction
}
// jax.random.gumbel: -log(-log(uniform(tiny, 1))).
GMX_HD float gmx_gumbel_from_bits(uint32_t bits) {
  const float tiny = 1.17549435e-38f;
  float u = gmx_uniform_from_bits(bits, tiny, 1.0f);
  return -gmx_logf(-gmx_logf(u));
}
