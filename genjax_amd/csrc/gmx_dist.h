// gmx_dist.h — per-particle samplers and log-densities for the distributions on
// the north-star path, restating TFP 0.23's JAX substrate formulas
// (SURVEY.md App. A.3).  Reference wrappers being replaced:
//   src/genjax/_src/generative_functions/distributions/tensorflow_probability/__init__.py
//     normal :259, uniform :294, flip :155, bernoulli :72, beta :82, categorical :102
// through `tfp_distribution.sampler / logpdf` (:52-62).
//
// Samplers take the SITE key and an element counter `e`: a scalar site uses
// e = 0; element j of a vector-valued site uses e = j (one site key, element j
// takes counter j — SURVEY.md App. A.3 last bullet).
#pragma once
#include "gmx_rng.h"

#define GMX_HALF_LOG_2PI 0.918938533204672741780329736406f

// ---- Normal(loc, scale) ----------------------------------------------------
// sample = jax.random.normal(key) * scale + loc       (mul, then add)
GMX_HD float gmx_normal_sample(gmx_key k, uint32_t e, float loc, float scale) {
  float z = gmx_std_normal_from_bits(gmx_bits32(k, e));
  float v = z * scale;
  return v + loc;
}
// log_prob = -0.5 * (x/scale - loc/scale)^2 - (0.5*log(2*pi) + log(scale))
GMX_HD float gmx_normal_logpdf(float x, float loc, float scale) {
  float a = x / scale;
  float b = loc / scale;
  float d = a - b;
  float un = -0.5f * (d * d);
  float ln = GMX_HALF_LOG_2PI + gmx_logf(scale);
  return un - ln;
}

// ---- Uniform(low, high) ----------------------------------------------------
// sample = low + (high - low) * uniform[0,1)
GMX_HD float gmx_uniform_sample(gmx_key k, uint32_t e, float lo, float hi) {
  float u = gmx_bits_to_unit(gmx_bits32(k, e));
  float r = hi - lo;
  float v = r * u;
  return lo + v;
}
GMX_HD float gmx_uniform_logpdf(float x, float lo, float hi) {
  if (gmx_isnan(x)) return x;
  if (x < lo || x > hi) return -gmx_inf();
  return -gmx_logf(hi - lo);
}

// ---- Bernoulli(probs=p, dtype=bool) == genjax.flip -------------------------
// sample = uniform[0,1) < p
GMX_HD int gmx_flip_sample(gmx_key k, uint32_t e, float p) {
  float u = gmx_bits_to_unit(gmx_bits32(k, e));
  return u < p ? 1 : 0;
}
// log_prob(e) = mul_no_nan(log1p(-p), 1-e) + mul_no_nan(log(p), e)
GMX_HD float gmx_flip_logpdf(int ev, float p) {
  float t0 = ev ? 0.0f : gmx_log1pf(-p);
  float t1 = ev ? gmx_logf(p) : 0.0f;
  return t0 + t1;
}

// ---- Bernoulli(logits=s) == genjax.bernoulli -------------------------------
// probs = sigmoid(s); sample = uniform < probs
GMX_HD int gmx_bernoulli_logits_sample(gmx_key k, uint32_t e, float s) {
  float u = gmx_bits_to_unit(gmx_bits32(k, e));
  return u < gmx_sigmoidf(s) ? 1 : 0;
}
// log p1 = -softplus(-s), log p0 = -softplus(s)
GMX_HD float gmx_bernoulli_logits_logpdf(int ev, float s) {
  return ev ? -gmx_softplusf(-s) : -gmx_softplusf(s);
}

// ---- Gamma / Beta ----------------------------------------------------------
// TFP draws Beta(c1, c0) as sigmoid(log g1 - log g0) with two log-space
// Marsaglia–Tsang gamma draws from seeds split with salt 'beta'.  TFP's exact
// stream cannot be restated from the reference tree (un-vendored dependency):
// PARITY UNPINNED for the sample stream — distributional parity only
// (tests check moments / KS), see DESIGN.md.  The stream defined here:
//   key_j  = split_child(site_key', j), j = 0,1   (site_key' = fold_in(site_key, e))
//   attempt t of a draw uses bits32(key_j, 2t) for the normal and
//   bits32(key_j, 2t+1) for the acceptance uniform; the boost uniform for
//   concentration < 1 uses counter 0xFFFFFFFF.
GMX_HD float gmx_log_gamma_sample(gmx_key k, float alpha) {
  float boost = 0.0f;
  float a = alpha;
  if (alpha < 1.0f) {
    // G(a) = G(a+1) * U^(1/a)  ->  log G(a) = log G(a+1) + log(U)/a
    float u = gmx_bits_to_unit(gmx_bits32(k, 0xFFFFFFFFull));
    u = 1.0f - u;                      // (0, 1]
    boost = gmx_logf(u) / alpha;
    a = alpha + 1.0f;
  }
  float d = a - 0.333333343f;
  float c = 1.0f / gmx_sqrtf(9.0f * d);
  float res = gmx_logf(d);             // fallback if the attempt cap is hit
  for (uint32_t t = 0; t < 64u; ++t) {
    float z = gmx_std_normal_from_bits(gmx_bits32(k, 2ull * t));
    float v = gmx_fma(c, z, 1.0f);
    if (v <= 0.0f) continue;
    float v3 = v * v * v;
    float u = gmx_bits_to_unit(gmx_bits32(k, 2ull * t + 1ull));
    u = 1.0f - u;                      // (0, 1]
    float lv3 = gmx_logf(v3);
    float z2 = z * z;
    float rhs = 0.5f * z2 + d * ((1.0f - v3) + lv3);
    if (gmx_logf(u) < rhs) {
      res = gmx_logf(d) + lv3;
      break;
    }
  }
  return res + boost;
}
GMX_HD float gmx_beta_sample(gmx_key k, uint32_t e, float c1, float c0) {
  gmx_key ke = gmx_fold_in(k, e);
  float lg1 = gmx_log_gamma_sample(gmx_split_child(ke, 0), c1);
  float lg0 = gmx_log_gamma_sample(gmx_split_child(ke, 1), c0);
  return gmx_sigmoidf(lg1 - lg0);
}
// log_prob = xlogy(c1-1, x) + xlog1py(c0-1, -x) - lbeta(c1, c0)
GMX_HD float gmx_beta_logpdf(float x, float c1, float c0) {
  float t = gmx_xlogyf(c1 - 1.0f, x) + gmx_xlog1pyf(c0 - 1.0f, -x);
  float lb = (gmx_lgammaf(c1) + gmx_lgammaf(c0)) - gmx_lgammaf(c1 + c0);
  return t - lb;
}

// ---- Categorical (Gumbel-max, streaming) -----------------------------------
// sample = argmax_k(logits[k] + gumbel(key, counter k)), lowest index wins ties.
// Used as a running reduction: feed categories in increasing k.
struct gmx_cat_state { float best; int idx; };
GMX_HD void gmx_cat_step(gmx_cat_state* s, gmx_key k, uint64_t ctr, int cat, float logit) {
  float g = gmx_gumbel_from_bits(gmx_bits32(k, ctr));
  float v = logit + g;
  // first max wins: strict >, but a first finite/NaN-free candidate always enters
  if (cat == 0 || v > s->best) { s->best = v; s->idx = cat; }
}
