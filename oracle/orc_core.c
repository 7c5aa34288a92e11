/* orc_core.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic the GenJAX hot path delegates to
 * jax 0.5.2 / tensorflow-probability 0.23.0 (un-vendored third-party
 * dependencies: poetry.lock:1627-1629, 1660-1662, 5015-5017 — absent from
 * /root/reference and not importable here, so their published algorithms are
 * restated).  Array-at-a-time entry points, called from
 * oracle/genjax_oracle.py through ctypes.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * Pinning status:
 *   - Threefry-2x32-20: PINNED by the Random123 known-answer vectors
 *     (tests/test_oracle_pins.py).
 *   - normal log-density: PINNED by the one numeric literal in the reference's
 *     tests (tests/generative_functions/test_static_gen_fn.py:317-318, -2.837877).
 *   - jax.random key algebra and the normal / uniform pipelines: PINNED by the
 *     outputs jax's own documentation prints (split(PRNGKey(0)) key data,
 *     normal / uniform of PRNGKey(0), and for jax >= 0.5's partitionable
 *     threefry: normal(key(42)), split(key(42)), the "individually" / "all at
 *     once" vectors) — tests/test_oracle_pins.py::test_jax_docs_*; constants
 *     recalled by the builder (no network), every printed digit matches.
 *   - Gumbel / categorical, Bernoulli, Beta-via-gamma streams and resampling
 *     indices: PARITY UNPINNED against the real jax/TFP (no golden output
 *     exists in the reference and it cannot be run here); restated from the
 *     published algorithms, anchored on the reference's call sites and on
 *     closed-form answers.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Every
 * float function is a fixed sequence of correctly rounded IEEE-754 binary32
 * operations, so results are reproducible bit-for-bit on any conforming target.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- bit helpers -------------------------------------------------------- */
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static int is_nan(float x) { return (f2u(x) & 0x7fffffffu) > 0x7f800000u; }
static float f_inf(void) { return u2f(0x7f800000u); }
static float f_nan(void) { return u2f(0x7fc00000u); }
static float pow2i(int k) { return u2f((uint32_t)(k + 127) << 23); }
static float f_abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }

/* ---- elementary functions (fixed op sequences, explicit fmaf) ----------- */
float orc_expf(float x) {
  if (is_nan(x)) return x;
  if (x > 88.72283935546875f) return f_inf();
  if (x < -87.33654022216797f) return 0.0f;
  float kf = rintf(x * 1.44269502162933349609375f);
  float r = fmaf(kf, -0.693359375f, x);
  r = fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  int k = (int)kf;
  int k1 = k >> 1;
  int k2 = k - k1;
  float y = (p * pow2i(k1)) * pow2i(k2);
  if (y < 1.17549435e-38f) return 0.0f;
  return y;
}

float orc_logf(float x) {
  uint32_t ux = f2u(x);
  if (is_nan(x)) return x;
  if (ux == 0u || ux == 0x80000000u) return -f_inf();
  if (ux >> 31) return f_nan();
  if (ux == 0x7f800000u) return x;
  int e = 0;
  if (ux < 0x00800000u) {
    x = x * 8388608.0f;
    ux = f2u(x);
    e = -23;
  }
  e += (int)(ux >> 23) - 126;
  float m = u2f((ux & 0x007fffffu) | 0x3f000000u);
  float f;
  if (m < 0.707106781186547524f) {
    e -= 1;
    f = (m + m) - 1.0f;
  } else {
    f = m - 1.0f;
  }
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = fmaf(p, f, -1.1514610310e-1f);
  p = fmaf(p, f, 1.1676998740e-1f);
  p = fmaf(p, f, -1.2420140846e-1f);
  p = fmaf(p, f, 1.4249322787e-1f);
  p = fmaf(p, f, -1.6668057665e-1f);
  p = fmaf(p, f, 2.0000714765e-1f);
  p = fmaf(p, f, -2.4999993993e-1f);
  p = fmaf(p, f, 3.3333331174e-1f);
  float y = (p * f) * z;
  float ef = (float)e;
  y = fmaf(ef, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = f + y;
  r = fmaf(ef, 0.693359375f, r);
  return r;
}

float orc_log1pf(float x) {
  if (is_nan(x)) return x;
  float u = 1.0f + x;
  if (u == 1.0f) return x;
  if (f2u(u) == 0x7f800000u) return u;
  float l = orc_logf(u);
  float d = u - 1.0f;
  return l * (x / d);
}

float orc_softplusf(float x) {
  float ax = f_abs(x);
  float t = orc_log1pf(orc_expf(-ax));
  return (x > 0.0f ? x : 0.0f) + t;
}

float orc_sigmoidf(float x) {
  if (x >= 0.0f) {
    float e = orc_expf(-x);
    return 1.0f / (1.0f + e);
  }
  float e = orc_expf(x);
  return e / (1.0f + e);
}

float orc_tanhf(float x) {
  float ax = f_abs(x);
  if (ax < 1e-4f) return x;
  float e = orc_expf(-2.0f * ax);
  float t = (1.0f - e) / (1.0f + e);
  return (f2u(x) >> 31) ? -t : t;
}

/* XLA's f32 erf_inv (Giles), SURVEY.md App. A.2 */
float orc_erfinvf(float x) {
  float w = -orc_log1pf(-(x * x));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  if (f_abs(x) == 1.0f) return x * f_inf();
  return p * x;
}

float orc_lgammaf(float x) {
  if (is_nan(x)) return x;
  if (x <= 0.0f) return f_inf();
  if (f2u(x) == 0x7f800000u) return x;
  float shift = 0.0f;
  if (x < 8.0f) {
    float prod = 1.0f;
    while (x < 8.0f) {
      prod = prod * x;
      x = x + 1.0f;
    }
    shift = orc_logf(prod);
  }
  float inv = 1.0f / x;
  float inv2 = inv * inv;
  float s = -5.9523809523809529e-4f;
  s = fmaf(s, inv2, 7.9365079365079365e-4f);
  s = fmaf(s, inv2, -2.7777777777777778e-3f);
  s = fmaf(s, inv2, 8.3333333333333329e-2f);
  float lx = orc_logf(x);
  float r = (x - 0.5f) * lx;
  r = r - x;
  r = r + 0.91893853320467274f;
  r = fmaf(s, inv, r);
  return r - shift;
}

static void sincosf_(float x, float* so, float* co) {
  float q = rintf(x * 0.636619746685028076171875f);
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188e-8f, r);
  float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float sr = fmaf(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  int n = ((int)q) & 3;
  float s, c;
  if (n == 0) { s = sr; c = cr; }
  else if (n == 1) { s = cr; c = -sr; }
  else if (n == 2) { s = -sr; c = -cr; }
  else { s = -cr; c = sr; }
  *so = s; *co = c;
}
float orc_sinf(float x) { float s, c; sincosf_(x, &s, &c); return s; }
float orc_cosf(float x) { float s, c; sincosf_(x, &s, &c); return c; }

float orc_powf(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (y == 1.0f) return x;
  if (y == 2.0f) return x * x;
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : f_inf();
  if (x < 0.0f) {
    float yi = rintf(y);
    if (yi != y) return f_nan();
    float r = orc_expf(y * orc_logf(-x));
    return (((int)yi) & 1) ? -r : r;
  }
  return orc_expf(y * orc_logf(x));
}

/* vectorised unary / binary dispatch for the numpy layer */
enum { U_EXP = 0, U_LOG, U_LOG1P, U_SQRT, U_SIN, U_COS, U_TANH, U_SIGMOID, U_SOFTPLUS,
       U_LGAMMA, U_ERFINV, U_RECIP };
void orc_unary(int which, int64_t n, const float* x, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float v = x[i], r;
    switch (which) {
      case U_EXP: r = orc_expf(v); break;
      case U_LOG: r = orc_logf(v); break;
      case U_LOG1P: r = orc_log1pf(v); break;
      case U_SQRT: r = sqrtf(v); break;
      case U_SIN: r = orc_sinf(v); break;
      case U_COS: r = orc_cosf(v); break;
      case U_TANH: r = orc_tanhf(v); break;
      case U_SIGMOID: r = orc_sigmoidf(v); break;
      case U_SOFTPLUS: r = orc_softplusf(v); break;
      case U_LGAMMA: r = orc_lgammaf(v); break;
      case U_ERFINV: r = orc_erfinvf(v); break;
      case U_RECIP: r = 1.0f / v; break;
      default: r = f_nan();
    }
    out[i] = r;
  }
}
void orc_pow(int64_t n, const float* x, const float* y, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_powf(x[i], y[i]);
}

/* ---- Threefry-2x32-20 ------------------------------ */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static void threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0, uint32_t* o1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[g & 1][j]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  *o0 = x0; *o1 = x1;
}
void orc_threefry2x32(int64_t n, const uint32_t* k0, const uint32_t* k1, const uint32_t* c0,
                      const uint32_t* c1, uint32_t* o0, uint32_t* o1) {
  for (int64_t i = 0; i < n; ++i) threefry(k0[i], k1[i], c0[i], c1[i], &o0[i], &o1[i]);
}

/* jax.random, threefry_partitionable=True (jax >= 0.5.0 default; App. A.2):
 * split(key,n)[i] = fold_in(key,i) = block(key, ctr=(hi(i), lo(i))) both words;
 * random_bits(key,32,shape)[j] = hi ^ lo of block(key, ctr=j). */
/* keys: [n,2]; ctr: [n] (u64) ; out: [n,2] */
void orc_derive(int64_t n, const uint32_t* keys, int64_t key_stride, const uint64_t* ctr,
                int64_t ctr_stride, uint32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* k = keys + 2 * i * key_stride;
    uint64_t c = ctr[i * ctr_stride];
    threefry(k[0], k[1], (uint32_t)(c >> 32), (uint32_t)c, &out[2 * i], &out[2 * i + 1]);
  }
}
void orc_bits32(int64_t n, const uint32_t* keys, int64_t key_stride, const uint64_t* ctr,
                int64_t ctr_stride, uint32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* k = keys + 2 * i * key_stride;
    uint64_t c = ctr[i * ctr_stride];
    uint32_t a, b;
    threefry(k[0], k[1], (uint32_t)(c >> 32), (uint32_t)c, &a, &b);
    out[i] = a ^ b;
  }
}
static uint32_t bits32_1(const uint32_t* k, uint64_t c) {
  uint32_t a, b;
  threefry(k[0], k[1], (uint32_t)(c >> 32), (uint32_t)c, &a, &b);
  return a ^ b;
}

/* jax.random.uniform / normal / gumbel from raw bits (App. A.2) */
static float unit_from_bits(uint32_t bits) { return u2f((bits >> 9) | 0x3f800000u) - 1.0f; }
static float uniform_from_bits(uint32_t bits, float lo, float hi) {
  float u = unit_from_bits(bits);
  float v = u * (hi - lo);
  v = v + lo;
  return v > lo ? v : lo;
}
static float std_normal_from_bits(uint32_t bits) {
  float u = uniform_from_bits(bits, -0.99999994f, 1.0f);
  return 1.41421354f * orc_erfinvf(u);
}
static float gumbel_from_bits(uint32_t bits) {
  float u = uniform_from_bits(bits, 1.17549435e-38f, 1.0f);
  return -orc_logf(-orc_logf(u));
}
void orc_unit_from_bits(int64_t n, const uint32_t* bits, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = unit_from_bits(bits[i]);
}
void orc_std_normal_from_bits(int64_t n, const uint32_t* bits, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = std_normal_from_bits(bits[i]);
}
void orc_gumbel_from_bits(int64_t n, const uint32_t* bits, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = gumbel_from_bits(bits[i]);
}

/* ---- TFP 0.23 distributions (App. A.3).  Strides are 0 (broadcast) or 1. -- */
#define HALF_LOG_2PI 0.918938533204672741780329736406f

/* Normal: sample = normal(key) * scale + loc */
void orc_normal_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* loc,
                       int64_t ls, const float* scale, int64_t ss, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float z = std_normal_from_bits(bits32_1(keys + 2 * i * ks, elem));
    float v = z * scale[i * ss];
    out[i] = v + loc[i * ls];
  }
}
/* log_prob = -0.5*(x/scale - loc/scale)^2 - (0.5 log 2pi + log scale) */
void orc_normal_logpdf(int64_t n, const float* x, int64_t xs, const float* loc, int64_t ls,
                       const float* scale, int64_t ss, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float sc = scale[i * ss];
    float a = x[i * xs] / sc;
    float b = loc[i * ls] / sc;
    float d = a - b;
    float un = -0.5f * (d * d);
    float ln = HALF_LOG_2PI + orc_logf(sc);
    out[i] = un - ln;
  }
}
/* Uniform(low, high): low + (high-low)*u */
void orc_uniform_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* lo,
                        int64_t los, const float* hi, int64_t his, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float u = unit_from_bits(bits32_1(keys + 2 * i * ks, elem));
    float r = hi[i * his] - lo[i * los];
    float v = r * u;
    out[i] = lo[i * los] + v;
  }
}
void orc_uniform_logpdf(int64_t n, const float* x, int64_t xs, const float* lo, int64_t los,
                        const float* hi, int64_t his, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float v = x[i * xs], l = lo[i * los], h = hi[i * his];
    if (is_nan(v)) out[i] = v;
    else if (v < l || v > h) out[i] = -f_inf();
    else out[i] = -orc_logf(h - l);
  }
}
/* flip(p) = Bernoulli(probs=p, dtype=bool): u < p */
void orc_flip_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* p,
                     int64_t ps, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    float u = unit_from_bits(bits32_1(keys + 2 * i * ks, elem));
    out[i] = u < p[i * ps] ? 1 : 0;
  }
}
void orc_flip_logpdf(int64_t n, const int32_t* ev, int64_t es, const float* p, int64_t ps, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    int e = ev[i * es] != 0;
    float pp = p[i * ps];
    float t0 = e ? 0.0f : orc_log1pf(-pp);
    float t1 = e ? orc_logf(pp) : 0.0f;
    out[i] = t0 + t1;
  }
}
/* bernoulli(logits) */
void orc_bernl_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* s,
                      int64_t ss, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    float u = unit_from_bits(bits32_1(keys + 2 * i * ks, elem));
    out[i] = u < orc_sigmoidf(s[i * ss]) ? 1 : 0;
  }
}
void orc_bernl_logpdf(int64_t n, const int32_t* ev, int64_t es, const float* s, int64_t ss, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float l = s[i * ss];
    out[i] = (ev[i * es] != 0) ? -orc_softplusf(-l) : -orc_softplusf(l);
  }
}

/* Beta(c1, c0) = sigmoid(log g1 - log g0); the gamma stream is the BUILD's
 * definition (TFP's rejection sampler and 'beta' salt cannot be restated from
 * the reference tree: PARITY UNPINNED, distributional checks only). */
static float log_gamma_draw(const uint32_t* k, float alpha) {
  float boost = 0.0f, a = alpha;
  if (alpha < 1.0f) {
    float u = unit_from_bits(bits32_1(k, 0xFFFFFFFFull));
    u = 1.0f - u;
    boost = orc_logf(u) / alpha;
    a = alpha + 1.0f;
  }
  float d = a - 0.333333343f;
  float c = 1.0f / sqrtf(9.0f * d);
  float res = orc_logf(d);
  for (uint32_t t = 0; t < 64u; ++t) {
    float z = std_normal_from_bits(bits32_1(k, 2ull * t));
    float v = fmaf(c, z, 1.0f);
    if (v <= 0.0f) continue;
    float v3 = v * v * v;
    float u = unit_from_bits(bits32_1(k, 2ull * t + 1ull));
    u = 1.0f - u;
    float lv3 = orc_logf(v3);
    float z2 = z * z;
    float rhs = 0.5f * z2 + d * ((1.0f - v3) + lv3);
    if (orc_logf(u) < rhs) {
      res = orc_logf(d) + lv3;
      break;
    }
  }
  return res + boost;
}
/* log of a Gamma(alpha, 1) draw from key split(key)[elem] (the Dirichlet element stream; build-defined) */
void orc_loggamma_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* alpha,
                         int64_t as, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* k = keys + 2 * i * ks;
    uint32_t ke[2];
    threefry(k[0], k[1], 0u, (uint32_t)elem, &ke[0], &ke[1]);
    out[i] = log_gamma_draw(ke, alpha[i * as]);
  }
}
void orc_beta_sample(int64_t n, const uint32_t* keys, int64_t ks, uint64_t elem, const float* c1,
                     int64_t s1, const float* c0, int64_t s0, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* k = keys + 2 * i * ks;
    uint32_t ke[2], ka[2], kb[2];
    threefry(k[0], k[1], 0u, (uint32_t)elem, &ke[0], &ke[1]);
    threefry(ke[0], ke[1], 0u, 0u, &ka[0], &ka[1]);
    threefry(ke[0], ke[1], 0u, 1u, &kb[0], &kb[1]);
    float lg1 = log_gamma_draw(ka, c1[i * s1]);
    float lg0 = log_gamma_draw(kb, c0[i * s0]);
    out[i] = orc_sigmoidf(lg1 - lg0);
  }
}
/* log_prob = xlogy(c1-1, x) + xlog1py(c0-1, -x) - lbeta(c1, c0) */
void orc_beta_logpdf(int64_t n, const float* x, int64_t xs, const float* c1, int64_t s1,
                     const float* c0, int64_t s0, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float v = x[i * xs], a = c1[i * s1], b = c0[i * s0];
    float am = a - 1.0f, bm = b - 1.0f;
    float t1 = (am == 0.0f) ? 0.0f : am * orc_logf(v);
    float t0 = (bm == 0.0f) ? 0.0f : bm * orc_log1pf(-v);
    float lb = (orc_lgammaf(a) + orc_lgammaf(b)) - orc_lgammaf(a + b);
    out[i] = (t1 + t0) - lb;
  }
}
/* Categorical(logits[n,K]) single draw per row: argmax_k(l[k] + gumbel(ctr k)),
 * first max wins.  ctr_mul/ctr_add let callers express batched layouts
 * (row i, cat k) -> ctr = ctr_add[i] + k (App. A.3 last bullet). */
void orc_categorical_sample(int64_t n, int64_t K, const uint32_t* keys, int64_t ks,
                            const float* logits, int64_t row_stride, const uint64_t* ctr_add,
                            int64_t cs, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* k = keys + 2 * i * ks;
    const float* l = logits + i * row_stride;
    uint64_t base = ctr_add ? ctr_add[i * cs] : 0ull;
    float best = 0.0f;
    int32_t bi = 0;
    for (int64_t c = 0; c < K; ++c) {
      float g = gumbel_from_bits(bits32_1(k, base + (uint64_t)c));
      float v = l[c] + g;
      if (c == 0 || v > best) { best = v; bi = (int32_t)c; }
    }
    out[i] = bi;
  }
}

/* ---- fixed-point weights (build-defined) ----------------- */
/* q_i = floor(exp(lw_i - ref) * 2^shift); NaN / negative / >= 2^63 (lw = +inf) -> 0 */
static uint64_t weight_fixed1(float lw, float ref, float scale) {
  float v = orc_expf(lw - ref) * scale;
  return (v >= 0.0f && v < 0x1p63f) ? (uint64_t)v : 0ull;
}
void orc_weight_fixed(int64_t n, const float* lw, float ref, int shift, uint64_t* q) {
  float scale = pow2i(shift);
  for (int64_t i = 0; i < n; ++i) q[i] = weight_fixed1(lw[i], ref, scale);
}
static float fmax_nanskip(float a, float b) { return (a > b || b != b) ? a : b; }
/* Block floating point: the exponent of a tile whose largest log-weight is m is k = ceil(m / ln 2), clamped
 * to +-2^29 (-inf / NaN -> -2^29); the tile's weights are taken relative to k * ln 2. */
#define ORC_TILE_EXP_LIM (1 << 29)
int32_t orc_tile_exp(float m) {
  const float t = m * u2f(0x3FB8AA3Bu);           /* 1 / ln 2 */
  if (!(t > -(float)ORC_TILE_EXP_LIM)) return -ORC_TILE_EXP_LIM;
  if (t > (float)ORC_TILE_EXP_LIM) return ORC_TILE_EXP_LIM;
  int32_t k = (int32_t)t;
  if ((float)k < t) ++k;
  return k;
}
float orc_tile_ref(int32_t k) { return (float)k * u2f(0x3F317218u); }
static uint64_t tile_scale(uint64_t v, int32_t k, int32_t K) {
  int64_t d = (int64_t)K - (int64_t)k;
  return d < 64 ? v >> d : 0ull;
}
/* Two-level integer CDF.  Tiles of ORC_CDF_TILE consecutive GLOBAL indices; per tile b:
 *   m_b = max lw, k_b = tile_exp(m_b), l_i = floor(exp(lw_i - k_b ln 2) * 2^shift), L_i = inclusive
 *   tile-local sum, A_b = L_last
 * then with M = max_b m_b (or the caller's, when the array is one shard of a larger population), K = tile_exp(M):
 *   G_b = A_b >> (K - k_b),   cdf_i = sum_{b' < b} G_b' + (L_i >> (K - k_b))
 * Integers throughout after the per-particle exp: nothing depends on how the work is split. */
#define ORC_CDF_TILE 1024
void orc_weight_cdf_tiled(int64_t n, const float* lw, int shift, int use_M, float M_in, uint64_t* cdf,
                          float* M_out, uint64_t* total_out) {
  int64_t tiles = (n + ORC_CDF_TILE - 1) / ORC_CDF_TILE;
  float scale = pow2i(shift);
  float M = -f_inf();
  for (int64_t i = 0; i < n; ++i) M = fmax_nanskip(M, lw[i]);
  if (use_M) M = M_in;
  const int32_t K = orc_tile_exp(M);
  uint64_t prefix = 0;
  for (int64_t b = 0; b < tiles; ++b) {
    int64_t lo = b * ORC_CDF_TILE, hi = lo + ORC_CDF_TILE < n ? lo + ORC_CDF_TILE : n;
    float m = -f_inf();
    for (int64_t i = lo; i < hi; ++i) m = fmax_nanskip(m, lw[i]);
    const int32_t k = orc_tile_exp(m);
    const float ref = orc_tile_ref(k);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) {
      run += weight_fixed1(lw[i], ref, scale);
      cdf[i] = prefix + tile_scale(run, k, K);
    }
    prefix += tile_scale(run, k, K);
  }
  *M_out = M;
  *total_out = prefix;
}
/* Exact integer inverse-CDF for large n (include/genmi.h, gmx_ancestors) with 128-bit integers:
 *   systematic / stratified: first i with cdf_i * (n_out*2^23) > (j*2^23 + u_j) * total
 *   multinomial:             first i with cdf_i * 2^23 >= total * (2^23 - u_j)
 * One binary search per output slot.  genjax_oracle.ancestors() states the same predicate with
 * Python integers; tests check the two against each other. */
void orc_ancestors(int kind, const uint32_t* key, const uint64_t* cdf, int64_t n_in, int64_t n_out,
                   int32_t* out) {
  typedef unsigned __int128 u128;
  uint64_t total = cdf[n_in - 1];
  uint64_t u0 = bits32_1(key, 0) >> 9;
  for (int64_t j = 0; j < n_out; ++j) {
    if (total == 0) { out[j] = (int32_t)(n_in - 1); continue; }
    u128 D, P;
    int strict;
    if (kind == 2) {
      uint64_t u = bits32_1(key, (uint64_t)j) >> 9;
      D = (u128)1 << 23; P = (u128)total * (u128)(((uint64_t)1 << 23) - u); strict = 0;
    } else {
      uint64_t u = (kind == 0) ? u0 : (uint64_t)(bits32_1(key, (uint64_t)j) >> 9);
      D = (u128)((uint64_t)n_out << 23); P = (u128)(((uint64_t)j << 23) + u) * (u128)total; strict = 1;
    }
    int64_t lo = 0, hi = n_in;
    while (lo < hi) {
      int64_t mid = lo + ((hi - lo) >> 1);
      u128 c = (u128)cdf[mid] * D;
      int hit = strict ? (c > P) : (c >= P);
      if (hit) hi = mid; else lo = mid + 1;
    }
    out[j] = (int32_t)(lo < n_in ? lo : n_in - 1);
  }
}

/* Multinomial resampling with SORTED uniforms (include/genmi.h, GMX_RESAMPLE_MULTINOMIAL_SORTED; csrc/gmx_sorted.h):
 *   E_j = 1 + trunc(-log(u_j) * 2^16), u_j = ((w_j >> 9) + 0.5) * 2^-23, j = 0 .. n, (w_2i, w_2i+1) = the two words
 *   of threefry(key, ctr = i);
 *   S_j = E_0 + .. + E_j, S_total = S_{n-1} + E_n;  ancestor(j) = first i with cdf_i * S_total > S_j * total. */
static uint32_t sorted_e(const uint32_t* key, uint64_t j) {
  uint32_t a, b;
  threefry(key[0], key[1], (uint32_t)((j >> 1) >> 32), (uint32_t)(j >> 1), &a, &b);
  float u = ((float)(((j & 1) ? b : a) >> 9) + 0.5f) * 1.1920928955078125e-07f;   /* (0, 1), exact */
  float e = -orc_logf(u);
  return 1u + (uint32_t)(e * 65536.0f);
}
void orc_sorted_exp(int64_t count, const uint32_t* key, uint32_t* out) {
  for (int64_t j = 0; j < count; ++j) out[j] = sorted_e(key, (uint64_t)j);
}
void orc_ancestors_sorted(const uint32_t* key, const uint64_t* cdf, int64_t n, int32_t* out) {
  typedef unsigned __int128 u128;
  uint64_t total = cdf[n - 1];
  if (total == 0) { for (int64_t j = 0; j < n; ++j) out[j] = (int32_t)(n - 1); return; }
  uint64_t stot = 0;
  for (int64_t j = 0; j <= n; ++j) stot += sorted_e(key, (uint64_t)j);
  uint64_t s = 0;
  int64_t i = 0;
  for (int64_t j = 0; j < n; ++j) {
    s += sorted_e(key, (uint64_t)j);
    u128 P = (u128)s * (u128)total;
    while (i < n - 1 && !((u128)cdf[i] * (u128)stot > P)) ++i;     /* thresholds increase with j: a merge */
    out[j] = (int32_t)i;
  }
}

/* MH accept: log(uniform(key)) < log_alpha */
void orc_mh_accept(int64_t n, const uint32_t* keys, int64_t ks, const float* log_alpha, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    float u = unit_from_bits(bits32_1(keys + 2 * i * ks, 0));
    float r = 1.0f - 0.0f;
    float v = r * u;
    v = 0.0f + v;
    out[i] = orc_logf(v) < log_alpha[i] ? 1 : 0;
  }
}
