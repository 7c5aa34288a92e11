/* orc_sweep.c — CPU ORACLE (test infrastructure): a multi-threaded C statement
 * of the bootstrap particle-filter sweep that genjax_amd.inference.smc.
 * BootstrapSweep runs on the GPU, for the linear-Gaussian model of BASELINE
 * config 2.  Used (a) by tests to cross-check the numpy oracle at sizes where
 * numpy is slow and (b) by bench.py as the `cpu_baseline` ("port": the oracle's
 * algorithm, same Threefry key tree, same inputs, timed on the host cores).
 *
 * Same arithmetic as orc_core.c (included below so the two cannot drift);
 * same build-defined key schedule as the product:
 *   step key = fold_in(run_key, t); (k_prop, k_res, k_mh) = split(step key, 3);
 *   particle i: pk = split(k_prop, N)[i]; site c uses fold_in(pk, c).
 * Model (genjax_amd/workloads.py make_lgssm):
 *   t = 0: x ~ normal(0, s0) @ "x"       t > 0: x ~ normal(a * x_prev, sx) @ "x"
 *          y ~ normal(x, sy) @ "y" constrained to ys[t]
 */
#include "orc_core.c"

#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>

typedef unsigned __int128 u128;

static void derive(const uint32_t* k, uint64_t c, uint32_t* o) {
  threefry(k[0], k[1], (uint32_t)(c >> 32), (uint32_t)c, &o[0], &o[1]);
}
static float normal_logpdf1(float x, float loc, float sc) {
  float a = x / sc, b = loc / sc, d = a - b;
  float un = -0.5f * (d * d);
  float ln = HALF_LOG_2PI + orc_logf(sc);
  return un - ln;
}

int orc_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* One sweep.  Outputs: x[n] (last step's particles), lw[n], anc[n] (last
 * ancestors), maxs[T], totals[T].  work: x2[n], cdf[n].  Returns 0. */
int orc_lgssm_sweep(int64_t n, int64_t T, const float* ys, uint32_t key0, uint32_t key1, float a,
                    float sx, float sy, float s0, int shift, float* x, float* x2, float* lw,
                    uint64_t* cdf, int32_t* anc, float* maxs, uint64_t* totals) {
  const uint32_t run_key[2] = {key0, key1};
  float* cur = x;
  float* prev = x2;
  const int64_t tiles = (n + ORC_CDF_TILE - 1) / ORC_CDF_TILE;
  uint64_t* tile_g = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)tiles);
  if (!tile_g) return 1;
  for (int64_t t = 0; t < T; ++t) {
    uint32_t sk[2], kp[2], kr[2];
    derive(run_key, (uint64_t)t, sk);          /* fold_in(run_key, t) */
    derive(sk, 0, kp);                         /* split(step key, 3)[0] */
    derive(sk, 1, kr);                         /* split(step key, 3)[1] */
    float* tmp = cur; cur = prev; prev = tmp;  /* prev = last step's particles */
    const float yt = ys[t];
    float M = -f_inf();
#pragma omp parallel for schedule(static) reduction(max : M)
    for (int64_t i = 0; i < n; ++i) {
      uint32_t pk[2], k1[2];
      derive(kp, (uint64_t)i, pk);             /* split(k_prop, N)[i] */
      derive(pk, 1, k1);                       /* site "x": fold_in(pk, 1) */
      float z = std_normal_from_bits(bits32_1(k1, 0));
      float loc = (t == 0) ? 0.0f : a * prev[anc[i]];
      float sc = (t == 0) ? s0 : sx;
      float v = z * sc;
      float xv = v + loc;
      cur[i] = xv;
      /* weight = 0.0 + logpdf(y; x, sy)  (generate: only the constrained site) */
      float w = 0.0f + normal_logpdf1(yt, xv, sy);
      lw[i] = w;
      if (w > M) M = w;
    }
    maxs[t] = M;
    /* two-level integer CDF (orc_core.c::orc_weight_cdf_tiled, tiles in parallel) */
    float scale = pow2i(shift);
    const int32_t K = orc_tile_exp(M);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < tiles; ++b) {
      int64_t lo = b * ORC_CDF_TILE, hi = lo + ORC_CDF_TILE < n ? lo + ORC_CDF_TILE : n;
      float m = -f_inf();
      for (int64_t i = lo; i < hi; ++i) m = fmax_nanskip(m, lw[i]);
      const int32_t k = orc_tile_exp(m);
      const float ref = orc_tile_ref(k);
      uint64_t run = 0;
      for (int64_t i = lo; i < hi; ++i) {
        run += weight_fixed1(lw[i], ref, scale);
        cdf[i] = tile_scale(run, k, K);
      }
      tile_g[b] = tile_scale(run, k, K);
    }
    uint64_t total = 0;
    for (int64_t b = 0; b < tiles; ++b) { uint64_t g = tile_g[b]; tile_g[b] = total; total += g; }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < tiles; ++b) {
      int64_t lo = b * ORC_CDF_TILE, hi = lo + ORC_CDF_TILE < n ? lo + ORC_CDF_TILE : n;
      for (int64_t i = lo; i < hi; ++i) cdf[i] += tile_g[b];
    }
    totals[t] = total;
    /* systematic resampling: first i with cdf_i * (n*2^23) > (j*2^23 + u0) * total */
    uint64_t u0 = bits32_1(kr, 0) >> 9;
    u128 D = (u128)((uint64_t)n << 23);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
      u128 P = (u128)(((uint64_t)j << 23) + u0) * total;
      int64_t lo = 0, hi = n;
      while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((u128)cdf[mid] * D > P) hi = mid; else lo = mid + 1;
      }
      anc[j] = (int32_t)(lo >= n ? n - 1 : lo);
    }
  }
  if (cur != x) memcpy(x, cur, sizeof(float) * (size_t)n);
  free(tile_g);
  return 0;
}
