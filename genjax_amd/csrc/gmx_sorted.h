// gmx_sorted.h — multinomial resampling with SORTED uniforms (GMX_RESAMPLE_MULTINOMIAL_SORTED, include/genmi.h):
// the definition, and the layout of the order-statistics table a resampling reads.  Host + device.
//
// n iid uniforms, sorted, are distributed as normalised partial sums of n + 1 unit exponentials (the gaps of a
// Poisson process): U_(j) = S_j / S_total.  Drawing them that way makes multinomial resampling an ORDERED scheme —
// slot j's ancestor is non-decreasing in j, exactly as for systematic / stratified — so it runs on the same kernel
// (k_offspring_tile: no CDF array, no search, no random cache lines) with another "slots below this CDF value":
//
//   E_j   = 1 + trunc(-log(u_j) * 2^16),  u_j = ((w_j >> 9) + 0.5) * 2^-23,                  j = 0 .. n   (integers)
//           (w_{2i}, w_{2i+1}) = the two words of threefry(key, ctr = i): two exponentials per block
//   S_j   = E_0 + ... + E_j  (j < n),     S_total = S_{n-1} + E_n
//   ancestor(j) = first i with cdf_i * S_total > S_j * total                                  (128-bit integers)
//
// Offspring counts are Multinomial(n, w) (up to the 2^-16 grid of the exponentials); the output is ordered by
// ancestor.  All sums are integers: the table is the same on any partitioning of the work.
//
// The table of one resampling (uint32 words; gmx_sorted_uniforms_words(n)), written by gmx_sorted_uniforms:
//   slow  [tiles * 1024]   low 32 bits of S_j (0 past n); tiles = ceil(n / 1024)
//   guide [NG + 2]         guide[g] = #{ j < n : (S_j >> sh) < g } for g <= (S_total >> sh) + 1;  NG = n + n/2 + 1024
//   tsum  u64 [tiles]      sum of E over tile t (scratch of the two-launch scan)
//   toff  u64 [tiles + 1]  sum of E before tile t; toff[tiles] = S_total
//   sh    [1]              the smallest s with (S_total >> s) <= NG - 2 (16 for any n above a few thousand)
// A CDF value c has  f(c) = #{ j : S_j * total < c * S_total }  slots below it: with t = c * S_total / total,
// f(c) = guide[g] + #{ j in bucket g : (S_j mod 2^sh) <= (floor(t) mod 2^sh) },  g = floor(t) >> sh  (t not an integer).
#pragma once
#include "gmx_math.h"
#include "gmx_rng.h"

#define GMX_SORTED_TILE 1024
#define GMX_SORTED_SCALE 65536.0f

GMX_HD uint32_t gmx_sorted_exp_word(uint32_t w) {
  const float u = ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f;     // (0, 1), exact
  const float e = -gmx_logf(u);                                            // (0, 16.64]
  return 1u + (uint32_t)(e * GMX_SORTED_SCALE);
}
// two exponentials per Threefry block: E_{2i}, E_{2i+1} from the two words of block i
GMX_HD void gmx_sorted_exp_pair(gmx_key key, uint64_t i, uint32_t* e0, uint32_t* e1) {
  uint32_t a, b;
  gmx_threefry2x32(key.k0, key.k1, (uint32_t)(i >> 32), (uint32_t)i, &a, &b);
  *e0 = gmx_sorted_exp_word(a); *e1 = gmx_sorted_exp_word(b);
}
GMX_HD uint32_t gmx_sorted_exp(gmx_key key, uint64_t j) {
  uint32_t e0, e1;
  gmx_sorted_exp_pair(key, j >> 1, &e0, &e1);
  return (j & 1ull) ? e1 : e0;
}

struct gmx_sorted_layout {
  int64_t tiles, ng;
  size_t off_guide, off_tsum, off_toff, off_sh, words;
};
GMX_HD gmx_sorted_layout gmx_sorted_layout_of(int64_t n) {
  gmx_sorted_layout L;
  L.tiles = (n + GMX_SORTED_TILE - 1) / GMX_SORTED_TILE;
  L.ng = n + (n >> 1) + 1024;
  L.off_guide = (size_t)L.tiles * GMX_SORTED_TILE;
  L.off_tsum = L.off_guide + (((size_t)L.ng + 2 + 3) & ~(size_t)3);
  L.off_toff = L.off_tsum + 2 * (size_t)L.tiles;
  L.off_sh = L.off_toff + 2 * ((size_t)L.tiles + 1);
  L.words = (L.off_sh + 2 + 3) & ~(size_t)3;
  return L;
}
GMX_HD uint32_t gmx_sorted_shift(uint64_t stot, int64_t ng) {
  uint32_t s = 0;
  while ((stot >> s) > (uint64_t)(ng - 2)) ++s;
  return s;
}
