// gmx_resample.h — device code of the ordered (systematic / stratified) resamplers, shared by the AOT kernels
// (gmx_kernels.hip: k_offspring, k_offspring_tile, k_shard_*) and by the hiprtc-specialised site programs
// (gmx_jit.h: the resampling prologue of a bootstrap step, gmx_run_args.rs).  Device only.
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


// ---------------------------------------------------------------------------------------------------------------
// gmx_rs_window: the ancestors of ONE workgroup's 1024 output slots, computed by that workgroup itself from the
// PREVIOUS step's log-weights and tile statistics — the resampling step folded into the prologue of the next site
// program, so that a bootstrap step is ONE launch (the launch boundary + ramp of a second kernel, ≈ 3.6 us on MI355X,
// is a third of a two-launch step).  Same integers as k_offspring_tile, by construction:
//
//   A  every workgroup turns the <= 2048 tile statistics (m_b, A_b) into M, K, G_b = A_b >> (K - k_b), the exclusive
//      tile prefixes P_b and the total (thread t owns tiles [t per, (t+1) per): one u64 wave scan + 4 wave totals),
//      and every tile's first slot S_b = f(P_b);  P_b, S_b go to LDS.
//   B  the slots [W, W + 1024) of this workgroup belong to sources in tiles b_lo .. b_hi (b_lo = last tile with
//      S_b <= W, b_hi = last tile with S_b < W + 1024: two ballot counts; on average two tiles, sum over all
//      workgroups <= 2 x tiles).  For each of them that owns a slot: rebuild its local CDF from the log-weights in
//      registers (4 exp per thread, u64 DPP scan — k_offspring_tile's code), slot edges e[0..4] from one f64 fma
//      each (exact 128-bit predicate within eps of a boundary), and PUSH: every source with a slot inside the window
//      writes its index at the (clipped) first of them into s_mark[slot - W].
//   C  a source's remaining slots follow its first one and source indices increase with the slot, so a MAX-SCAN of
//      s_mark (4 consecutive slots per thread, DPP, 4 wave carries) fills every slot; the result is stored
//      (anc_out_d, 16 bytes per thread) and read back transposed as the thread's own rows p * 256 + t.
//
// LDS (dynamic, gmx_rs_window_lds(n) bytes): s_mark[1024] | scratch | P[tiles] | S[tiles + 1].
// Barriers: every thread of the workgroup reaches every __syncthreads (trip counts are workgroup-uniform).
// ---------------------------------------------------------------------------------------------------------------
#define GMX_RS_TILE 1024
#define GMX_RS_MAX_TILES 2048
#define GMX_RS_PER_MAX (GMX_RS_MAX_TILES / GMX_BLOCK)

GMX_HD size_t gmx_rs_window_lds(int64_t n) {
  const size_t tiles = (size_t)((n + GMX_RS_TILE - 1) / GMX_RS_TILE);
  return 4096 + 256 + 8 * tiles + 4 * (tiles + 1) + 16;
}

// inclusive max-scan of u32 over the wave
__device__ __forceinline__ uint32_t gmx_wave_umax_scan(uint32_t v) {
#define GMX_OP(CTRL) "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " CTRL "\n\t"
  asm(GMX_DPP_ASM_ENTER GMX_DPP_ASM_STEPS(GMX_OP) : "+v"(v));
#undef GMX_OP
  return v;
}

template <int kind>
__device__ __forceinline__ void gmx_rs_window(const gmx_resample_in& Q, const int64_t n, uint32_t* arow, char* smem) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t n32 = (int32_t)n;
  const int n_tiles = (int)((n + GMX_RS_TILE - 1) / GMX_RS_TILE);
  const int per = (n_tiles + GMX_BLOCK - 1) / GMX_BLOCK;                 // 1 .. 8, uniform
  uint32_t* s_mark = reinterpret_cast<uint32_t*>(smem);                   // [1024]
  uint64_t* s_w64 = reinterpret_cast<uint64_t*>(smem + 4096);             // [4] phase A, [4 + 8 parity + wave] phase B
  float* s_wf = reinterpret_cast<float*>(smem + 4096 + 128);              // [4]
  uint32_t* s_wi = reinterpret_cast<uint32_t*>(smem + 4096 + 160);        // [4] counts, [4 + wave] carries
  uint64_t* s_P = reinterpret_cast<uint64_t*>(smem + 4096 + 256);         // [n_tiles]
  int32_t* s_S = reinterpret_cast<int32_t*>(s_P + n_tiles);               // [n_tiles + 1]
  const int32_t W = (int32_t)blockIdx.x * GMX_RS_TILE;                    // first slot of the window (< n)
  const int32_t Wend = (W + GMX_RS_TILE < n32) ? W + GMX_RS_TILE : n32;
  const float scale = gmx_pow2i(Q.shift);
  gmx_key key; key.k0 = Q.key0; key.k1 = Q.key1;
  const uint32_t u0 = Q.u0;

  // ---- A: tile statistics -> prefixes, total, first slots ----
  uint64_t ta[GMX_RS_PER_MAX];
  float tm[GMX_RS_PER_MAX];
#pragma unroll
  for (int r = 0; r < GMX_RS_PER_MAX; ++r) {          // loads first (clamped rows)
    ta[r] = 0ull; tm[r] = -gmx_inf();
    if (r < per) {
      const int t = tid * per + r;
      const int tc = t < n_tiles ? t : n_tiles - 1;
      ta[r] = Q.tile_agg_d[tc];
      tm[r] = Q.tile_max_d[tc];
    }
  }
  reinterpret_cast<uint4*>(s_mark)[tid] = make_uint4(0u, 0u, 0u, 0u);
  float M = -gmx_inf();
#pragma unroll
  for (int r = 0; r < GMX_RS_PER_MAX; ++r) {
    if (r < per) {
      const bool ok = tid * per + r < n_tiles;
      ta[r] = ok ? ta[r] : 0ull;
      tm[r] = ok ? tm[r] : -gmx_inf();
      M = gmx_rmax(M, tm[r]);
    }
  }
  M = wave_max(M);
  if (lane == 0) s_wf[wave] = M;
  __syncthreads();
  M = gmx_rmax(gmx_rmax(s_wf[0], s_wf[1]), gmx_rmax(s_wf[2], s_wf[3]));
  const int32_t K = gmx_tile_exp(M);
  uint64_t P[GMX_RS_PER_MAX];
  uint64_t run = 0;
#pragma unroll
  for (int r = 0; r < GMX_RS_PER_MAX; ++r) {
    P[r] = run;
    if (r < per) run += gmx_tile_scale(ta[r], gmx_tile_exp(tm[r]), K);
  }
  const uint64_t inc = wave_scan_u64(run);
  if (lane == 63) s_w64[wave] = inc;
  __syncthreads();
  uint64_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const uint64_t v = s_w64[w]; total += v; wave_off += (w < wave) ? v : 0ull; }
  if (blockIdx.x == 0 && tid == 0) {
    if (Q.total_out_d) *Q.total_out_d = total;
    if (Q.max_out_d) *Q.max_out_d = M;
  }
  if (total == 0) {       // no mass at all (all weights -inf / NaN): every slot maps to the last particle
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      arow[p] = (uint32_t)(n32 - 1);
      const int32_t j = W + p * GMX_BLOCK + tid;
      if (j < n32) Q.anc_out_d[j] = n32 - 1;
    }
    return;               // workgroup-uniform
  }
  const uint64_t base = wave_off + (inc - run);
  const double n_over_total = (double)n / (double)total;
  const double eps = (double)n * 0x1p-44 + 0x1p-40;
  const uint64_t D = (uint64_t)n << 23;
  uint32_t c_lo = 0, c_hi = 0;         // wave-uniform counts (SALU)
#pragma unroll
  for (int r = 0; r < GMX_RS_PER_MAX; ++r) {
    if (r < per) {
      const int t = tid * per + r;
      const bool ok = t < n_tiles;
      P[r] += base;
      const sb_est q = slots_below_est<kind>(key, u0, P[r], total, n_over_total, eps, n32);
      int32_t S = q.j;
      if (__any(q.near && ok)) {       // cold
        if (q.near && ok) S = (int32_t)slots_below_exact(kind, key, (uint64_t)u0, P[r], D, total, (int64_t)S, n);
      }
      if (ok) { s_P[t] = P[r]; s_S[t] = S; }
      c_lo += (uint32_t)__popcll(__ballot(ok && S <= W));
      c_hi += (uint32_t)__popcll(__ballot(ok && S < Wend));
    }
  }
  if (lane == 0) s_wi[wave] = c_lo | (c_hi << 16);
  if (tid == 0) s_S[n_tiles] = n32;
  __syncthreads();
  uint32_t cnt = (s_wi[0] + s_wi[1]) + (s_wi[2] + s_wi[3]);
  cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
  const int b_lo = (int)(cnt & 0xffffu) - 1, b_hi = (int)(cnt >> 16) - 1;   // S_0 = 0 <= W < Wend: both >= 0

  // ---- B: push the sources of tiles b_lo .. b_hi into the window ----
  int it = 0;
  for (int b = b_lo; b <= b_hi; ++b) {
    const int32_t Sb = __builtin_amdgcn_readfirstlane(s_S[b]), Sb1 = __builtin_amdgcn_readfirstlane(s_S[b + 1]);
    if (Sb1 == Sb) continue;                          // a tile without offspring (uniform)
    const uint64_t prefix = s_P[b];
    const float tmax_b = Q.tile_max_d[b];
    const int64_t i0 = (int64_t)b * GMX_RS_TILE + (int64_t)tid * 4;
    float x[4];
    if ((int64_t)(b + 1) * GMX_RS_TILE <= n) {
      const float4 v = *reinterpret_cast<const float4*>(Q.lw_d + i0);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t ic = i0 + c < n ? i0 + c : n - 1;
        const float xv = Q.lw_d[ic];
        x[c] = (i0 + c < n) ? xv : -gmx_inf();
      }
    }
    const int32_t k_b = gmx_tile_exp(tmax_b);
    const float ref_b = gmx_tile_ref(k_b);
    uint64_t q[4];
    uint64_t rn = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint64_t w = weight_fixed(x[c], ref_b, scale);
      rn += (i0 + c < n) ? w : 0ull;
      q[c] = rn;
    }
    const uint64_t sc = wave_scan_u64(rn);
    uint64_t* s_sc = s_w64 + 4 + 4 * (it & 1);
    if (lane == 63) s_sc[wave] = sc;
    __syncthreads();
    uint64_t woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += (w < wave) ? s_sc[w] : 0ull;
    const uint64_t loc = woff + (sc - rn);
    uint64_t cv[5];
    cv[0] = prefix + gmx_tile_scale(loc, k_b, K);
#pragma unroll
    for (int c = 0; c < 4; ++c) cv[c + 1] = (i0 + c < n) ? prefix + gmx_tile_scale(loc + q[c], k_b, K) : total;
    int32_t e[5];
    uint32_t near_bits = 0;
#pragma unroll
    for (int c = 1; c <= 4; ++c) {
      const sb_est r = slots_below_est<kind>(key, u0, cv[c], total, n_over_total, eps, n32);
      e[c] = r.j;
      near_bits |= r.near ? (1u << c) : 0u;
    }
    {
      const sb_est r = slots_below_est<kind>(key, u0, cv[0], total, n_over_total, eps, n32);
      near_bits |= (lane == 0 && r.near) ? 1u : 0u;
      e[0] = (int32_t)wave_shr1_u32((uint32_t)e[4], (uint32_t)r.j);
      if (lane == 0) e[0] = r.j;
    }
    if (__any(near_bits != 0u)) {      // cold: the exact integer predicate for the flagged evaluations
      int32_t fixed0 = e[0];
#pragma unroll 1
      for (int c = 0; c <= 4; ++c) {
        if (near_bits & (1u << c)) {
          const uint64_t cc = c == 0 ? cv[0] : c == 1 ? cv[1] : c == 2 ? cv[2] : c == 3 ? cv[3] : cv[4];
          const int32_t j0 = c == 0 ? e[0] : c == 1 ? e[1] : c == 2 ? e[2] : c == 3 ? e[3] : e[4];
          const int32_t j = (int32_t)slots_below_exact(kind, key, (uint64_t)u0, cc, D, total, (int64_t)j0, n);
          if (c == 0) fixed0 = j; else if (c == 1) e[1] = j; else if (c == 2) e[2] = j; else if (c == 3) e[3] = j; else e[4] = j;
        }
      }
      const uint32_t up = wave_shr1_u32((uint32_t)e[4], (uint32_t)fixed0);
      e[0] = (lane == 0) ? fixed0 : (int32_t)up;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int32_t lo = e[c] > W ? e[c] : W;
      if (e[c + 1] > lo && lo < Wend) s_mark[lo - W] = (uint32_t)(i0 + c);
    }
    ++it;
  }

  // ---- C: fill (max-scan), store, transpose ----
  __syncthreads();
  uint4 mk = reinterpret_cast<const uint4*>(s_mark)[tid];
  mk.y = mk.y > mk.x ? mk.y : mk.x;
  mk.z = mk.z > mk.y ? mk.z : mk.y;
  mk.w = mk.w > mk.z ? mk.w : mk.z;
  const uint32_t incl = gmx_wave_umax_scan(mk.w);
  uint32_t carry = wave_shr1_u32(incl, 0u);
  if (lane == 63) s_wi[4 + wave] = incl;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 3; ++w) { const uint32_t v = s_wi[4 + w]; carry = (w < wave && v > carry) ? v : carry; }
  mk.x = mk.x > carry ? mk.x : carry;
  mk.y = mk.y > carry ? mk.y : carry;
  mk.z = mk.z > carry ? mk.z : carry;
  mk.w = mk.w > carry ? mk.w : carry;
  reinterpret_cast<uint4*>(s_mark)[tid] = mk;
  const int32_t j0 = W + 4 * tid;
  if (W + GMX_RS_TILE <= n32 && (((uintptr_t)Q.anc_out_d) & 15u) == 0u) {
    *reinterpret_cast<uint4*>(Q.anc_out_d + j0) = mk;
  } else {
    if (j0 + 0 < n32) Q.anc_out_d[j0 + 0] = (int32_t)mk.x;
    if (j0 + 1 < n32) Q.anc_out_d[j0 + 1] = (int32_t)mk.y;
    if (j0 + 2 < n32) Q.anc_out_d[j0 + 2] = (int32_t)mk.z;
    if (j0 + 3 < n32) Q.anc_out_d[j0 + 3] = (int32_t)mk.w;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) arow[p] = s_mark[p * GMX_BLOCK + tid];
}
