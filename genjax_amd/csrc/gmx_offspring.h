// gmx_offspring.h — the body of the fused ordered resampler (k_offspring_tile, gmx_kernels.hip), as a device function:
// the AOT kernel is a wrapper around it, and a specialised site program that gathers can run it FIRST, in its own launch
// (gmx_run_args.rs; TAGGED = true): a bootstrap SMC step is then ONE launch [resample step t-1 ; extend to step t].
// Device only.
#pragma once
#include "gmx_block.h"
#include "gmx_resample.h"
#include "gmx_sorted.h"

#define CDF_VEC_OT 4
#ifndef RS_THREADS
#define RS_THREADS 256                 /* 4 waves per tile                         */
#endif
#define RS_TILE (RS_THREADS * CDF_VEC_OT) /* 1024 log-weights per tile = one definition tile of the CDF */
#ifndef RS_MAX_TILES
#define RS_MAX_TILES 2048              /* n <= 2^21                                */
#endif
static_assert(RS_TILE == 1024 && RS_THREADS == GMX_BLOCK, "resampling tiles are the CDF's definition tiles");
static_assert(RS_MAX_TILES % GMX_BLOCK == 0, "tile table shape");

// TAGGED (a site program's prologue): every ancestor is stored as {tag: bits 24..31 | index: bits 0..23} with a
// write-through (sc1) store — the workgroups of the SAME launch that gather through these slots poll them until the tag
// is the step's (4-byte stores are single-copy atomic: the data is its own flag; no fence, no counter).
#define GMX_ANC_TAG_SHIFT 24           /* {tag: bits 24..31 | index: bits 0..23}: 16.7 M particles per launch, tags 1 .. 255 */
#define GMX_ANC_INDEX_MASK 0xffffffu
#define GMX_ANC_TAG_MAX 255u           /* (every slot is rewritten at every step: a stale word is at most one use old) */
__device__ __forceinline__ void gmx_store_u32x4_sc1(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 v; v.x = a; v.y = b; v.z = c; v.w = d;
  asm volatile("s_nop 1\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gmx_store_u32_sc1(uint32_t* p, uint32_t a) {
  __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- GMX_RESAMPLE_MULTINOMIAL_SORTED: "slots below a CDF value" from the order-statistics table (csrc/gmx_sorted.h) ----
typedef uint32_t rs_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));    // 16-byte load / store at a 4-byte-aligned address
typedef uint32_t sorted_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte load at a 4-byte-aligned address
struct sorted_ctx {
  const uint32_t* slow; const uint32_t* guide; const uint64_t* toff;
  uint64_t stot, gmax; uint32_t sh, mask; double ratio;     // ratio = S_total / total; gmax = the last bucket
};
__device__ __forceinline__ sorted_ctx sorted_ctx_of(const uint32_t* table, int64_t n, uint64_t total, uint64_t stot, uint32_t sh) {
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  sorted_ctx X;
  X.slow = table; X.guide = table + L.off_guide;
  X.toff = reinterpret_cast<const uint64_t*>(table + L.off_toff);
  X.stot = stot;              // = toff[tiles], table[off_sh]: loaded by the caller, early
  X.sh = sh;
  X.mask = (1u << X.sh) - 1u;
  X.gmax = X.stot >> X.sh;
  X.ratio = (double)X.stot / (double)(total ? total : 1ull);
  return X;
}
__device__ __forceinline__ int64_t sorted_below_exact(const sorted_ctx& X, uint64_t c, uint64_t total, int64_t j, int64_t n) {
  const u128 R = mul64(c, X.stot);
  j = j < 0 ? 0 : (j > n ? n : j);
  auto below = [&](int64_t k) {
    const uint64_t off = X.toff[k >> 10];
    const uint64_t S = off + (uint64_t)(uint32_t)(X.slow[k] - (uint32_t)off);     // S_k - off < 2^31
    return gt128(R, mul64(S, total));
  };
  while (j > 0 && !below(j - 1)) --j;
  while (j < n && below(j)) ++j;
  return j;
}

// One thread owns 4 consecutive sources (one float4 of log-weights); 256 threads are one tile; a block is RS_TPB
// consecutive tiles.  RS_TPB = 1.  Every block has to turn ALL tile statistics into its prefix and the total, and
// with RS_TPB = 4 (1024 threads) that pass is shared by four tiles — measured on MI355X (config 2): the kernel
// itself 7.8 -> 7.3 us per launch, but the site program that follows went 13.1 -> 14.9 us: with one tile per
// 256-thread block, tile j of BOTH kernels runs as block j, i.e. on XCD j mod 8 (blocks are dealt round-robin), so
// the log-weights, ancestors and states one kernel leaves are found in that XCD's L2 by the other; four consecutive
// tiles per block put three of them on another XCD.  (An interleaved assignment — tiles b mod 8 + 8 k — would keep the
// affinity; it needs four separate prefixes per block and gives most of the saving back.)
// Every load is issued before anything waits (unconditional, clamped addresses), wave-level reductions and scans are
// DPP (gmx_block.h), the slot ranges are straight-line f64 code with one cold exact path.
#define RS_TPB 1
#define RS_BLOCK (GMX_BLOCK * RS_TPB)
#define RS_WAVES (RS_BLOCK / GMX_WAVE)
static_assert(RS_MAX_TILES % RS_BLOCK == 0, "tile table shape");
// PER: rows of the tile table a thread holds (PER * 256 >= n_tiles; 1, 2, 4 or 8 — the launch picks the smallest):
// the statistics pass is unrolled over exactly the rows that exist.
#define RS_FILL_SLOTS 2048             /* slots filled per pass (8 per thread): a tile owns ~1024 */
// A tile's slots [T0, T1) get their ancestors through LDS: every source with a slot writes its index at its first one,
// a max-scan fills the rest (source indices increase with the slot), and the block stores 8 consecutive slots per
// thread (two 16-byte stores): the same cost whatever the weights (a thread writing its own sources' slots in a loop
// diverges over the offspring counts: measured 2688 -> 34 us for N(0, 4) log-weights); a tile owning more than 2048
// slots takes more passes (block-uniform).
template <int kind, int PER, bool TAGGED>
__device__ __forceinline__ void
gmx_offspring_tile_body(uint32_t k0, uint32_t k1, uint32_t u0_host, const float* __restrict__ lw,
                        const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int64_t n, int n_tiles, float scale,
                        float* __restrict__ max_out, uint64_t* __restrict__ total_out, int32_t* __restrict__ anc,
                        const uint32_t* __restrict__ uslot, uint32_t tag,
                        int tile0 = -1, uint64_t a_prefix = 0ull, uint64_t a_total = 0ull, uint64_t a_mk = 0ull) {
  // tile0 >= 0: the tile this call works on, when it is not the workgroup's number — a LOOPED launch (gmx_jit.h,
  // GMX_JIT_RS_LOOP: more tiles than the device holds workgroups; each workgroup handles tiles blockIdx + c * gridDim).
  // PER == 0 with agg == nullptr: the tile's prefix, the total and (M, K) arrive as arguments (gmx_tile_table_pass below).
  const int blk = tile0 >= 0 ? tile0 : (int)blockIdx.x;
  const uint32_t tagw = TAGGED ? (tag << GMX_ANC_TAG_SHIFT) : 0u;
  __shared__ uint64_t s_below[RS_WAVES], s_all[RS_WAVES], s_scan[RS_WAVES], s_g[RS_TPB];
  __shared__ float s_max[RS_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = threadIdx.x >> 8, ltid = threadIdx.x & (GMX_BLOCK - 1);      // which of the block's tiles / thread within it
  const int first_tile = blk * RS_TPB;
  const int my_tile = first_tile + grp;
  const bool tile_ok = my_tile < n_tiles;                                        // wave-uniform
  const int tile_c = tile_ok ? my_tile : n_tiles - 1;
  const int64_t i0 = (int64_t)tile_c * RS_TILE + (int64_t)ltid * CDF_VEC_OT;
  // ---- issue every load first ----
  float x[CDF_VEC_OT];
  const bool full_tile = (int64_t)(tile_c + 1) * RS_TILE <= n;                   // wave-uniform
  if (full_tile) {
    float4 v = *reinterpret_cast<const float4*>(lw + i0);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC_OT; ++c) {
      const int64_t ic = i0 + c < n ? i0 + c : n - 1;
      const float xv = lw[ic];
      x[c] = (i0 + c < n) ? xv : -gmx_inf();
    }
  }
  // PER == 0: the tile PREFIXES are there already (`agg` = the block gmx_tile_prefix wrote): this workgroup reads its
  // own prefix, the total and (M, K) — three loads instead of a pass over the whole table.
  constexpr bool PREF = (PER == 0);
  constexpr int PERN = PREF ? 1 : PER;
  static_assert(PER >= 0 && PER * RS_BLOCK <= RS_MAX_TILES, "rows of the tile table per thread");
  static_assert(!PREF || RS_TPB == 1, "the prefix form works on one tile per block");
  uint64_t ta[PERN];
  float tm[PERN];
  uint64_t pf_prefix = 0, pf_total = 0, pf_mk = 0;
  if (PREF) {
    if (agg) { pf_prefix = agg[tile_c]; pf_total = agg[n_tiles]; pf_mk = agg[n_tiles + 1]; }
    else { pf_prefix = a_prefix; pf_total = a_total; pf_mk = a_mk; }
  } else {
#pragma unroll
    for (int r = 0; r < PERN; ++r) {              // loads only (clamped rows): nothing here waits
      ta[r] = 0ull; tm[r] = -gmx_inf();
      if (r * RS_BLOCK < n_tiles) {                // uniform: rows of the table that exist
        const int t = r * RS_BLOCK + (int)threadIdx.x;
        const int tc = t < n_tiles ? t : n_tiles - 1;
        ta[r] = agg[tc];
        tm[r] = tmax[tc];
      }
    }
  }
  const float tmax_mine = tmax[tile_c];
  uint64_t sx_stot = 0; uint32_t sx_sh = 0;             // the sorted kind's table header (uniform), with the other early loads
  if constexpr (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED) {
    const gmx_sorted_layout L = gmx_sorted_layout_of(n);
    sx_stot = reinterpret_cast<const uint64_t*>(uslot + L.off_toff)[L.tiles];
    sx_sh = uslot[L.off_sh];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!PREF) {
#pragma unroll
    for (int r = 0; r < PERN; ++r) {
      const int t = r * RS_BLOCK + (int)threadIdx.x;
      ta[r] = (t < n_tiles) ? ta[r] : 0ull;
      tm[r] = (t < n_tiles) ? tm[r] : -gmx_inf();
    }
  }
  const int32_t k_b = gmx_tile_exp(tmax_mine);
  const float ref_b = gmx_tile_ref(k_b);
  // phase 1: the global max, and the wave totals of each tile's local weights
  float M = -gmx_inf();
  if (PREF) {
    M = gmx_u2f((uint32_t)pf_mk);
  } else {
#pragma unroll
    for (int r = 0; r < PERN; ++r)
      if (r * RS_BLOCK < n_tiles) M = gmx_rmax(M, tm[r]);
    M = wave_max(M);
  }
  uint64_t q[CDF_VEC_OT];
  uint64_t run = 0;
#pragma unroll
  for (int c = 0; c < CDF_VEC_OT; ++c) {
    const uint64_t w = weight_fixed(x[c], ref_b, scale);
    run += (tile_ok && i0 + c < n) ? w : 0ull;
    q[c] = run;
  }
  const uint64_t inc = wave_scan_u64(run);
  if (!PREF && lane == 0) s_max[wave] = M;
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  if (!PREF) {
    M = s_max[0];
#pragma unroll
    for (int w = 1; w < RS_WAVES; ++w) M = gmx_rmax(M, s_max[w]);
  }
  uint64_t wave_off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_off += (w < (wave & 3)) ? s_scan[4 * grp + w] : 0ull;     // the tile's own four waves
  const uint64_t loc = wave_off + (inc - run);          // tile-local mass before this thread's sources
  // phase 2: G_t = A_t * 2^(k_t - K) for every tile -> the mass before this block's first tile, the block's own
  // four G, and the total
  const int32_t K = PREF ? (int32_t)(uint32_t)(pf_mk >> 32) : gmx_tile_exp(M);
  uint64_t prefix = 0, total = 0;
  if (PREF) {
    prefix = pf_prefix; total = pf_total;
  } else {
    uint64_t below = 0, all = 0;
#pragma unroll
    for (int r = 0; r < PERN; ++r) {
      if (r * RS_BLOCK < n_tiles) {
        const int t = r * RS_BLOCK + (int)threadIdx.x;
        const uint64_t G = gmx_tile_scale(ta[r], gmx_tile_exp(tm[r]), K);
        all += G;
        below += (t < first_tile) ? G : 0ull;
        if (t >= first_tile && t < first_tile + RS_TPB) s_g[t - first_tile] = G;      // t >= n_tiles: G = 0
      }
    }
    below = wave_sum_u64(below);
    all = wave_sum_u64(all);
    if (lane == 0) { s_below[wave] = below; s_all[wave] = all; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) { prefix += s_below[w]; total += s_all[w]; }
#pragma unroll
    for (int j = 0; j < RS_TPB; ++j) prefix += (j < grp) ? s_g[j] : 0ull;
  }
  if (blk == 0 && threadIdx.x == 0) { *total_out = total; *max_out = M; }
  if (!tile_ok) return;                                  // a whole tile group past the end (uniform per wave; no barrier follows)
  gmx_key key; key.k0 = k0; key.k1 = k1;
  if (total == 0) {       // no mass at all (all weights -inf / NaN): every slot maps to the last particle
#pragma unroll
    for (int c = 0; c < CDF_VEC_OT; ++c)
      if (i0 + c < n) {
        if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + (i0 + c), (uint32_t)(n - 1) | tagw);
        else anc[i0 + c] = (int32_t)(n - 1);
      }
    return;
  }
  const double n_over_total = (double)n / (double)total;
  const double eps = (double)n * 0x1p-44 + 0x1p-40;
  const int32_t n32 = (int32_t)n;
  // CDF values at the lower edge of the thread's first source and at the upper edge of each source
  uint64_t cv[CDF_VEC_OT + 1];
  cv[0] = prefix + gmx_tile_scale(loc, k_b, K);
#pragma unroll
  for (int c = 0; c < CDF_VEC_OT; ++c) cv[c + 1] = (i0 + c < n) ? prefix + gmx_tile_scale(loc + q[c], k_b, K) : total;
  int32_t e[CDF_VEC_OT + 1];
  bool near = false;
  uint32_t near_bits = 0;
  constexpr bool SORTED = (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED);      // `uslot` is the order-statistics table
  sorted_ctx SX;
  if constexpr (SORTED) SX = sorted_ctx_of(uslot, n, total, sx_stot, sx_sh);
  auto below_est = [&](uint64_t c) -> sb_est {
    return slots_below_est<kind == GMX_RESAMPLE_MULTINOMIAL_SORTED ? GMX_RESAMPLE_SYSTEMATIC : kind>(
        key, u0_host, c, total, n_over_total, eps, n32, uslot);              // (never called for the sorted kind)
  };
  sb_est R[CDF_VEC_OT + 1];
  if constexpr (SORTED) {
    // the evaluations side by side: every guide read issued before any is used, then every slot probe — written one
    // evaluation after the other, the (rare) walk past the probed slots orders the loads: ten dependent round trips.
    // cv[0] (the lower edge of the thread's first source) is the previous thread's cv[4]: lanes take it from their
    // neighbour below, lane 0 of waves 1 .. 3 from the wave before through LDS (after the cold path, so a corrected
    // value travels), and only wave 0 evaluates its own (the tile's first edge).
    uint32_t mv[CDF_VEC_OT + 1], gv[CDF_VEC_OT + 1], lov[CDF_VEC_OT + 1], hiv[CDF_VEC_OT + 1], kv[CDF_VEC_OT + 1];
    auto eval = [&](const int c0, const int c1) {
#pragma unroll
      for (int c = c0; c <= c1; ++c) {
        const double cd = __builtin_fma((double)(uint32_t)(cv[c] >> 32), 4294967296.0, (double)(uint32_t)cv[c]);
        const double t = cd * SX.ratio;
        const uint64_t tq = (uint64_t)t;
        const double frac = t - (double)tq;
        const double teps = __builtin_fma(t, 0x1p-49, 0x1p-40);
        R[c].near = (frac < teps) || (frac > 1.0 - teps);
        const uint64_t gq = tq >> SX.sh;
        gv[c] = (uint32_t)(gq < SX.gmax ? gq : SX.gmax);      // (t within rounding of S_total: entries past gmax + 1 are not defined)
        mv[c] = (uint32_t)tq & SX.mask;
      }
#pragma unroll
      for (int c = c0; c <= c1; ++c) {
        const sorted_u32x2_a4 gh = *reinterpret_cast<const sorted_u32x2_a4*>(SX.guide + gv[c]);
        lov[c] = gh.x; hiv[c] = gh.y;
      }
#pragma unroll
      for (int c = c0; c <= c1; ++c) {
        hiv[c] = hiv[c] < (uint32_t)n32 ? hiv[c] : (uint32_t)n32;
        lov[c] = lov[c] < hiv[c] ? lov[c] : hiv[c];
      }
      rs_u32x4_a4 pa[CDF_VEC_OT + 1], pb[CDF_VEC_OT + 1];
#pragma unroll
      for (int c = c0; c <= c1; ++c) {        // eight slots per evaluation (past slot n - 1: still inside the table, not counted)
        pa[c] = *reinterpret_cast<const rs_u32x4_a4*>(SX.slow + lov[c]);
        pb[c] = *reinterpret_cast<const rs_u32x4_a4*>(SX.slow + lov[c] + 4u);
      }
      bool more = false;
#pragma unroll
      for (int c = c0; c <= c1; ++c) {        // the slots of a bucket ascend: the count of "<= m" is the count of leading ones
        const uint32_t pv[8] = {pa[c].x, pa[c].y, pa[c].z, pa[c].w, pb[c].x, pb[c].y, pb[c].z, pb[c].w};
        uint32_t k = lov[c];
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) k += (lov[c] + i < hiv[c] && (pv[i] & SX.mask) <= mv[c]) ? 1u : 0u;
        kv[c] = k;
        more |= (k == lov[c] + 8u && k < hiv[c]);
      }
      if (more) {                              // a bucket with more than eight slots below t: once in ~1e5 evaluations
#pragma unroll 1
        for (int c = c0; c <= c1; ++c) {
          uint32_t k = kv[c];
          if (k == lov[c] + 8u)
            while (k < hiv[c] && (SX.slow[k] & SX.mask) <= mv[c]) ++k;
          kv[c] = k;
        }
      }
#pragma unroll
      for (int c = c0; c <= c1; ++c) {
        R[c].j = (int32_t)kv[c];
        if (cv[c] == 0ull) { R[c].j = 0; R[c].near = false; }
        if (cv[c] >= total) { R[c].j = n32; R[c].near = false; }
      }
    };
    R[0].j = 0; R[0].near = false;
    eval(1, CDF_VEC_OT);
    if (wave == 0) eval(0, 0);
  } else {
#pragma unroll
    for (int c = 1; c <= CDF_VEC_OT; ++c) R[c] = below_est(cv[c]);
    R[0] = below_est(cv[0]);
  }
#pragma unroll
  for (int c = 1; c <= CDF_VEC_OT; ++c) {
    e[c] = R[c].j;
    near_bits |= R[c].near ? (1u << c) : 0u;
  }
  // lower bound of the thread's first source = upper bound of the previous thread's last one; lane 0 evaluates its own
  {
    const sb_est r = R[0];
    near_bits |= (lane == 0 && r.near) ? 1u : 0u;
    e[0] = (int32_t)wave_shr1_u32((uint32_t)e[CDF_VEC_OT], (uint32_t)r.j);
    if (lane == 0) e[0] = r.j;
  }
  near = near_bits != 0u;
  if (__any(near)) {
    // cold: the exact integer predicate for the flagged evaluations (rolled; operands picked by selects)
    const uint64_t D = (uint64_t)n << 23;
    int32_t fixed0 = e[0];
#pragma unroll 1
    for (int c = 0; c <= CDF_VEC_OT; ++c) {
      if (near_bits & (1u << c)) {
        const uint64_t cc = c == 0 ? cv[0] : c == 1 ? cv[1] : c == 2 ? cv[2] : c == 3 ? cv[3] : cv[4];
        const int32_t j0 = c == 0 ? e[0] : c == 1 ? e[1] : c == 2 ? e[2] : c == 3 ? e[3] : e[4];
        int32_t j;
        if constexpr (SORTED) j = (int32_t)sorted_below_exact(SX, cc, total, (int64_t)j0, n);
        else j = (int32_t)slots_below_exact(kind, key, (uint64_t)u0_host, cc, D, total, (int64_t)j0, n);
        if (c == 0) fixed0 = j; else if (c == 1) e[1] = j; else if (c == 2) e[2] = j; else if (c == 3) e[3] = j; else e[4] = j;
      }
    }
    // a corrected upper bound is the next lane's lower bound
    const uint32_t up = wave_shr1_u32((uint32_t)e[CDF_VEC_OT], (uint32_t)fixed0);
    e[0] = (lane == 0) ? fixed0 : (int32_t)up;
  }
  if constexpr (SORTED) {        // lane 0 of waves 1 .. 3: the (settled) upper edge of the wave before
    __shared__ int32_t s_edge[RS_WAVES];
    if (lane == 63) s_edge[wave] = e[CDF_VEC_OT];
    __syncthreads();
    if (lane == 0 && wave > 0) e[0] = s_edge[wave - 1];
  }
  // Sources past n have e[c] = n = e of the last real source, so they own no slot.
  const int32_t e4 = e[4];
  const int32_t src0 = (int32_t)i0;
  static_assert(RS_TPB == 1, "the LDS fill works on one tile per block");
  __shared__ __attribute__((aligned(16))) uint32_t s_mark[RS_FILL_SLOTS];
  __shared__ int32_t s_rng[2];
  __shared__ uint32_t s_carry[RS_WAVES];
  if (threadIdx.x == 0) s_rng[0] = e[0];
  if (threadIdx.x == RS_BLOCK - 1) s_rng[1] = e4;
  reinterpret_cast<uint4*>(s_mark)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
  reinterpret_cast<uint4*>(s_mark)[threadIdx.x + RS_BLOCK] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int32_t T0 = __builtin_amdgcn_readfirstlane(s_rng[0]), T1 = __builtin_amdgcn_readfirstlane(s_rng[1]);
  for (int32_t base = T0; base < T1; base += RS_FILL_SLOTS) {      // block-uniform trip count (1 unless the tile owns > 2048 slots)
    if (base != T0) {
      reinterpret_cast<uint4*>(s_mark)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
      reinterpret_cast<uint4*>(s_mark)[threadIdx.x + RS_BLOCK] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    // a source's first slot inside this pass (clipped at `base`: a range that began in an earlier pass continues at 0)
#pragma unroll
    for (int c = 0; c < CDF_VEC_OT; ++c) {
      const int32_t lo = e[c] > base ? e[c] : base;
      if (e[c + 1] > lo && lo - base < RS_FILL_SLOTS) s_mark[lo - base] = (uint32_t)(src0 + c);
    }
    __syncthreads();
    uint4 a = reinterpret_cast<const uint4*>(s_mark)[2 * threadIdx.x];
    uint4 b = reinterpret_cast<const uint4*>(s_mark)[2 * threadIdx.x + 1];
    a.y = a.y > a.x ? a.y : a.x; a.z = a.z > a.y ? a.z : a.y; a.w = a.w > a.z ? a.w : a.z;
    b.x = b.x > a.w ? b.x : a.w; b.y = b.y > b.x ? b.y : b.x; b.z = b.z > b.y ? b.z : b.y; b.w = b.w > b.z ? b.w : b.z;
    const uint32_t incl = gmx_wave_umax_scan(b.w);
    uint32_t carry = wave_shr1_u32(incl, 0u);
    if (lane == 63) s_carry[wave] = incl;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < RS_WAVES - 1; ++w) { const uint32_t v = s_carry[w]; carry = (w < wave && v > carry) ? v : carry; }
    a.x = a.x > carry ? a.x : carry; a.y = a.y > carry ? a.y : carry; a.z = a.z > carry ? a.z : carry; a.w = a.w > carry ? a.w : carry;
    b.x = b.x > carry ? b.x : carry; b.y = b.y > carry ? b.y : carry; b.z = b.z > carry ? b.z : carry; b.w = b.w > carry ? b.w : carry;
    const int32_t j = base + 8 * (int32_t)threadIdx.x;
    if (j + 8 <= T1) {
      rs_u32x4_a4 va, vb;
      va.x = a.x; va.y = a.y; va.z = a.z; va.w = a.w; vb.x = b.x; vb.y = b.y; vb.z = b.z; vb.w = b.w;
      if (TAGGED) {
        gmx_store_u32x4_sc1(reinterpret_cast<uint32_t*>(anc) + j, va.x | tagw, va.y | tagw, va.z | tagw, va.w | tagw);
        gmx_store_u32x4_sc1(reinterpret_cast<uint32_t*>(anc) + j + 4, vb.x | tagw, vb.y | tagw, vb.z | tagw, vb.w | tagw);
      } else {
        *reinterpret_cast<rs_u32x4_a4*>(anc + j) = va;
        *reinterpret_cast<rs_u32x4_a4*>(anc + j + 4) = vb;
      }
    } else {
      if (j + 0 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 0, a.x | tagw); else anc[j + 0] = (int32_t)a.x; }
      if (j + 1 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 1, a.y | tagw); else anc[j + 1] = (int32_t)a.y; }
      if (j + 2 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 2, a.z | tagw); else anc[j + 2] = (int32_t)a.z; }
      if (j + 3 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 3, a.w | tagw); else anc[j + 3] = (int32_t)a.w; }
      if (j + 4 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 4, b.x | tagw); else anc[j + 4] = (int32_t)b.x; }
      if (j + 5 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 5, b.y | tagw); else anc[j + 5] = (int32_t)b.y; }
      if (j + 6 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 6, b.z | tagw); else anc[j + 6] = (int32_t)b.z; }
      if (j + 7 < T1) { if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(anc) + j + 7, b.w | tagw); else anc[j + 7] = (int32_t)b.w; }
    }
  }
}



// The statistics pass of a LOOPED launch (GMX_JIT_RS_LOOP): the workgroup handles the C tiles blk, blk + G, blk + 2 G, ...
// and walks the whole table of n_tiles entries TWICE: the global max first (the exponent every G_t is scaled to), then, chunk
// by chunk of G entries, the chunk's sum and the sum of its entries in front of this workgroup's tile — prefix[c] = (all
// chunks before c) + (entries cG .. cG + blk - 1): the same integers gmx_tile_prefix_block / k_tile_prefix_big produce (sums
// of u64 are exact in any order; a max does not care about order).  One pass per WORKGROUP, not per tile: at 8e6 particles
// (7813 tiles, 12 bytes each) every workgroup reads 2 x 94 KB from L2.  Two accumulators per thread whatever C.
__device__ __forceinline__ void
gmx_tile_table_pass(const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int n_tiles, int blk, int G, int C,
                    uint64_t* __restrict__ s_pf /* LDS, C entries */, uint64_t& total, uint64_t& mk) {
  __shared__ float s_tm[RS_WAVES];
  __shared__ uint64_t s_acc[RS_WAVES][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float M = -gmx_inf();
  for (int t = (int)threadIdx.x; t < n_tiles; t += RS_BLOCK) M = gmx_rmax(M, tmax[t]);
  M = wave_max(M);
  if (lane == 0) s_tm[wave] = M;
  __syncthreads();
  M = s_tm[0];
#pragma unroll
  for (int w = 1; w < RS_WAVES; ++w) M = gmx_rmax(M, s_tm[w]);
  const int32_t K = gmx_tile_exp(M);
  uint64_t running = 0;
#pragma clang loop unroll(disable)
  for (int c = 0; c < C; ++c) {
    const int lo = c * G, hi = (lo + G < n_tiles) ? lo + G : n_tiles, mine = lo + blk;
    uint64_t tot = 0, part = 0;
    for (int t = lo + (int)threadIdx.x; t < hi; t += RS_BLOCK) {
      const uint64_t g_ = gmx_tile_scale(agg[t], gmx_tile_exp(tmax[t]), K);
      tot += g_;
      part += (t < mine) ? g_ : 0ull;
    }
    tot = wave_sum_u64(tot);
    part = wave_sum_u64(part);
    if (lane == 0) { s_acc[wave][0] = tot; s_acc[wave][1] = part; }
    __syncthreads();
    uint64_t T_ = 0, P_ = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) { T_ += s_acc[w][0]; P_ += s_acc[w][1]; }
    if (threadIdx.x == 0) s_pf[c] = running + P_;
    running += T_;
    __syncthreads();
  }
  total = running;
  mk = (uint64_t)gmx_f2u(M) | ((uint64_t)(uint32_t)K << 32);
}
