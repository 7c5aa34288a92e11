// gmx_shard_fill.h — the body of the sharded step's routing kernel (k_shard_step_fill, gmx_kernels.hip) as a device
// function: the AOT kernel is a wrapper around it, and a specialised site program of a SHARDED sweep can run it FIRST,
// in its own launch (gmx_run_args.sh; TAGGED = true): a sharded SMC step is then ONE launch [route step t-1 ; extend to
// step t] and no collective (include/genmi.h "Fused peer exchange").  Device only.
#pragma once
#include "gmx_block.h"
#include "gmx_resample.h"
#include "gmx_offspring.h"
#include "gmx_peer.h"

#define SHARD_MAX_WORLD 64
#ifndef CDF_VEC
#define CDF_VEC 4                      /* consecutive items per thread (one float4) */
#endif

// f(c) = number of slots j in [0, n_out) with P_j < c * D, i.e. with
//   j + u_j / 2^23 < v,  v = c * n_out / total.
// Fast path: v in f64 (relative error <= 2^-50) decides everything unless it is
// within `eps` of the boundary; only then the exact 128-bit predicate is used,
// so the answer is ALWAYS the one the integer definition gives.
__device__ __forceinline__ int64_t slots_below(int kind, gmx_key key, uint64_t u0, uint64_t c, uint64_t D,
                                               uint64_t total, double n_over_total, double eps, int64_t n_out) {
  if (c == 0) return 0;
  if (c >= total) return n_out;
  // (double)c as hi * 2^32 + lo: the product is exact and the sum rounds once, i.e. the correctly rounded
  // conversion, in 3 instructions instead of the generic 64-bit sequence (this function is the hot loop)
  const double v = ((double)(uint32_t)(c >> 32) * 4294967296.0 + (double)(uint32_t)c) * n_over_total;
  int64_t j0;
  double dj0;
  if (n_out <= 0x7fffffffLL) {                   // uniform: 32-bit conversions
    int32_t t = (int32_t)v;                      // floor (v >= 0), saturating
    if (t >= (int32_t)n_out) t = (int32_t)n_out - 1;
    j0 = t; dj0 = (double)t;
  } else {
    j0 = (int64_t)v;
    if (j0 >= n_out) j0 = n_out - 1;
    dj0 = (double)j0;
  }
  const double frac = v - dj0;
  bool exact = (frac < eps) || (frac > 1.0 - eps);
  int64_t j = j0;
  if (!exact) {
    // slots j < j0 are below; slot j0 is below iff u_{j0} / 2^23 < frac
    uint32_t u = (kind == GMX_RESAMPLE_SYSTEMATIC) ? (uint32_t)u0 : (gmx_bits32(key, (uint64_t)j0) >> 9);
    double du = (double)u * (1.0 / 8388608.0);
    double diff = frac - du;
    if (diff > eps) return j0 + 1;
    if (diff < -eps) return j0;
    exact = true;
  }
  // rare: within eps of a boundary -> exact integer predicate  P_j < X
  return slots_below_exact(kind, key, u0, c, D, total, j, n_out);
}

__device__ __forceinline__ int64_t shard_readlane64(int64_t x, int l) {
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), l) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l));
}
struct shard_peer {                              // what the body reads
  uint64_t* const* land;                         // device array [world]: every rank's landing block, mapped here
  const uint32_t* tag_base; uint64_t* status;
  int32_t step, leaves;
  const uint32_t* const* state;                  // [leaves]: leaf l = this rank's states [n]           } arrays that live in
  uint32_t* const* tail;                         // [leaves]: leaf l = the tail [world * cap] of its extended state } KERNEL-ARGUMENT
};                                               // memory: indexed at run time without a trip through scratch
struct shard_peer_args {                         // the AOT kernel's by-value parameter (the arrays themselves)
  uint64_t* const* land;
  const uint32_t* tag_base; uint64_t* status;
  int32_t step, leaves;
  const uint32_t* state[GMX_PEER_MAX_LEAVES];
  uint32_t* tail[GMX_PEER_MAX_LEAVES];
};
// one row of another rank's statistics: spin (bounded) until all three granules carry `tag`
__device__ __forceinline__ bool shard_peer_row(const uint64_t* row, uint32_t tag, uint64_t& agg, float& tmax) {
  uint64_t g0 = gmx_granule_peek(row), g1 = gmx_granule_peek(row + 1), g2 = gmx_granule_peek(row + 2);
  bool ok = (uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag && (uint32_t)(g2 >> 32) == tag;
  if (!ok) {
    const uint64_t t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(2);
      g0 = gmx_granule_peek(row); g1 = gmx_granule_peek(row + 1); g2 = gmx_granule_peek(row + 2);
      ok = (uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag && (uint32_t)(g2 >> 32) == tag;
    } while (!ok && wall_clock64() - t0 < GMX_PEER_TIMEOUT_TICKS);
  }
  agg = ok ? ((uint64_t)(uint32_t)g0 | ((uint64_t)(uint32_t)g1 << 32)) : 0ull;
  tmax = ok ? gmx_u2f((uint32_t)g2) : -gmx_inf();
  return ok;
}
template <int kind, bool SMALL, bool PEER, bool TAGGED>
__device__ __forceinline__ void
gmx_shard_fill_body(uint32_t k0, uint32_t k1, uint32_t u0_host, const float* __restrict__ lw,
                    const uint8_t* __restrict__ stats_all, size_t stride, int n_tiles, float scale, int rank, int world,
                    int32_t n, int32_t cap, int64_t* __restrict__ plan, uint64_t* __restrict__ total_out,
                    float* __restrict__ max_out, const uint32_t* __restrict__ state, uint32_t* __restrict__ send,
                    int32_t* __restrict__ next_idx, const shard_peer& P, uint32_t anc_tag) {
  // TAGGED (a site program's prologue, gmx_run_args.sh): every ancestor index is stored as {tag: bits 24..31 | index}
  // with a write-through store — the workgroups of the SAME launch that gather through these slots poll them until
  // the tag is the launch's (gmx_offspring.h); a slot whose ancestor is remote gets its word AFTER the received state
  // is in the local tail (release fence in between): the word announces the value.
  const uint32_t tagw = TAGGED ? (anc_tag << GMX_ANC_TAG_SHIFT) : 0u;
  auto set_idx = [&](int32_t slot_local, uint32_t v) {
    if (TAGGED) gmx_store_u32_sc1(reinterpret_cast<uint32_t*>(next_idx) + slot_local, v | tagw);
    else next_idx[slot_local] = (int32_t)v;
  };
  // (a release FENCE at agent scope writes back the whole L2 of this XCD, an acquire fence invalidates it — measured:
  //  the sweep at 42 us/step against 19 with every thread fencing once.  What is needed is less: the received state
  //  goes to the tail as a write-through store, its acknowledgement is awaited (s_waitcnt), THEN the word is stored,
  //  also write-through; the reader's load of the tail row depends on the word it polled and misses every cache)
  auto announce = [&]() { if (TAGGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  __shared__ uint64_t s_part[SHARD_MAX_WORLD][4];
  __shared__ uint64_t s_below[4], s_scan[4];
  __shared__ float s_max[4];
  __shared__ int32_t s_bounds[SHARD_MAX_WORLD + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_mark[RS_FILL_SLOTS];
  __shared__ int32_t s_rng[2];
  __shared__ uint32_t s_carry[4];
  __shared__ uint32_t s_gave_up[4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int my_tile = (int)blockIdx.x;
  const int32_t i0 = my_tile * RS_TILE + tid * CDF_VEC;
  const int32_t N = n * world, base = rank * n;
  const int tiles_pad = n_tiles + (n_tiles & 1);
  gmx_key key; key.k0 = k0; key.k1 = k1;
  // ---- loads first: this tile's log-weights ----
  float x[CDF_VEC];
  if ((my_tile + 1) * RS_TILE <= n) {
    const float4 v = *reinterpret_cast<const float4*>(lw + i0);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) {
      const int32_t ic = i0 + c < n ? i0 + c : n - 1;
      const float xv = lw[ic];
      x[c] = (i0 + c < n) ? xv : -gmx_inf();
    }
  }
  // PEER: `stats_all` is this rank's own block; the other ranks' rows are granules in the landing block
  const uint8_t* own = PEER ? stats_all : stats_all + (size_t)rank * stride;
  const float tmax_mine = reinterpret_cast<const float*>(own + (size_t)tiles_pad * 8)[my_tile];
  uint32_t tag = 0u;
  const uint64_t* land_own = nullptr;            // this rank's own landing block (entry `rank` of the table)
  bool timed_out = false;     // (PEER) a wait gave up — or the sweep had lost a peer before this launch: no wait at all then
  if (PEER) {
    tag = *P.tag_base + (uint32_t)P.step; land_own = P.land[rank];
    // a sweep that has already lost a peer does not wait again (the value is first needed where a wait would start, or
    // at the barrier below: the loads that follow are not held up by it)
    timed_out = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
  }
  // ---- pass 1 over the table: the global max ----
  float m = -gmx_inf();
  uint64_t ta[4];
  float tm[4];
  int rr[4], tt[4];
  if (SMALL) {
    const int rows = world * n_tiles;
    const float inv = 1.0f / (float)n_tiles;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rho = k * GMX_BLOCK + tid;
      const int rc = rho < rows ? rho : 0;
      // rc / n_tiles, exactly: rows <= 1024 and world <= 8 here, so (rc + 0.5) / n_tiles is at least 0.5 / 1024 away
      // from an integer and the two roundings (1 / n_tiles, the product) move it by less than 8 * 2^-23 — no fix-up
      const int r = (int)(((float)rc + 0.5f) * inv);
      const int t = rc - r * n_tiles;
      rr[k] = rho < rows ? r : -1;
      tt[k] = t;
      if (PEER && r != rank) {
        if (rho < rows && !timed_out) timed_out |= !shard_peer_row(land_own + gmx_peer_stats_at(tag, world, n_tiles, r, t), tag, ta[k], tm[k]);
        else if (rho < rows) { ta[k] = 0ull; tm[k] = -gmx_inf(); }
        else { ta[k] = 0ull; tm[k] = -gmx_inf(); }
      } else {
        const uint8_t* blk = PEER ? own : stats_all + (size_t)r * stride;
        ta[k] = reinterpret_cast<const uint64_t*>(blk)[t];
        tm[k] = reinterpret_cast<const float*>(blk + (size_t)tiles_pad * 8)[t];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) m = gmx_rmax(m, rr[k] >= 0 ? tm[k] : -gmx_inf());
  } else {
    for (int r = 0; r < world; ++r) {
      if (PEER && r != rank) {
        for (int t = tid; t < n_tiles; t += GMX_BLOCK) {
          uint64_t a_ = 0ull; float m_ = -gmx_inf();
          if (!timed_out) timed_out |= !shard_peer_row(land_own + gmx_peer_stats_at(tag, world, n_tiles, r, t), tag, a_, m_);
          m = gmx_rmax(m, m_);
        }
      } else {
        const float* tmax = reinterpret_cast<const float*>((PEER ? own : stats_all + (size_t)r * stride) + (size_t)tiles_pad * 8);
        for (int t = tid; t < n_tiles; t += GMX_BLOCK) m = gmx_rmax(m, tmax[t]);
      }
    }
  }
  // (PEER: a peer whose statistics never arrived — the vote rides on the barrier below, not on one of its own)
  m = wave_max(m);
  const int32_t k_b = gmx_tile_exp(tmax_mine);
  const float ref_b = gmx_tile_ref(k_b);
  uint64_t q[CDF_VEC], run = 0;
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) {
    run += (i0 + c < n) ? weight_fixed(x[c], ref_b, scale) : 0ull;
    q[c] = run;
  }
  const uint64_t inc = wave_scan_u64(run);
  if (lane == 0) s_max[wave] = m;
  if (lane == 63) s_scan[wave] = inc;
  if (PEER) { const bool gave_up = __any(timed_out ? 1 : 0) != 0; if (lane == 0) s_gave_up[wave] = gave_up ? 1u : 0u; }
  __syncthreads();
  if (PEER) {      // a peer whose statistics never arrived: the workgroup leaves (bounded, no garbage routed)
    if ((s_gave_up[0] | s_gave_up[1] | s_gave_up[2] | s_gave_up[3]) != 0u) {       // block-uniform
      if (tid == 0) { __hip_atomic_store(P.status, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); plan[GMX_PLAN_OVERFLOW] = 1; }
      return;
    }
  }
  const float M = gmx_rmax(gmx_rmax(s_max[0], s_max[1]), gmx_rmax(s_max[2], s_max[3]));
  const int32_t K = gmx_tile_exp(M);
  // ---- pass 2: every rank's total (partial sums per wave), and the mass of this rank's earlier tiles ----
  uint64_t below = 0;
  if (SMALL) {
    uint64_t G[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      G[k] = rr[k] >= 0 ? gmx_tile_scale(ta[k], gmx_tile_exp(tm[k]), K) : 0ull;
      below += (rr[k] == rank && tt[k] < my_tile) ? G[k] : 0ull;
    }
    for (int r = 0; r < world; ++r) {              // world <= 8
      uint64_t sum = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += (rr[k] == r) ? G[k] : 0ull;
      sum = wave_sum_u64(sum);
      if (lane == 0) s_part[r][wave] = sum;
    }
  } else {
    for (int r = 0; r < world; ++r) {
      const uint8_t* blk = PEER ? own : stats_all + (size_t)r * stride;
      const uint64_t* agg = reinterpret_cast<const uint64_t*>(blk);
      const float* tmax = reinterpret_cast<const float*>(blk + (size_t)tiles_pad * 8);
      uint64_t sum = 0;
      for (int t = tid; t < n_tiles; t += GMX_BLOCK) {
        uint64_t a_; float m_;
        if (PEER && r != rank) (void)shard_peer_row(land_own + gmx_peer_stats_at(tag, world, n_tiles, r, t), tag, a_, m_);   // (complete: pass 1 waited)
        else { a_ = agg[t]; m_ = tmax[t]; }
        const uint64_t G = gmx_tile_scale(a_, gmx_tile_exp(m_), K);
        sum += G;
        below += (r == rank && t < my_tile) ? G : 0ull;
      }
      sum = wave_sum_u64(sum);
      if (lane == 0) s_part[r][wave] = sum;
    }
  }
  below = wave_sum_u64(below);
  if (lane == 0) s_below[wave] = below;
  __syncthreads();
  // every WAVE derives the global total and this rank's CDF offset itself (lane s holds rank s's total: 4 LDS reads
  // and one 64-bit scan), so nothing below waits for the slot bounds; wave 0's lanes s < world evaluate those — rank
  // s's first slot, by the exact predicate — while the other waves are already at their sources' slot runs; the
  // barrier before the LDS fill publishes them.
  uint64_t rt = 0;
  if (lane < world) rt = (s_part[lane][0] + s_part[lane][1]) + (s_part[lane][2] + s_part[lane][3]);
  const uint64_t rscan = wave_scan_u64(rt);
  const uint64_t total = wave_last_u64(rscan);
  const uint64_t my_off = rscan - rt;
  const uint64_t cdf_offset = (uint64_t)shard_readlane64((int64_t)my_off, rank);
  if (wave == 0) {
    if (lane < world && lane > 0) {          // (rank 0's first slot is slot 0: nothing to evaluate at world 1)
      const double not_ = total ? (double)N / (double)total : 0.0;
      const double eps_ = (double)N * 0x1p-44 + 0x1p-40;
      s_bounds[lane] = total ? (int32_t)slots_below(kind, key, (uint64_t)u0_host, my_off, (uint64_t)N << 23, total, not_, eps_, (int64_t)N) : 0;
    }
    if (lane == 0) { s_bounds[0] = 0; s_bounds[world] = N; }
  }
  bool overflow = false;
  // PEER: ship source `src` (local index) as slot k of the block for rank d — one granule per leaf, into d's landing
  auto put = [&](int32_t d, int32_t k, uint32_t src) {
    if (PEER) {
      uint64_t* land_d = P.land[d];
      for (int l = 0; l < P.leaves; ++l)
        gmx_granule_put(land_d + gmx_peer_state_at(tag, world, n_tiles, cap, P.leaves, l, rank, k), P.state[l][src], tag);
    } else {
      send[(int64_t)d * cap + k] = state[src];
    }
  };
  // PEER: my slot whose ancestor rank s ships as element k: wait for its granule(s), store them in the local tail
  auto receive = [&](int32_t s_, int32_t k) {
    if (PEER) {
      for (int l = 0; l < P.leaves; ++l) {
        uint32_t v = 0u;
        if (timed_out || !gmx_granule_wait(land_own + gmx_peer_state_at(tag, world, n_tiles, cap, P.leaves, l, s_, k), tag, v)) {
          timed_out = true;
          __hip_atomic_store(P.status, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (TAGGED) gmx_store_u32_sc1(P.tail[l] + ((int64_t)s_ * cap + k), v);
        else P.tail[l][(int64_t)s_ * cap + k] = v;
      }
    }
  };
  if (total == 0) {       // no mass at all: the globally last particle (rank world-1, local n-1) sources every slot
    if (blockIdx.x == 0 && tid == 0) {
      plan[GMX_PLAN_TOTAL] = 0; plan[GMX_PLAN_OFFSET] = 0;
      for (int s = 0; s < world; ++s) plan[GMX_PLAN_BOUNDS + s] = 0;
      plan[GMX_PLAN_BOUNDS + world] = (int64_t)N;
      if (total_out) *total_out = 0ull;
      if (max_out) *max_out = M;
    }
    if (rank != world - 1) {            // my slot base + i arrives as element i of rank world-1's block
#pragma unroll
      for (int c = 0; c < CDF_VEC; ++c) {
        const int32_t i = i0 + c;
        if (i < n) {
          if (i < cap) { receive(world - 1, i); announce(); set_idx(i, (uint32_t)(n + (world - 1) * cap + i)); }
          else { set_idx(i, 0u); overflow = true; }
        }
      }
    }
    if (rank == world - 1) {
#pragma unroll
      for (int c = 0; c < CDF_VEC; ++c) {
        const int32_t k = i0 + c;
        if (k < n) {
          set_idx(k, (uint32_t)(n - 1));
          for (int d = 0; d < world - 1; ++d) {
            if (k < cap) put(d, k, (uint32_t)(n - 1)); else overflow = true;
          }
        }
      }
    }
    if (overflow) plan[GMX_PLAN_OVERFLOW] = 1;
    return;
  }
  // (a) the slot runs of my 4 sources
  const double n_over_total = (double)N / (double)total;
  const double eps = (double)N * 0x1p-44 + 0x1p-40;
  uint64_t wave_off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? s_scan[w] : 0ull;
  const uint64_t loc = wave_off + (inc - run);
  const uint64_t prefix = cdf_offset + ((s_below[0] + s_below[1]) + (s_below[2] + s_below[3]));
  uint64_t cv[CDF_VEC + 1];
  cv[0] = prefix + gmx_tile_scale(loc, k_b, K);
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) cv[c + 1] = (i0 + c < n) ? prefix + gmx_tile_scale(loc + q[c], k_b, K) : 0ull;
  // sources past the shard's end own no slot: their upper edge is the last real source's
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) cv[c + 1] = (i0 + c < n) ? cv[c + 1] : cv[c];
  int32_t e[CDF_VEC + 1];
  uint32_t near_bits = 0;
#pragma unroll
  for (int c = 1; c <= CDF_VEC; ++c) {
    const sb_est r = slots_below_est<kind>(key, u0_host, cv[c], total, n_over_total, eps, N);
    e[c] = r.j;
    near_bits |= r.near ? (1u << c) : 0u;
  }
  {
    const sb_est r = slots_below_est<kind>(key, u0_host, cv[0], total, n_over_total, eps, N);
    near_bits |= (lane == 0 && r.near) ? 1u : 0u;
    e[0] = (int32_t)wave_shr1_u32((uint32_t)e[CDF_VEC], (uint32_t)r.j);
    if (lane == 0) e[0] = r.j;
  }
  if (__any(near_bits != 0u)) {          // cold: the exact integer predicate for the flagged evaluations
    const uint64_t D = (uint64_t)N << 23;
    int32_t fixed0 = e[0];
#pragma unroll 1
    for (int c = 0; c <= CDF_VEC; ++c) {
      if (near_bits & (1u << c)) {
        const uint64_t cc = c == 0 ? cv[0] : c == 1 ? cv[1] : c == 2 ? cv[2] : c == 3 ? cv[3] : cv[4];
        const int32_t j0 = c == 0 ? e[0] : c == 1 ? e[1] : c == 2 ? e[2] : c == 3 ? e[3] : e[4];
        const int32_t j = (int32_t)slots_below_exact(kind, key, (uint64_t)u0_host, cc, D, total, (int64_t)j0, (int64_t)N);
        if (c == 0) fixed0 = j; else if (c == 1) e[1] = j; else if (c == 2) e[2] = j; else if (c == 3) e[3] = j; else e[4] = j;
      }
    }
    const uint32_t up = wave_shr1_u32((uint32_t)e[CDF_VEC], (uint32_t)fixed0);
    e[0] = (lane == 0) ? fixed0 : (int32_t)up;
  }
  // ---- route through LDS ----
  const int32_t e4 = e[CDF_VEC];
  if (tid == 0) s_rng[0] = e[0];
  if (tid == GMX_BLOCK - 1) s_rng[1] = e4;
  reinterpret_cast<uint4*>(s_mark)[tid] = make_uint4(0u, 0u, 0u, 0u);
  reinterpret_cast<uint4*>(s_mark)[tid + GMX_BLOCK] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int32_t T0 = __builtin_amdgcn_readfirstlane(s_rng[0]), T1 = __builtin_amdgcn_readfirstlane(s_rng[1]);
  const int32_t S = s_bounds[rank];          // the first slot whose ancestor lives on this rank
  if (blockIdx.x == 0 && tid == 0) {       // published for inspection / tests; the overflow word is left alone
    plan[GMX_PLAN_TOTAL] = (int64_t)total; plan[GMX_PLAN_OFFSET] = (int64_t)cdf_offset;
    for (int s = 0; s <= world; ++s) plan[GMX_PLAN_BOUNDS + s] = (int64_t)s_bounds[s];
    if (total_out) *total_out = total;
    if (max_out) *max_out = M;
  }
  // (b) my slots base + i0..+3: an ancestor on rank s != rank arrives at recv[s*cap + k]  (PEER: at the END of the
  // kernel, behind this workgroup's own puts — it WAITS for the value there)
  auto remote_slots = [&]() {
    if (world > 1) {
#pragma unroll
      for (int c = 0; c < CDF_VEC; ++c) {
        const int32_t i = i0 + c;
        if (i < n) {
          const int32_t jj = base + i;
          int s = 0;
          while (s + 1 < world && s_bounds[s + 1] <= jj) ++s;
          if (s != rank) {
            const int32_t first = s_bounds[s] > base ? s_bounds[s] : base;
            const int32_t k = jj - first;
            if (k < cap) { receive(s, k); announce(); set_idx(i, (uint32_t)(n + s * cap + k)); }
            else { set_idx(i, 0u); overflow = true; }
          }
        }
      }
    }
  };
  if (!PEER) remote_slots();
  for (int32_t pass = T0; pass < T1; pass += RS_FILL_SLOTS) {      // block-uniform (1 pass unless the tile owns > 2048 slots)
    if (pass != T0) {
      __syncthreads();
      reinterpret_cast<uint4*>(s_mark)[tid] = make_uint4(0u, 0u, 0u, 0u);
      reinterpret_cast<uint4*>(s_mark)[tid + GMX_BLOCK] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) {
      const int32_t lo = e[c] > pass ? e[c] : pass;
      if (e[c + 1] > lo && lo - pass < RS_FILL_SLOTS) s_mark[lo - pass] = (uint32_t)(i0 + c);
    }
    __syncthreads();
    uint4 a = reinterpret_cast<const uint4*>(s_mark)[2 * tid];
    uint4 b = reinterpret_cast<const uint4*>(s_mark)[2 * tid + 1];
    a.y = a.y > a.x ? a.y : a.x; a.z = a.z > a.y ? a.z : a.y; a.w = a.w > a.z ? a.w : a.z;
    b.x = b.x > a.w ? b.x : a.w; b.y = b.y > b.x ? b.y : b.x; b.z = b.z > b.y ? b.z : b.y; b.w = b.w > b.z ? b.w : b.z;
    const uint32_t incl = gmx_wave_umax_scan(b.w);
    uint32_t carry = wave_shr1_u32(incl, 0u);
    if (lane == 63) s_carry[wave] = incl;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; ++w) { const uint32_t v = s_carry[w]; carry = (w < wave && v > carry) ? v : carry; }
    uint32_t src[8];
    src[0] = a.x > carry ? a.x : carry; src[1] = a.y > carry ? a.y : carry; src[2] = a.z > carry ? a.z : carry;
    src[3] = a.w > carry ? a.w : carry; src[4] = b.x > carry ? b.x : carry; src[5] = b.y > carry ? b.y : carry;
    src[6] = b.z > carry ? b.z : carry; src[7] = b.w > carry ? b.w : carry;
    const int32_t j = pass + 8 * tid;
    if (j < T1) {
      int32_t d = 0;                                              // owner of slot j: j / n
      if (SMALL) { for (int s = 1; s < world; ++s) d += (j >= s * n) ? 1 : 0; }      // (world <= 8: no integer division)
      else d = (int32_t)((uint32_t)j / (uint32_t)n);
      int32_t d_end = (d + 1) * n;
      if (d == rank && j + 8 <= T1 && j + 8 <= d_end) {            // the common case: 8 slots of my own shard
        rs_u32x4_a4 va, vb;
        va.x = src[0]; va.y = src[1]; va.z = src[2]; va.w = src[3]; vb.x = src[4]; vb.y = src[5]; vb.z = src[6]; vb.w = src[7];
        if (TAGGED) {
          uint32_t* w = reinterpret_cast<uint32_t*>(next_idx) + (j - base);
          gmx_store_u32x4_sc1(w, va.x | tagw, va.y | tagw, va.z | tagw, va.w | tagw);        // (4-byte alignment suffices)
          gmx_store_u32x4_sc1(w + 4, vb.x | tagw, vb.y | tagw, vb.z | tagw, vb.w | tagw);
        } else {
          *reinterpret_cast<rs_u32x4_a4*>(next_idx + (j - base)) = va;
          *reinterpret_cast<rs_u32x4_a4*>(next_idx + (j - base) + 4) = vb;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int32_t jj = j + c;
          if (jj < T1) {
            if (jj >= d_end) { ++d; d_end += n; }
            if (d == rank) set_idx(jj - base, src[c]);
            else {
              const int32_t first = S > d * n ? S : d * n;        // first slot this rank sends to d
              const int32_t k = jj - first;
              if (k < cap) put(d, k, src[c]); else overflow = true;
            }
          }
        }
      }
    }
  }
  if (PEER) remote_slots();
  if (overflow) plan[GMX_PLAN_OVERFLOW] = 1;
}
