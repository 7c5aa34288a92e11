// gmx_peer.h — the granules of the fused peer exchange (include/genmi.h "Fused peer exchange"), shared by the
// hiprtc-specialised site programs (the statistics put of their epilogue), the AOT kernels (k_shard_step_peer,
// k_peer_put_stats) and the tests' CPU mirror (layout only).
//
// A granule is one naturally aligned 8-byte word {data: low 32 bits, tag: high 32 bits}.  It is written by ONE
// system-scope relaxed atomic store (`global_store_dwordx2 ... sc0 sc1`: write-through, nothing left dirty in the
// writer's L2) and read by system-scope relaxed atomic loads (`global_load_dwordx2 ... sc0 sc1`: past the caches) until
// its tag is the step's.  8-byte single-copy atomicity is all the protocol needs: no flag, no counter, no fence.
#pragma once
#include "gmx_math.h"
#include "genmi.h"

#define GMX_PEER_TIMEOUT_TICKS 400000000ull  /* a wait gives up after 4 s of the 100 MHz wall clock (status_d[0] = 1): a
                                               peer that never arrives must not hang the GPU; ranks are microseconds
                                               apart inside a sweep and a host barrier apart before its first launch */

// ---- layout of a rank's landing block, in u64 words ----
GMX_HD size_t gmx_peer_stats_words(int world, int tiles) { return (size_t)2 * (size_t)world * (size_t)tiles * 3u; }
GMX_HD size_t gmx_peer_state_words(int world, int64_t cap, int leaves) {
  return (size_t)2 * (size_t)leaves * (size_t)world * (size_t)cap;
}
// the three granules of (source rank s, tile b) in half `tag & 1`
GMX_HD size_t gmx_peer_stats_at(uint32_t tag, int world, int tiles, int s, int b) {
  return ((((size_t)(tag & 1u) * (size_t)world + (size_t)s) * (size_t)tiles) + (size_t)b) * 3u;
}
// granule k of the block source rank s ships here, leaf l, half `tag & 1`
GMX_HD size_t gmx_peer_state_at(uint32_t tag, int world, int tiles, int64_t cap, int leaves, int l, int s, int64_t k) {
  return gmx_peer_stats_words(world, tiles) +
         ((((size_t)(tag & 1u) * (size_t)leaves + (size_t)l) * (size_t)world + (size_t)s) * (size_t)cap) + (size_t)k;
}
GMX_HD uint64_t gmx_granule(uint32_t data, uint32_t tag) { return (uint64_t)data | ((uint64_t)tag << 32); }

#if defined(__HIPCC__)
__device__ __forceinline__ void gmx_granule_put(uint64_t* p, uint32_t data, uint32_t tag) {
  __hip_atomic_store(p, gmx_granule(data, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t gmx_granule_peek(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin (bounded by the wall clock) until granule *p carries `tag`; false = gave up
__device__ __forceinline__ bool gmx_granule_wait(const uint64_t* p, uint32_t tag, uint32_t& data) {
  uint64_t v = gmx_granule_peek(p);
  if ((uint32_t)(v >> 32) != tag) {
    const uint64_t t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(2);
      v = gmx_granule_peek(p);
      if ((uint32_t)(v >> 32) == tag) break;
    } while (wall_clock64() - t0 < GMX_PEER_TIMEOUT_TICKS);
  }
  data = (uint32_t)v;
  return (uint32_t)(v >> 32) == tag;
}
// the tile's statistics into a peer's landing block: three granules, in any order
__device__ __forceinline__ void gmx_peer_put_tile(uint64_t* land, uint32_t tag, int world, int tiles, int src_rank, int tile,
                                                  uint64_t agg, float tmax) {
  uint64_t* row = land + gmx_peer_stats_at(tag, world, tiles, src_rank, tile);
  gmx_granule_put(row + 0, (uint32_t)agg, tag);
  gmx_granule_put(row + 1, (uint32_t)(agg >> 32), tag);
  gmx_granule_put(row + 2, gmx_f2u(tmax), tag);
}
#endif
