// gmx_jit.h — device side of a specialised site program.
//
// gmx_program_specialize() emits a translation unit of the form
//
//   #include "gmx_jit.h"
//   __device__ static constexpr uint32_t GMX_JIT_CONST[] = { c0, c1, ... };
//   GMX_JIT_BEGIN(N_REGS, FULL, N_DYN)
//     GMX_JIT_OP(w0, w1)          // one line per instruction
//     ...
//   GMX_JIT_END
//
// i.e. the SAME gmx_vm_step template (gmx_vm.h) the interpreter loops over,
// instantiated once per instruction with compile-time instruction words
// (gmx_cword): the dispatch switch, the operand selection and the register
// indices are constant expressions, so what is compiled is straight-line code
// calling the hand-written samplers / log-densities — bit-identical to the
// interpreter by construction (same source, same -ffp-contract=off).
#pragma once
#include "gmx_block.h"
#include "gmx_vm.h"
#include "gmx_peer.h"
#if defined(GMX_JIT_RS)      /* gmx_program_set_fuse_resample: the kernel can resample the previous step first */
#include "gmx_offspring.h"
#endif
#if defined(GMX_JIT_SH)      /* gmx_program_set_fuse_shard_step: ... or route the previous step of a sharded sweep first */
#include "gmx_shard_fill.h"
#endif

// a background program (gmx_program_set_background) keeps the default wave priority 0 and has a name of its own
// (so that kernel traces tell the noise programs from the chain's site programs)
#if defined(GMX_JIT_BACKGROUND)
#define GMX_JIT_PRIO
#define GMX_JIT_NAME gmx_jit_background_kernel
#else
#define GMX_JIT_PRIO GMX_SETPRIO
#define GMX_JIT_NAME gmx_jit_kernel
#endif

// A LOOPED launch (GMX_JIT_RS_LOOP, gmx_program_set_fuse_resample_loop): more 1024-particle tiles than the device holds
// workgroups at once — a bootstrap step of more than 2^20 particles in ONE launch.  The grid is min(tiles, 1024)
// workgroups; workgroup b handles tiles b, b + G, b + 2 G, ... in TWO passes: first the resampling of the previous step
// for every one of its tiles (after ONE pass over the statistics table: gmx_tile_table_pass) — nothing there waits for
// another workgroup — then, tile by tile, the wait for its own particles' ancestor words, the gather and the site program.
// Every word a workgroup waits for is written by the FIRST pass of a workgroup of the same launch, and first passes wait
// for nothing: residency is needed for G workgroups only, whatever n.
#if defined(GMX_JIT_RS_LOOP)
#define GMX_JIT_BLK gmx_blk
#define GMX_RS_LOOP_MAX 16             /* tiles per workgroup: 16 x 1024 workgroups x 1024 particles = 2^24 */
#else
#define GMX_JIT_BLK blockIdx.x
#endif

template <int NDYN, int PPV>
struct gmx_jit_ctx {
  const uint32_t* consts;   // constexpr constants (pool entries NDYN..)
  const gmx_run_args* A;
  float* lds4;
  uint64_t* lds8;           // 4 x u64 scratch (tile statistics)
  float red_x[PPV];         // OP_REDMAX operands of the thread's particles (-inf when inactive)
  int cur;                  // which of the thread's particles this step works on
  uint32_t blk;             // the 1024-particle tile / block-partial row this workgroup works on (blockIdx.x, or a looped launch's tile)
  uint32_t part;            // index of the 256-particle group this step works on (block partial row)
  uint32_t rows;            // number of 256-particle groups = ceil(n / 256)
  int64_t n_rows;           // n (particles of this launch)
  float acc_max;            // OP_REDMAX: running max over the thread's PP particles
  uint64_t* peer_land;      // gmx_run_args.peer: thread p < world (p != rank): rank p's landing block; else null
  uint32_t peer_tag;        // ... and this step's tag (loaded at the top of the kernel: the epilogue must not wait for it)
  bool first, last;         // this step handles the first / last of the thread's particles
  __device__ __forceinline__ uint32_t pool(uint32_t i) const {
    return i < (uint32_t)NDYN ? A->uni[i] : consts[i - (uint32_t)NDYN];
  }
  __device__ __forceinline__ const void* in_ptr(uint32_t s) const { return A->in_d[s]; }
  __device__ __forceinline__ void* out_ptr(uint32_t s) const { return A->out_d[s]; }
  __device__ __forceinline__ const void* tab_ptr(uint32_t s) const { return A->tab_d[s]; }
  // one block reduction and ONE partial row per workgroup (a max does not care how particles are grouped)
  // With 4 particles per thread the workgroup IS one 1024-particle tile of the two-level CDF
  // (include/genmi.h "Resampling"): when the caller asks (tile_agg_d), also write
  // A_b = sum floor(exp(x - k_b ln 2) * 2^tile_shift), k_b = ceil(block max / ln 2), so no separate pass
  // reads the log-weights again.
  __device__ __forceinline__ void red_max(float x, bool active) {
    const float m = active ? x : -gmx_inf();
    red_x[cur] = m;
    acc_max = first ? m : gmx_rmax(acc_max, m);
    if (last) {
      const float bm = block_max(acc_max, lds4);
      if (threadIdx.x == 0 && A->red_out_d) A->red_out_d[blk] = bm;
      if (PPV == 4 && A->tile_agg_d) {
        const float ref = gmx_tile_ref(gmx_tile_exp(bm));
        uint64_t s = 0;
#pragma unroll
        for (int p = 0; p < PPV; ++p) s += gmx_exp_fixed(red_x[p] - ref, A->tile_shift);   // inactive: red_x = -inf, weight 0
        s = wave_sum_u64(s);
        if ((threadIdx.x & 63) == 0) lds8[threadIdx.x >> 6] = s;
        __syncthreads();
        const uint64_t a_b = (lds8[0] + lds8[1]) + (lds8[2] + lds8[3]);
        if (threadIdx.x == 0) A->tile_agg_d[blk] = a_b;
        // sharded: the same two numbers straight into every other rank's landing table (thread p serves rank p)
        if (peer_land) gmx_peer_put_tile(peer_land, peer_tag, A->peer.world, A->peer.tiles, A->peer.rank, (int)blk, a_b, bm);
      }
    }
  }
  __device__ __forceinline__ void red_lse(float x, bool active) {
    gmx_red_lse(A->red_out_d, lds4, part, rows, x, active);
  }
};

// PP particles per thread: particle p of a thread is row (blockIdx * PP + p) * 256 + threadIdx, so every
// load / store stays a coalesced 256-particle group and the block partial rows are the ones the
// interpreter would write.  (Four waves per SIMD already interleave four dependent chains; what PP buys
// is fewer, fatter workgroups — a workgroup is one 1024-particle tile of the resampler's CDF.)
//
// Input loads are PREFETCHED (gmx_program_specialize emits GMX_JIT_PRE* for the program's OP_LDIN
// instructions): all of a thread's loads are issued at the top of the kernel, unconditionally, on row indices
// clamped into [0, n) — hipcc will not hoist a load out of an `if (active)` region, and with the loads at
// their first use every one of them was followed by its own s_waitcnt (eight dependent round trips for the
// bootstrap step: ancestor, then state, for each of four particles).  Gathered inputs go in two stages:
// the ancestors at the top, the rows they name after the first key derivation (one Threefry block per
// particle hides the first round trip); the second round trip is hidden by the rest of the RNG work.
// (the routing prologue of a sharded step is register-hungry: held to 64 VGPRs it spilled 168 bytes per lane to scratch
//  — measured 44 us/step; four waves per SIMD = 128 VGPRs are what a launch with every workgroup resident needs anyway)
#if defined(GMX_JIT_SH)
#define GMX_JIT_OCC __attribute__((amdgpu_waves_per_eu(1, 4)))
#else
#define GMX_JIT_OCC
#endif
#if defined(GMX_JIT_RS_LOOP)
// first pass: the resampling of step t - 1 for every tile of this workgroup; then the loop over its tiles opens
#define GMX_JIT_LOOP_OPEN                                                                        \
    const int gmx_tiles = (int)((n + (int64_t)(PP * GMX_BLOCK) - 1) / (int64_t)(PP * GMX_BLOCK)); \
    const int gmx_G = (int)gridDim.x;                                                            \
    const int gmx_C = (gmx_tiles + gmx_G - 1) / gmx_G;                                           \
    if (PP == 4 && A.rs.lw_d) {                                                                  \
      __shared__ uint64_t gmx_pf[GMX_RS_LOOP_MAX];                                               \
      uint64_t gmx_tot, gmx_mk;                                                                  \
      gmx_tile_table_pass(A.rs.tile_max_d, A.rs.tile_agg_d, gmx_tiles, (int)blockIdx.x, gmx_G, gmx_C, gmx_pf, gmx_tot, gmx_mk); \
      GMX_JIT_NOUNROLL for (int gmx_c = 0; gmx_c < gmx_C; ++gmx_c) {                             \
        const int gmx_tile = (int)blockIdx.x + gmx_c * gmx_G;                                    \
        if (gmx_tile < gmx_tiles)                                                                \
          gmx_offspring_tile_body<GMX_RESAMPLE_SYSTEMATIC, 0, true>(A.rs.key0, A.rs.key1, A.rs.u0, A.rs.lw_d, A.rs.tile_max_d, \
              (const uint64_t*)nullptr, n, gmx_tiles, gmx_pow2i(A.rs.shift), A.rs.max_out_d, A.rs.total_out_d, \
              const_cast<int32_t*>(A.ancestors_d), nullptr, A.rs.tag, gmx_tile, gmx_pf[gmx_c], gmx_tot, gmx_mk); \
        __syncthreads();                                                                         \
      }                                                                                          \
    }                                                                                            \
    GMX_JIT_NOUNROLL for (int gmx_lc = 0; gmx_lc < gmx_C; ++gmx_lc) {                            \
      const uint32_t gmx_blk = blockIdx.x + (uint32_t)gmx_lc * (uint32_t)gmx_G;                  \
      if ((int)gmx_blk >= gmx_tiles) break;                /* block-uniform */                   \
      if (gmx_lc) __syncthreads();
#define GMX_JIT_LOOP_CLOSE }
#else
#define GMX_JIT_LOOP_OPEN
#define GMX_JIT_LOOP_CLOSE
#endif
#define GMX_JIT_BEGIN(NREGS, FULLV, NDYN, PPV, NPRE)                                             \
  extern "C" __global__ void __launch_bounds__(GMX_BLOCK) GMX_JIT_OCC GMX_JIT_NAME(int64_t n, const gmx_run_args A) { \
 GMX_JIT_PRIO                                                                                 \
    __shared__ float lds4[4];                                                                    \
    __shared__ uint64_t lds8[4];                                                                 \
    constexpr int PP = PPV;                                                                      \
    typedef gmx_regs_vgpr<NREGS> regs_t;                                                         \
    typedef gmx_jit_ctx<NDYN, PPV> ctx_t;                                                        \
    constexpr bool full_v = FULLV;                                                               \
    ctx_t ctx;                                                                                   \
    ctx.consts = GMX_JIT_CONST; ctx.A = &A; ctx.lds4 = lds4; ctx.lds8 = lds8; ctx.part = 0; ctx.cur = 0; \
    ctx.rows = (uint32_t)((n + GMX_BLOCK - 1) / GMX_BLOCK); ctx.n_rows = n;      \
    ctx.peer_land = nullptr; ctx.peer_tag = 0u;                                                  \
    if (PPV == 4 && A.peer.land_d) {                    /* launch-uniform */                     \
      ctx.peer_tag = *A.peer.tag_base_d + (uint32_t)A.peer.step;                                 \
      if (threadIdx.x < (uint32_t)A.peer.world && (int)threadIdx.x != A.peer.rank)               \
        ctx.peer_land = (uint64_t*)A.peer.land_d[threadIdx.x];                                   \
    }                                                                                            \
    GMX_JIT_LOOP_OPEN                                                                            \
    ctx.blk = GMX_JIT_BLK;                                                                       \
    regs_t R[PP];                                                                                \
    /* particle rows as 32-bit numbers (gmx_program_run admits n < 2^31 for a specialised kernel): the 64-bit \
       forms below are zero-extensions, so address arithmetic is a shift-add, not 64-bit compares / selects */  \
    const uint32_t n32 = (uint32_t)n;                                                            \
    int64_t idx[PP];                                                                             \
    uint32_t cidx[PP];         /* idx clamped into [0, n): a valid row for prefetches of inactive lanes */ \
    uint32_t arow[PP];         /* ancestors[cidx] */                                             \
    uint32_t pre[(NPRE) > 0 ? (NPRE) : 1][PP];                                                   \
    bool act[PP];                                                                                \
    uint32_t gmx_t = 0u;       /* iteration number of the innermost enclosing GMX_JIT_LOOP (0 outside) */ \
    uint32_t gmx_t0 = 0u, gmx_tf = 0u;   /* the outer loop's; the row-major index over both (GMX_F_FLAT) */ \
    /* a 2-D launch (gmx_program_run, GMX_KEY_ROWSPLIT background programs): n particles per ROW, blockIdx.y \
       is the row — row r's particles are rows r * n .. r * n + n - 1 of every leaf */           \
    const uint32_t row0 = blockIdx.y * n32;                                                      \
    _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                             \
      R[p].init();                                                                               \
      const uint32_t i32 = (GMX_JIT_BLK * (uint32_t)PP + (uint32_t)p) * (uint32_t)GMX_BLOCK + threadIdx.x; \
      idx[p] = (int64_t)(row0 + i32);                                                            \
      act[p] = i32 < n32;                                                                        \
      cidx[p] = row0 + (act[p] ? i32 : n32 - 1u);                                                \
      arow[p] = 0u;                                                                              \
    }                                                                                            \
    (void)cidx; (void)arow; (void)pre; (void)gmx_t; (void)gmx_t0; (void)gmx_tf;

// the ancestors of the thread's particles: loaded — or, for a fused bootstrap step (gmx_run_args.rs), WRITTEN first
// (this workgroup's tile of the previous step's resampling: gmx_offspring.h) and then polled until the words of this
// workgroup's own particles carry the step's tag (workgroup-uniform branch)
// the ancestor words of the thread's own particles, written by workgroups of THIS launch as {tag | index}: polled until
// they carry the launch's tag (bounded by the wall clock: 2 s, then the sticky status word), then clamped below LIMIT
#define GMX_JIT_POLL_ANC(TAG, STATUS, LIMIT)                                                     \
      const uint32_t* gmx_aw = reinterpret_cast<const uint32_t*>(A.ancestors_d);                 \
      uint32_t gmx_av[PP];                                                                       \
      bool gmx_ok = true;                                                                        \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                           \
        gmx_av[p] = __hip_atomic_load(gmx_aw + cidx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        gmx_ok &= (gmx_av[p] >> GMX_ANC_TAG_SHIFT) == (TAG);                                     \
      }                                                                                          \
      if (!gmx_ok) {                                                                             \
        const uint64_t gmx_t0 = wall_clock64();                                                  \
        do {                                                                                     \
          __builtin_amdgcn_s_sleep(1);                                                           \
          gmx_ok = true;                                                                         \
          _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                       \
            gmx_av[p] = __hip_atomic_load(gmx_aw + cidx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            gmx_ok &= (gmx_av[p] >> GMX_ANC_TAG_SHIFT) == (TAG);                                 \
          }                                                                                      \
        } while (!gmx_ok && wall_clock64() - gmx_t0 < 200000000ull);                             \
        if (!gmx_ok) __hip_atomic_store((STATUS), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      }                                                                                          \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                           \
        const uint32_t gmx_i = gmx_av[p] & GMX_ANC_INDEX_MASK;                                   \
        arow[p] = gmx_i < (LIMIT) ? gmx_i : (LIMIT) - 1u;                                        \
      }

#if defined(GMX_JIT_SH)
// a SHARDED sweep's step: this workgroup's tile of the routing of step t - 1 first (gmx_run_args.sh; gmx_shard_fill.h),
// then the ancestor words of its own particles — indices into the extended state [ n local | world * capacity received ]
#define GMX_JIT_PRE_ANC                                                                          \
    if (PP == 4 && A.sh.lw_d) {                                                                  \
      shard_peer gmx_sp;                                                                         \
      gmx_sp.land = (uint64_t* const*)A.sh.peer.land_d; gmx_sp.tag_base = A.sh.peer.tag_base_d;  \
      gmx_sp.status = A.sh.peer.status_d; gmx_sp.step = A.sh.peer.step; gmx_sp.leaves = A.sh.peer.leaves; \
      gmx_sp.state = (const uint32_t* const*)A.sh.state_d; gmx_sp.tail = (uint32_t* const*)A.sh.tail_d;  /* kernel-argument memory */ \
      gmx_shard_fill_body<GMX_RESAMPLE_SYSTEMATIC, true, true, true>(                            \
          A.sh.key0, A.sh.key1, A.sh.u0, A.sh.lw_d, (const uint8_t*)A.sh.stats_own_d, (size_t)0, A.sh.peer.tiles, \
          gmx_pow2i(A.sh.shift), A.sh.peer.rank, A.sh.peer.world, (int32_t)n, (int32_t)A.sh.peer.capacity, A.sh.plan_d, \
          A.sh.total_out_d, A.sh.max_out_d, (const uint32_t*)nullptr, (uint32_t*)nullptr,         \
          const_cast<int32_t*>(A.ancestors_d), gmx_sp, A.sh.tag);                                 \
      const uint32_t gmx_lim = n32 + (uint32_t)A.sh.peer.world * (uint32_t)A.sh.peer.capacity;   \
      GMX_JIT_POLL_ANC(A.sh.tag, A.sh.status_d, gmx_lim)                                         \
      /* (no fence: a row in the tail was stored write-through BEFORE its word, the gather's load depends on the word) */ \
    } else {                                                                                     \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) arow[p] = (uint32_t)A.ancestors_d[cidx[p]]; \
    }
#elif defined(GMX_JIT_RS_LOOP)
#define GMX_JIT_PRE_ANC                                                                          \
    if (PP == 4 && A.rs.lw_d) {                                                                  \
      GMX_JIT_POLL_ANC(A.rs.tag, A.rs.status_d, n32)                                             \
    } else {                                                                                     \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) arow[p] = (uint32_t)A.ancestors_d[cidx[p]]; \
    }
#elif defined(GMX_JIT_RS)
#define GMX_JIT_PRE_ANC                                                                          \
    if (PP == 4 && A.rs.lw_d) {                                                                  \
      gmx_offspring_tile_body<GMX_RESAMPLE_SYSTEMATIC, 4, true>(A.rs.key0, A.rs.key1, A.rs.u0, A.rs.lw_d, A.rs.tile_max_d, \
          A.rs.tile_agg_d, n, (int)gridDim.x, gmx_pow2i(A.rs.shift), A.rs.max_out_d, A.rs.total_out_d,          \
          const_cast<int32_t*>(A.ancestors_d), nullptr, A.rs.tag);                                \
      GMX_JIT_POLL_ANC(A.rs.tag, A.rs.status_d, n32)                                             \
    } else {                                                                                     \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) arow[p] = (uint32_t)A.ancestors_d[cidx[p]]; \
    }
#else
#define GMX_JIT_PRE_ANC                                                                          \
    _Pragma("unroll") for (int p = 0; p < PP; ++p) arow[p] = (uint32_t)A.ancestors_d[cidx[p]];
#endif

#define GMX_JIT_PRE_LOAD(K, SLOT, U8, ROW)                                                       \
    _Pragma("unroll") for (int p = 0; p < PP; ++p)                                               \
      pre[K][p] = (U8) ? (uint32_t)((const uint8_t*)A.in_d[SLOT])[ROW[p]] : ((const uint32_t*)A.in_d[SLOT])[ROW[p]];
#define GMX_JIT_PRE(K, SLOT, U8) GMX_JIT_PRE_LOAD(K, SLOT, U8, cidx)
#define GMX_JIT_PRE_G(K, SLOT, U8) GMX_JIT_PRE_LOAD(K, SLOT, U8, arow)
#define GMX_JIT_FENCE __builtin_amdgcn_sched_barrier(0);

// OP_LDIN whose value was prefetched
#define GMX_JIT_LDPRE(DST, K)                                                                    \
    _Pragma("unroll") for (int p = 0; p < PP; ++p) R[p].set(DST, pre[K][p]);

#define GMX_JIT_OP(W0, W1)                                                                       \
    _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                             \
      ctx.part = GMX_JIT_BLK * PP + p; ctx.first = (p == 0); ctx.last = (p == PP - 1); ctx.cur = p; \
      gmx_vm_step<regs_t, full_v, gmx_cword<W0, W1>, ctx_t>(R[p], gmx_cword<W0, W1>(), idx[p], act[p], A, ctx, gmx_t, gmx_tf); \
    }

// OP_LOOP / OP_ENDLOOP: a counted loop around the instructions in between (launch-uniform trip count)
// (no unrolling of the counted loops: hiprtc on a GPU box — not the same build offline — unrolled a four-instruction copy
//  loop inside a plate's loop by four and assigned the address of the fourth store to v[10:11] while v10 still held the
//  loop-invariant row offset the next group of four indexes the table with: a wild load, a memory fault
//  (profiles/r05z_jit_miscompile.txt has the disassembly rocgdb took from the faulting wave).  Rolled, the loops are
//  the code the interpreter runs, instruction for instruction.)
#define GMX_JIT_NOUNROLL _Pragma("clang loop unroll(disable)")
#define GMX_JIT_LOOP(COUNT) GMX_JIT_NOUNROLL for (gmx_t0 = 0u; gmx_t0 < (COUNT); ++gmx_t0) { gmx_t = gmx_t0; gmx_tf = gmx_t0;
#define GMX_JIT_ENDLOOP } gmx_t = 0u; gmx_t0 = 0u; gmx_tf = 0u;
// a loop INSIDE a GMX_JIT_LOOP (a long scan inside a large plate): gmx_t counts the inner iterations, gmx_tf the pairs
#define GMX_JIT_LOOP2(COUNT) GMX_JIT_NOUNROLL for (uint32_t gmx_t1 = 0u; gmx_t1 < (COUNT); ++gmx_t1) { gmx_t = gmx_t1; gmx_tf = gmx_t0 * (COUNT) + gmx_t1;
#define GMX_JIT_ENDLOOP2 } gmx_t = gmx_t0; gmx_tf = gmx_t0;
// a third level (a plate of plates of plates): gmx_tf runs over the triples, row-major; inside a GMX_JIT_LOOP2 only
#define GMX_JIT_LOOP3(COUNT) { const uint32_t gmx_tp = gmx_t, gmx_tfp = gmx_tf;                  \
    GMX_JIT_NOUNROLL for (uint32_t gmx_t2 = 0u; gmx_t2 < (COUNT); ++gmx_t2) { gmx_t = gmx_t2; gmx_tf = gmx_tfp * (COUNT) + gmx_t2;
#define GMX_JIT_ENDLOOP3 } gmx_t = gmx_tp; gmx_tf = gmx_tfp; }

#define GMX_JIT_END GMX_JIT_LOOP_CLOSE }
