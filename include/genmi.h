/* genmi.h — C-ABI of libgenmi_hip.so, the MI355X (gfx950) implementation of
 * GenJAX's vectorised inference hot path.
 *
 * The reference (genjax-dev/genjax @ 2025-07-18) is pure Python over
 * jax 0.5.2 + tensorflow-probability 0.23 and has no FFI boundary of its own
 * (SURVEY.md §8b).  The entry points below are what a `jax.ffi` / ctypes
 * binding for this path would bind; each cites the reference code it
 * replaces (paths relative to the reference tree).  INTEGRATION.md shows the
 * reference-side stubs.
 *
 * Conventions
 *  - plain pointers and sizes only; `_d` = device pointer, `_h` = host pointer;
 *  - every launch takes a stream (`hipStream_t` passed as void*), is
 *    asynchronous, allocates nothing and never synchronises, so a caller may
 *    capture any sequence of calls into a hipGraph;
 *  - return 0 on success, non-zero on error (text via gmx_last_error());
 *  - purely functional: inputs are never written; the caller owns all buffers;
 *  - the library keeps no mutable global state besides the per-thread error
 *    string.
 */
#ifndef GENMI_H
#define GENMI_H

#if defined(__HIPCC_RTC__)   /* compiled by hiprtc (site-program specialisation): no libc headers */
typedef __UINT8_TYPE__ uint8_t;
typedef __UINT16_TYPE__ uint16_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __INT32_TYPE__ int32_t;
typedef __INT64_TYPE__ int64_t;
typedef __SIZE_TYPE__ size_t;
typedef __UINTPTR_TYPE__ uintptr_t;
#else
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GMX_ABI_VERSION 8

typedef void* gmx_stream;            /* hipStream_t */
typedef struct gmx_program gmx_program;

int gmx_version(void);
/* sizeof(gmx_run_args) as this library was built: a binding whose own layout of the struct differs must refuse to
 * launch (a stale binding would hand the kernels shifted pointers). */
size_t gmx_run_args_bytes(void);
const char* gmx_last_error(void);

/* ------------------------------------------------------------------------
 * PRNG key algebra (Threefry-2x32, jax `threefry_partitionable` semantics).
 * Replaces jax.random.split / fold_in at
 *   src/genjax/_src/inference/smc.py:154,171,191,255,269,299-300,318-319,386
 *   src/genjax/_src/generative_functions/static.py:261,350,420,525,634
 *   src/genjax/_src/generative_functions/combinators/vmap.py:186,201
 * ---------------------------------------------------------------------- */

/* Host-side single block, for key bookkeeping and known-answer tests. */
void gmx_threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                           uint32_t out[2]);

/* out_keys_d[i] = split(key, n)[index_offset + i], i in [0, n). */
int gmx_split(const uint32_t key[2], int64_t n, int64_t index_offset,
              uint32_t* out_keys_d /* [n,2] */, gmx_stream stream);
/* out_keys_d[r*inner + j] = split(keys_d[r], inner)[j]. */
int gmx_split_rows(const uint32_t* keys_d /* [rows,2] */, int64_t rows,
                   int64_t inner, uint32_t* out_keys_d /* [rows*inner,2] */,
                   gmx_stream stream);
/* out_d[i] = fold_in(keys_d[i], data). */
int gmx_fold_in(const uint32_t* keys_d /* [n,2] */, uint32_t data, int64_t n,
                uint32_t* out_d /* [n,2] */, gmx_stream stream);
/* raw 32-bit draws: out_d[i*m + j] = random_bits(keys_d[i], 32, (m,))[j]. */
int gmx_random_bits(const uint32_t* keys_d /* [n,2] */, int64_t n, int64_t m,
                    uint32_t* out_d /* [n*m] */, gmx_stream stream);

/* ------------------------------------------------------------------------
 * Site programs: one fused kernel per generative-function-interface call.
 *
 * A site program is the straight-line result of running a `@gen` function's
 * Python source once with symbolic values (the job of `stage` +
 * `StatefulInterpreter`, src/genjax/_src/core/compiler/staging.py:286-298,
 * interpreters/stateful.py:47-86): an ordered list of sample sites with
 * their argument expressions, specialised to one GFI method:
 *   simulate  static.py:787-793 + SimulateHandler :254-278
 *   generate / importance  static.py:795-810 + GenerateHandler :341-380
 *   assess    static.py:983-989 + AssessHandler :297-321
 *   update    static.py:827-865 + UpdateHandler :407-466
 *   regenerate / Rejuvenate edits  static.py:867-946,
 *             src/genjax/_src/inference/requests/rejuvenate.py:70-94
 * with the leaf semantics of distributions/distribution.py:108-300.
 * One GPU thread executes the program for one particle; every value lives
 * in registers; only declared inputs/outputs touch HBM (SoA, coalesced).
 * The encoding is documented in genjax_amd/csrc/gmx_program.h.
 * ---------------------------------------------------------------------- */

#define GMX_MAX_IN 64
#define GMX_MAX_OUT 64
#define GMX_MAX_TAB 8
#define GMX_MAX_UNI 64

/* how the per-particle key (register pair loaded by OP_LDKEY) is formed */
enum {
  GMX_KEY_NONE = 0,     /* program draws nothing (assess)                      */
  GMX_KEY_ARRAY = 1,    /* keys_d[i]                                           */
  GMX_KEY_SPLIT = 2,    /* split((key0,key1), *)[index_offset + i]             */
  GMX_KEY_ROWSPLIT = 3, /* child index_offset + i % inner of keys_d[i / key_inner] (= split(row key, .)[...]) */
  GMX_KEY_BCAST = 4     /* (key0,key1) for every particle                      */
};

/* The resampling of the PREVIOUS step folded into this launch (gmx_run_args.rs): with lw_d set, a specialised
 * 4-particles-per-thread program that gathers (gmx_program_fuses_resample() == 1) does not find its ancestors in
 * ancestors_d — it WRITES them there first: workgroup b runs gmx_resample_tiles' workgroup b (the tile's CDF rebuilt from
 * the previous step's log-weights and tile statistics, the exact slot ranges, the LDS slot fill), storing every ancestor
 * as {tag: bits 24..31 | index: bits 0..23} (GMX_ANC_TAG_SHIFT = 24, ABI v8; tags 1..255) with a write-through store, and then POLLS the slots of its own particles
 * until they carry the tag — they are filled by its neighbours of the same launch, mostly — before it gathers through
 * them.  A bootstrap SMC step (smc.py:370-396's role; SURVEY.md App. B resampling) is then ONE launch instead of two:
 * the grid-wide dependency (every tile's statistics) still rides on the launch boundary, the mostly-local one (who owns
 * my slots) on tagged words.  Same integers as gmx_resample_tiles.  Every workgroup of the launch must be resident at
 * once (they wait for each other): n <= 2^20 (1024 workgroups); a wait gives up after 2 s of the device's wall clock and
 * sets *status_d.  The buffers read here must not be the ones this launch writes (log-weights, statistics: ping-pong);
 * after the launch ancestors_d holds TAGGED words (index = word & 0xffffff, tag = word >> 24). */
typedef struct gmx_resample_in {
  const float* lw_d;              /* [n] log-weights of the previous step (16-byte aligned); NULL = not fused   */
  const float* tile_max_d;        /* [ceil(n/1024)] m_b  (plane 0 of the previous launch's red_out_d)           */
  const uint64_t* tile_agg_d;     /* [ceil(n/1024)] A_b  (the previous launch's tile_agg_d, same tile_shift)    */
  float* max_out_d;               /* [1]: M      (as gmx_resample_tiles' max_d)                                 */
  uint64_t* total_out_d;          /* [1]: total  (as gmx_resample_tiles' total_d)                               */
  uint64_t* status_d;             /* [1]: sticky error word (a wait that timed out)                             */
  int32_t shift;                  /* the CDF's fixed-point shift (= the previous launch's tile_shift)           */
  uint32_t tag;                   /* 1 .. 255, different from the previous launch's                             */
  uint32_t key0, key1;            /* resampling key (systematic)                                                */
  uint32_t u0;                    /* filled in by gmx_program_run: bits32(key, 0) >> 9                          */
  uint32_t reserved_;
} gmx_resample_in;

/* The peers of a sharded SMC step ("Fused peer exchange", below): passed by value to the site program
 * (gmx_run_args.peer) and to gmx_shard_step_peer. */
#define GMX_PEER_MAX_LEAVES 32    /* routed 4-byte leaves per particle: a state of 16 components travels with its MH-moved copy */
typedef struct gmx_peer {
  void* const* land_d;            /* device array [world]: every rank's LANDING block as mapped into this process
                                     (gmx_p2p_alloc / gmx_p2p_open; land_d[rank] = this rank's own); NULL = no peers    */
  const uint32_t* tag_base_d;     /* device word: the tag of step 0 of the running sweep (gmx_peer_bump)               */
  uint64_t* status_d;             /* local device words: [0] sticky timeout flag (a peer's data never arrived)          */
  int32_t rank, world;
  int32_t step;                   /* t: everything this step puts carries tag = *tag_base_d + t; parity = tag & 1      */
  int32_t tiles;                  /* CDF tiles per rank = ceil(n_per_rank / 1024)                                     */
  int64_t capacity;               /* states one rank may ship to ONE peer per step                                     */
  int32_t leaves;                 /* routed 4-byte leaves per particle (1: a scalar state; D; 2 D with an MH move):
                                     1 .. GMX_PEER_MAX_LEAVES                                                          */
  int32_t reserved_;
} gmx_peer;

/* The ROUTING of the previous step of a SHARDED sweep folded into this launch (gmx_run_args.sh): with lw_d set, a
 * specialised 4-particles-per-thread program that gathers (gmx_program_fuses_shard_step() == 1) first runs its
 * workgroup's tile of gmx_shard_step_peer for step t - 1 — polls this rank's landing table for every rank's tile
 * statistics of that step, derives the global exponent, the totals and the slot bounds, rebuilds its tile of the CDF,
 * puts the states other ranks' slots need into their landing blocks, stores the ancestors of this rank's own slots as
 * tagged words {tag: bits 24..31 | index into the extended state: bits 0..23} in ancestors_d, waits for the granules
 * its own slots need and stores them in the local tails — and then polls the ancestor words of ITS OWN particles and
 * gathers through them.  A sharded SMC step is then ONE launch and no collective: the site
 * program's epilogue of step t - 1 announced the statistics (gmx_run_args.peer), this prologue consumes them.  Same
 * integers, same routing as gmx_shard_step_peer.  Systematic resampling; world <= 8 and world * tiles <= 1024 (the
 * table fits the registers of one workgroup); n + world * capacity <= 2^24 (the index field); every workgroup of the
 * launch resident at once (n <= gmx_program_resident_particles()).  The log-weights and the statistics block read
 * here must not be the ones this launch writes (two sets, alternating). */
typedef struct gmx_shard_in {
  const float* lw_d;              /* [n] log-weights of step t - 1 (16-byte aligned); NULL = not fused                  */
  const void* stats_own_d;        /* this rank's statistics block of step t - 1 (gmx_shard_stats_bytes: A_b, then m_b)  */
  int64_t* plan_d;                /* gmx_shard_plan_words(world): as gmx_shard_step_peer                                */
  uint64_t* total_out_d;          /* [1]: the global integer total of step t - 1                                         */
  float* max_out_d;               /* [1]: the global max log-weight of step t - 1                                        */
  uint64_t* status_d;             /* [1]: sticky error word (an ancestor word that never arrived)                        */
  int32_t shift;
  uint32_t tag;                   /* 1 .. 255: the tag of this launch's ancestor words                                   */
  uint32_t key0, key1;            /* resampling key of step t - 1                                                        */
  uint32_t u0;                    /* filled in by gmx_program_run: bits32(key, 0) >> 9                                   */
  int32_t reserved_;
  gmx_peer peer;                  /* the exchange of step t - 1: peer.step = t - 1, peer.leaves routed leaves            */
  const void* state_d[GMX_PEER_MAX_LEAVES];   /* leaf l: this rank's states of step t - 1 [n] (the head of its extended state) */
  void* tail_d[GMX_PEER_MAX_LEAVES];          /* leaf l: the tail [world * capacity] of that extended state                   */
} gmx_shard_in;

typedef struct gmx_run_args {
  const void* in_d[GMX_MAX_IN];   /* per-particle inputs (4-byte or 1-byte elems) */
  void* out_d[GMX_MAX_OUT];       /* per-particle outputs                        */
  const void* tab_d[GMX_MAX_TAB]; /* small lookup tables (4-byte elems)          */
  uint32_t uni[GMX_MAX_UNI];      /* launch-uniform 32-bit values (raw bits)     */
  const int32_t* ancestors_d;     /* row index for inputs loaded with GATHER     */
  int32_t key_mode;
  uint32_t key0, key1;
  const uint32_t* keys_d;
  int64_t key_inner;
  int64_t index_offset;           /* global index of local particle 0 (sharding) */
  float* red_out_d;               /* [2][grid] block partials of OP_REDMAX/LSE: plane 0 = block max,
                                     plane 1 = sum exp(x - block max); grid = gmx_program_grid()  */
  uint64_t* tile_agg_d;           /* optional, programs for which gmx_program_writes_tile_stats() is 1:
                                     [grid] A_b = sum over the workgroup's 1024 particles of
                                     floor(exp(x - k_b ln 2) * 2^tile_shift), k_b = ceil(block max / ln 2),
                                     x = the OP_REDMAX operand
                                     (the log-weight): with plane 0 of red_out_d the tile statistics
                                     gmx_resample_tiles needs — no separate pass over the log-weights  */
  int32_t tile_shift;
  int32_t reserved_;
  int64_t step_stride;            /* elements between consecutive steps of a [T, n] leaf addressed with GMX_F_STEP inside
                                     an OP_LOOP (programs with a counted loop: the Scan combinator); normally n         */
  gmx_resample_in rs;             /* optional (rs.lw_d != NULL): resample the previous step first, in this launch (above)  */
  gmx_shard_in sh;                /* optional (sh.lw_d != NULL): route the previous step of a sharded sweep first (above)   */
  gmx_peer peer;                  /* optional (peer.land_d != NULL), with tile_agg_d: the workgroup ALSO puts its tile's
                                     statistics straight into every other rank's landing table ("Fused peer exchange"
                                     below) — the all-gather of a sharded SMC step without a collective launch        */
} gmx_run_args;

int gmx_program_create(const uint32_t* blob_h, size_t n_words, gmx_program** out);
int gmx_program_destroy(gmx_program* p);
/* Optional: compile a kernel specialised to this program (hiprtc, gfx950): the
 * interpreter source partially evaluated against the constant instruction
 * stream, so results are bit-identical to the interpreter's.  Blocking; call
 * it outside stream capture.  Returns non-zero (and leaves the interpreter in
 * place) when hiprtc is unavailable or GENMI_JIT=0. */
int gmx_program_specialize(gmx_program* p);
int gmx_program_is_specialized(const gmx_program* p);
/* FNV-1a (64-bit) of the specialised kernel's code object, 0 when not specialised: the identity of the code a
 * measurement was taken on (bench.py only uses profile-sourced instruction counts whose recorded hash equals this). */
uint64_t gmx_program_code_hash(const gmx_program* p);
/* A specialised kernel is the output of a compiler (hiprtc was caught twice miscompiling one: DESIGN.md section 5).  The
 * host layer therefore runs every freshly specialised program ONCE beside the ahead-of-time interpreter on the first few
 * hundred particles of its first launch and compares the outputs bit for bit (engine.Compiled._cross_check; programs of
 * at most 31 live values — what the interpreter holds).  On a difference it calls gmx_program_despecialize: the module is
 * unloaded, the program runs on the interpreter from then on, `why` becomes gmx_last_error() and the process-wide count
 * gmx_jit_rejected_count() goes up.  (The reference has no counterpart: XLA is trusted there.) */
int gmx_program_despecialize(gmx_program* p, const char* why);
int64_t gmx_jit_rejected_count(void);
/* number of thread blocks gmx_program_run will launch for n particles
 * (= rows of red_out_d the caller must provide). */
int64_t gmx_program_grid(const gmx_program* p, int64_t n);
/* 1 when the specialised kernel runs 4 particles per thread (a workgroup = one 1024-particle
 * tile of the CDF) and the program has exactly one OP_REDMAX: it then honours tile_agg_d. */
int gmx_program_writes_tile_stats(const gmx_program* p);
/* Ask, BEFORE gmx_program_specialize, for a kernel that can resample first (gmx_run_args.rs): the prologue lengthens the
 * hiprtc compile, so only the programs a sweep chains ask for it.  gmx_program_fuses_resample: 1 once such a kernel
 * exists (4 particles per thread, every gathered load through ancestors_d at the top of the kernel). */
int gmx_program_set_fuse_resample(gmx_program* p);
/* ... and for a LOOPED one: a launch of more particles than the device holds workgroups for (n > 2^20, up to 2^24) runs
 * min(tiles, resident, 1024) workgroups, each walking tiles b, b + G, ...: first the resampling of the previous step for
 * all of its tiles (one pass over the statistics table per workgroup; nothing there waits), then, tile by tile, the wait
 * for its own ancestor words, the gather and the site program.  gmx_program_resident_particles answers for such a program
 * with what ONE launch covers (16 tiles per workgroup). */
int gmx_program_set_fuse_resample_loop(gmx_program* p);
int gmx_program_fuses_resample(const gmx_program* p);
/* The number of particles ONE launch of the specialised kernel covers with every workgroup resident at once
 * (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor of its code object x the device's CUs x particles per
 * workgroup); 0 when not specialised.  The resample-first launch (gmx_run_args.rs) is refused beyond it: its
 * workgroups wait for each other, and one that is never scheduled would stall the rest until their timeout.
 * Callers fall back to two launches per step (gmx_resample_tiles + the site program). */
int64_t gmx_program_resident_particles(const gmx_program* p);
/* The same for the ROUTING of a sharded step (gmx_run_args.sh): ask before gmx_program_specialize; 1 once such a kernel
 * exists. */
int gmx_program_set_fuse_shard_step(gmx_program* p);
int gmx_program_fuses_shard_step(const gmx_program* p);
/* Mark a program as BACKGROUND work before it is specialised: work that depends on nothing a dependent chain of
 * launches produces — e.g. the standard-normal draws of the next SMC steps (keys and particle indices only), which
 * BootstrapSweep's noise-ahead form issues on a second stream beside the chain [site program -> resampler].  The
 * specialised kernel keeps wave priority 0 (the chain's kernels raise theirs) and every launch asks for `lds_pad`
 * bytes of dynamic LDS it never touches — a cap on the workgroups of this kernel a CU holds (160 KB of LDS per CU),
 * so that the chain's kernels always find wave slots.  No effect on results, nor on the interpreter.
 * A background program that reads no per-particle input and reduces nothing, run with GMX_KEY_ROWSPLIT keys over
 * n = rows x key_inner particles (rows <= 65535), is launched as a 2-D grid, one row of keys per blockIdx.y: the
 * draws of `rows` SMC steps from one launch, output leaves laid out [rows, key_inner].
 * (No reference counterpart: XLA schedules its fused loops itself.) */
int gmx_program_set_background(gmx_program* p, uint32_t lds_pad);
int gmx_program_run(const gmx_program* p, int64_t n, const gmx_run_args* args_h,
                    gmx_stream stream);

/* ------------------------------------------------------------------------
 * Log-normaliser.  Replaces jax.scipy.special.logsumexp at
 *   src/genjax/_src/inference/smc.py:96-97 (log-ML estimate), :107, :464.
 * out_d[r] = logsumexp(lw_d[r, 0:cols]); deterministic (fixed reduction tree).
 * workspace: gmx_logsumexp_workspace(rows, cols) bytes.
 * ---------------------------------------------------------------------- */
size_t gmx_logsumexp_workspace(int64_t rows, int64_t cols);
int gmx_logsumexp(const float* lw_d, int64_t rows, int64_t cols, float* out_d,
                  float* out_max_d /* optional [rows] */, void* workspace_d,
                  gmx_stream stream);

/* ------------------------------------------------------------------------
 * Plate sums.  Replaces the `jnp.sum` over the plate axis of `Vmap.simulate / generate / assess / edit`
 * (src/genjax/_src/generative_functions/combinators/vmap.py:180-218, 236-275) when the plate's elements run on the
 * launch axis (one key, a large plate): out_d[r] = sum of x_d[r, 0:cols] in a FIXED tree — per tile of 4096 items
 * thread t adds items t, t + 256, ... in order, a wave butterfly (xor 32 ... 1), (w0 + w1) + (w2 + w3); then the same
 * over the tile partials.  Deterministic: the same bits on every run (the oracle restates the tree).
 * ---------------------------------------------------------------------- */
/* out_d[r] = the sum of x[r, 0 .. cols) added in ELEMENT ORDER (0.0f + x[r, 0] + x[r, 1] + ...; element (r, c) at
 * x_d[r * stride_row + c * stride_col]): the plate score of a batched plate trace as its counted loop accumulates it
 * (vmap.py:214-216's sum), recomputed after an IndexRequest replaced one element. */
int gmx_sum_rows_inorder(const float* x_d, int64_t rows, int64_t cols, int64_t stride_row, int64_t stride_col,
                         float* out_d, gmx_stream stream);
size_t gmx_sum_rows_workspace(int64_t rows, int64_t cols);
int gmx_sum_rows(const float* x_d, int64_t rows, int64_t cols, float* out_d, void* workspace_d, gmx_stream stream);

/* ------------------------------------------------------------------------
 * Resampling.  The reference has only the single-index Gumbel-max draw
 * (`ParticleCollection.sample_particle`, smc.py:102-109) and the cookbook's
 * O(N*K) SIR idiom (docs/cookbook/inactive/inference/importance_sampling.ipynb
 * cell 16); SURVEY.md App. B defines the scalable forms implemented here.
 *
 * Step 1  gmx_weight_cdf: the integer CDF, in block floating point so that the only
 *         global quantities are two numbers per tile of 1024 consecutive GLOBAL indices:
 *           m_b = max lw over the tile,  k_b = ceil(m_b / ln 2) (f32; clamped to +-2^29),
 *           l_i = floor(exp(lw_i - k_b ln 2) * 2^shift) (u64),
 *           L_i = inclusive tile-local sum,  A_b = L_last;
 *         with M = max_b m_b and K = ceil(M / ln 2):
 *           G_b = A_b >> (K - k_b),  cdf_i = sum_{b'<b} G_b' + (L_i >> (K - k_b)).
 *         shift = 62 - ceil(log2(n_total)) so a global sum cannot overflow.  Integers
 *         throughout after the per-particle exp: the same on any machine and under any
 *         partitioning whose shards start on a tile boundary.  total * 2^-shift =
 *         sum_i exp(lw_i - K ln 2): evidence terms use K ln 2 (f32) as the reference.
 * Step 2  gmx_ancestors: ancestor of output slot j by exact 128-bit integer
 *         comparison against the CDF (no floating point):
 *           SYSTEMATIC   first i with cdf_i * (n_out*2^23) > (j*2^23 + u0) * total
 *           STRATIFIED   same with a per-slot u_j
 *           MULTINOMIAL  first i with cdf_i * 2^23 >= total * (2^23 - u_j)
 *                        (jax.random.choice inverse-CDF semantics)
 *         u = 23-bit uniforms (bits >> 9) from `key`.
 * Step 3  gmx_gather: dst[leaf][j] = src[leaf][ancestors[j]]
 *         (`get_particle` tree_map, smc.py:90-91).
 * gmx_categorical_rows: one Gumbel-max index per row (sample_particle).
 * ---------------------------------------------------------------------- */
enum { GMX_RESAMPLE_SYSTEMATIC = 0, GMX_RESAMPLE_STRATIFIED = 1, GMX_RESAMPLE_MULTINOMIAL = 2,
       GMX_RESAMPLE_MULTINOMIAL_TILED = 3 /* gmx_multinomial_tiled only */,
       GMX_RESAMPLE_MULTINOMIAL_SORTED = 4 /* gmx_resample_sorted only */ };

/* *max_d = max of the n block maxima a program's OP_REDMAX wrote (plane 0 of red_out_d). */
int gmx_reduce_max(const float* partials_d, int64_t n, float* max_d, gmx_stream stream);
size_t gmx_weight_cdf_workspace(int64_t n);
/* max_d: device scalar (max over ALL shards' log-weights); if max_partials_d
 * is non-null the max is first reduced from the n_partials block maxima
 * a program's OP_REDMAX wrote (plane 0 of red_out_d) and stored to max_d. */
int gmx_weight_cdf(const float* lw_d, int64_t n, int shift,
                   const float* max_partials_d, int64_t n_partials, float* max_d,
                   uint64_t* cdf_d /* [n] inclusive */, uint64_t* total_d /* [1] */,
                   void* workspace_d, gmx_stream stream);
int gmx_ancestors(int kind, const uint32_t key[2], const uint64_t* cdf_d, int64_t n_in,
                  uint64_t cdf_offset /* added to every cdf entry (sharding) */,
                  const uint64_t* total_d /* [1] global total */,
                  int64_t n_out_total, int64_t slot_offset, int64_t n_slots,
                  int32_t* ancestors_d /* [n_slots], clamped to [0,n_in) local */,
                  gmx_stream stream);
/* MULTINOMIAL over the whole (unsharded) CDF through a guide table: ancestors identical to
 * gmx_ancestors(GMX_RESAMPLE_MULTINOMIAL, key, cdf_d, n_in, 0, total_d, n_out, 0, n_out, ...), several times faster
 * for large n_in (the per-slot binary search over n_in entries becomes two table reads + a search over ~3 entries;
 * the table is the systematic offspring assignment with offset 0, built first).  workspace_d:
 * gmx_multinomial_workspace(n_in) bytes. */
size_t gmx_multinomial_workspace(int64_t n_in);
int gmx_multinomial(const uint32_t key[2], const uint64_t* cdf_d, int64_t n_in, const uint64_t* total_d /* [1] */,
                    int64_t n_out, int32_t* ancestors_d /* [n_out] */, void* workspace_d, gmx_stream stream);
/* ---- THE resampling entry point --------------------------------------------------------------------------------
 * gmx_resample(kind, key, lw_d, n, shift, ..., workspace_d): log-weights -> ancestors, *max_d, *total_d for EVERY kind
 * (GMX_RESAMPLE_SYSTEMATIC / _STRATIFIED / _MULTINOMIAL / _MULTINOMIAL_TILED / _MULTINOMIAL_SORTED) and every
 * n < 2^31 (tiled: n <= 2^21): a reference-side binder needs this one call (`gmx_resample(kind, ...)`).
 * It dispatches to the STAGED forms declared after it, which are public for callers that hold part of the work
 * already — a specialised site program leaves the tile statistics itself (gmx_run_args.tile_agg_d), a background
 * stream draws the slot uniforms / the order-statistics table ahead:
 *   systematic / stratified   gmx_tile_stats -> gmx_resample_tiles            (n <= 2^21)
 *                             gmx_tile_stats -> gmx_tile_prefix -> gmx_resample_tiles_p          (any n)
 *   multinomial_sorted        gmx_tile_stats [-> gmx_tile_prefix] -> gmx_resample_sorted[_p] (the table built first)
 *   multinomial_tiled         gmx_tile_stats -> gmx_multinomial_tiled(phase = -1)
 *   multinomial (iid order)   gmx_weight_cdf -> gmx_multinomial
 * workspace_d: gmx_resample_workspace(n) bytes, 16-byte aligned (covers every kind).  max_partials_d / n_partials are
 * accepted and unused.  DEPRECATED, kept for existing callers: gmx_ancestors over a CDF array as the way to resample
 * with an ORDERED kind (use gmx_resample), and the CDF-array shard forms gmx_shard_plan / _route / _step below
 * (superseded by the tile-statistics forms gmx_shard_step_fused / gmx_shard_step_peer). */
/* Fused one-GPU form of steps 1+2 for SYSTEMATIC / STRATIFIED (n <= 2^21): no CDF in
 * memory and no inter-block waiting; ancestors identical to gmx_weight_cdf +
 * gmx_ancestors.  gmx_tile_stats writes (m_b, A_b) per 1024-particle tile — a
 * specialised site program writes the same two numbers itself from its OP_REDMAX
 * epilogue when gmx_run_args.tile_agg_d is set (gmx_program_writes_tile_stats), which
 * makes an SMC step two launches.  gmx_resample_tiles turns log-weights + tile stats
 * into ancestors; *max_d / *total_d receive the max log-weight and the integer total
 * (for the evidence increment). */
size_t gmx_resample_workspace(int64_t n);
int gmx_tile_stats(const float* lw_d, int64_t n, int shift, float* tile_max_d /* [ceil(n/1024)] */,
                   uint64_t* tile_agg_d /* [ceil(n/1024)] */, gmx_stream stream);
int gmx_resample_tiles(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                       const float* tile_max_d, const uint64_t* tile_agg_d, float* max_d,
                       uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream);
int gmx_resample(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                 const float* max_partials_d, int64_t n_partials, float* max_d,
                 uint64_t* total_d, int32_t* ancestors_d, void* workspace_d, gmx_stream stream);
/* gmx_resample_tiles when the tile PREFIXES are there already (gmx_tile_prefix, one workgroup, from the statistics):
 * a workgroup of the resampler reads ONE prefix, the total and
 * K instead of reducing all <= 2048 tile statistics — the same integers, the same ancestors.  This form (and
 * gmx_tile_stats, gmx_tile_prefix) takes any n < 2^31: past 2048 tiles (n > 2^21) one workgroup walks the table in
 * chunks with a running carry — how a population of 1e7 is resampled without a CDF array.
 * gmx_tile_prefix_words(n): u64 words of the block (ceil(n / 1024) + 2, rounded up to 16). */
size_t gmx_tile_prefix_words(int64_t n);
int gmx_tile_prefix(const float* tile_max_d, const uint64_t* tile_agg_d, int64_t n, uint64_t* tile_pref_d,
                    gmx_stream stream);
int gmx_resample_tiles_p(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                         const float* tile_max_d, const uint64_t* tile_pref_d, float* max_d,
                         uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream);
/* Two-stage multinomial resampling from log-weights + tile statistics (GMX_RESAMPLE_MULTINOMIAL_TILED; build-defined,
 * SURVEY App. B): offspring counts are Multinomial(n, w), the output is ordered by the ancestor's 1024-particle CDF tile
 * (iid order inside a tile), so the gather that follows reads tile by tile instead of n random lines.
 *   (k1, k2) = split(key, 2)
 *   stage 1  slot j picks a tile: P = (u_j * total) >> 23, u_j = bits32(k1, j) >> 9; tile = first b whose end-of-tile
 *            CDF exceeds P; c_b = slots that picked b (LDS histograms, then integer atomics: order-independent)
 *   stage 2  tile b's c_b slots are consecutive output positions (tiles in order); its r-th slot picks the local
 *            position Q = (v * G_b) >> 23, v = bits32(fold_in(k2, b), r) >> 9, in the tile's own CDF (rebuilt in LDS)
 * u_d (optional, [n]): the stage-1 uniforms drawn ahead by gmx_slot_uniforms(keys = k1).  workspace_d:
 * gmx_multinomial_tiled_workspace(n) bytes, 16-byte aligned: TWO buffers of tile counts.  A call counts into one and
 * leaves the other zero.  phase 0 / 1: into buffer `phase`, which must be zero — as the previous call, made with the
 * other phase, left it (a sequence of calls alternating 0, 1, 0, ... needs no memset); phase -1: into buffer 0 after
 * zeroing it (a memset node in a captured graph): a one-off call, or the first of a sequence.  n <= 2^21. */
size_t gmx_multinomial_tiled_workspace(int64_t n);
int gmx_multinomial_tiled(const uint32_t key[2], const float* lw_d, int64_t n, int shift, const float* tile_max_d,
                          const uint64_t* tile_agg_d, const uint32_t* u_d, float* max_d, uint64_t* total_d,
                          int32_t* ancestors_d, void* workspace_d, int phase, gmx_stream stream);
/* The stratified resampler's per-slot uniforms drawn AHEAD of the resampling (they depend on the key and the slot
 * number only): gmx_slot_uniforms fills out_d[r][j] = bits32(keys_d[r], j) >> 9 for `rows` resampling keys (device
 * array [rows, 2]) in one 2-D launch — meant for a background stream beside a dependent chain of launches: `lds_pad`
 * bytes of unused LDS per workgroup cap its residency (0: none).  gmx_resample_tiles_u = gmx_resample_tiles
 * (stratified) reading slot j's uniform from u_d[j] instead of drawing it: the same ancestors, one Threefry block per
 * slot-edge evaluation less on the chain.  (No reference counterpart: build-defined resampling, SURVEY App. B.) */
int gmx_slot_uniforms(const uint32_t* keys_d /* [rows,2] */, int rows, int64_t n, uint32_t* out_d /* [rows,n] */,
                      int lds_pad, gmx_stream stream);
int gmx_resample_tiles_u(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                         const float* tile_max_d, const uint64_t* tile_agg_d, const uint32_t* u_d /* [n] */,
                         float* max_d, uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream);
/* Multinomial resampling with SORTED uniforms (GMX_RESAMPLE_MULTINOMIAL_SORTED; build-defined, SURVEY App. B;
 * csrc/gmx_sorted.h): n sorted iid uniforms are the normalised partial sums of n + 1 unit exponentials, so
 *   E_j = 1 + trunc(-log(u_j) * 2^16),  u_j = ((w_j >> 9) + 0.5) * 2^-23,  j = 0 .. n,
 *         (w_2i, w_2i+1) = the two words of threefry(key, ctr = i)
 *   S_j = E_0 + ... + E_j (j < n),  S_total = S_{n-1} + E_n
 *   ancestor(j) = first i with cdf_i * S_total > S_j * total                               (128-bit integers)
 * gives Multinomial(n, w) offspring counts with the output ordered by ancestor — an ordered scheme like systematic /
 * stratified: it runs on the same kernel (no CDF array, no search, no random cache lines) and the gather behind it
 * streams.  The order-statistics table (low words of S_j, a guide over buckets of S, tile offsets; layout in
 * csrc/gmx_sorted.h) depends on the key and n only: gmx_sorted_uniforms writes it for `rows` keys (device array
 * [rows, 2]; out_d: rows x gmx_sorted_uniforms_words(n) uint32, 16-byte aligned) in two 2-D launches meant for a
 * background stream (`lds_pad` as for gmx_slot_uniforms).  gmx_resample_sorted: log-weights + tile statistics +
 * table -> ancestors; table_ready = 0: the table is built here first (table_d is then the scratch for it).  n <= 2^21. */
size_t gmx_sorted_uniforms_words(int64_t n);
int gmx_sorted_uniforms(const uint32_t* keys_d /* [rows,2] */, int rows, int64_t n, uint32_t* out_d, int lds_pad,
                        gmx_stream stream);
int gmx_resample_sorted(const uint32_t key[2], const float* lw_d, int64_t n, int shift, const float* tile_max_d,
                        const uint64_t* tile_agg_d, uint32_t* table_d, int table_ready, float* max_d,
                        uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream);
/* the same past 2^21 particles: tile_pref_d = the prefix block gmx_tile_prefix made of the statistics (any n < 2^31) */
int gmx_resample_sorted_p(const uint32_t key[2], const float* lw_d, int64_t n, int shift, const float* tile_max_d,
                          const uint64_t* tile_pref_d, uint32_t* table_d, int table_ready, float* max_d,
                          uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream);
int gmx_gather(const void* const* src_d, void* const* dst_d, const int32_t* elem_bytes,
               int32_t n_leaves, const int32_t* ancestors_d, int64_t n_out,
               gmx_stream stream);
int gmx_categorical_rows(const uint32_t* keys_d /* [rows,2] */, const float* logits_d,
                         int64_t rows, int64_t cols, int32_t* out_idx_d, gmx_stream stream);

/* ------------------------------------------------------------------------
 * Global resampling across ranks (one process per GPU).  No
 * reference counterpart: the reference is single-device; this
 * replaces what a sharded `vmap`-over-particles resample would need.
 * Rank r owns the global particle indices / output slots [r*n, (r+1)*n).
 * After gmx_weight_cdf (with the GLOBAL max in *max_d) and an all-gather of
 * the ranks' totals, the rest of the step runs on the device with no host
 * synchronisation:
 *   gmx_shard_plan   bounds[s] = f(sum_{q<s} total_q): the first slot whose
 *                    ancestor lives on rank s (exact integer predicate of
 *                    gmx_ancestors); plan_d = { global total, this rank's
 *                    CDF offset, overflow flag, -, bounds[0..world] }.
 *   gmx_shard_route  for every slot whose ancestor i is on this rank:
 *                      slot owned by this rank  -> next_idx_d[slot - r*n] = i
 *                      slot owned by rank d     -> send_d[d*capacity + k] = state_d[i]
 *                    (k = position among the slots this rank sends to d), and
 *                    for every slot of this rank whose ancestor is on rank s != r:
 *                      next_idx_d[slot - r*n] = n + s*capacity + k
 *                    — an index into the extended state [ n local | world*capacity
 *                    received ] that ONE equal-split all-to-all of send_d fills.
 * plan_d[GMX_PLAN_OVERFLOW] is set (sticky) when some (source, destination)
 * pair needs more than `capacity` slots; the caller reads it once per sweep and
 * re-runs with capacity = n, which always suffices.  State elements are 4 bytes.
 * ---------------------------------------------------------------------- */
enum { GMX_PLAN_TOTAL = 0, GMX_PLAN_OFFSET = 1, GMX_PLAN_OVERFLOW = 2, GMX_PLAN_BOUNDS = 4 };
size_t gmx_shard_plan_words(int world);          /* int64 words of plan_d */
int gmx_shard_plan(int kind, const uint32_t key[2], const uint64_t* totals_d /* [world] */, int rank,
                   int world, int64_t n_per_rank, int64_t* plan_d,
                   uint64_t* total_out_d /* [1] or NULL: the global total */, gmx_stream stream);
int gmx_shard_route(int kind, const uint32_t key[2], int64_t* plan_d, const uint64_t* cdf_d, int rank,
                    int world, int64_t n_per_rank, int64_t capacity, const void* state_d /* [n] */,
                    void* send_d /* [world*capacity] */, int32_t* next_idx_d /* [n] */,
                    gmx_stream stream);
/* gmx_shard_plan + gmx_shard_route as ONE launch (every block derives the boundaries itself);
 * plan_d still receives total / offset / bounds, and keeps its sticky overflow word. */
int gmx_shard_step(int kind, const uint32_t key[2], const uint64_t* totals_d /* [world] */, int64_t* plan_d,
                   uint64_t* total_out_d /* [1] or NULL */, const uint64_t* cdf_d, int rank, int world,
                   int64_t n_per_rank, int64_t capacity, const void* state_d, void* send_d,
                   int32_t* next_idx_d, gmx_stream stream);
/* The same step for GMX_RESAMPLE_MULTINOMIAL_SORTED (ref: inference/smc.py:274-339 resamples with
 * `categorical(log_weights)` per slot = multinomial; csrc/gmx_sorted.h draws the n sorted uniforms as order
 * statistics): table_d is gmx_sorted_uniforms(step key, n_per_rank * world) — integers, so every rank builds the
 * same table — and a CDF value's slot count is a guided look-up in it.  Two launches (plan, route). */
int gmx_shard_step_sorted(const uint32_t* table_d, const uint64_t* totals_d /* [world] */, int64_t* plan_d,
                          uint64_t* total_out_d, const uint64_t* cdf_d, int rank, int world, int64_t n_per_rank,
                          int64_t capacity, const void* state_d, void* send_d, int32_t* next_idx_d, gmx_stream stream);
/* The same step from TILE STATISTICS — two collectives per step instead of three, no local CDF array.
 * Each rank's statistics block (gmx_shard_stats_bytes(n_per_rank) bytes: agg[tiles_pad] u64, then
 * tmax[tiles_pad] f32, tiles_pad = ceil(n_per_rank / 1024) rounded up to even; written by gmx_tile_stats or by
 * the site program itself, gmx_run_args.tile_agg_d / red_out_d) is all-gathered — that one collective carries what
 * the max all-reduce and the totals all-gather carried.  gmx_shard_totals (one block) turns the gathered
 * table into the global max (*max_d) and every rank's integer total (totals_d[world]);
 * gmx_shard_step_tiles is gmx_shard_step with this rank's CDF rebuilt per tile in registers from lw_d and
 * its own block.  world <= 64; shards start on a tile boundary (n_per_rank % 1024 == 0 when world > 1).  Any
 * n_per_rank < 2^31 for these two calls (a workgroup sums the mass of its rank's earlier tiles by striding over the
 * block — BASELINE config 4's k = 1e7 on few ranks; before ABI v8's second revision: <= 2^21); the ONE-launch forms
 * below (gmx_shard_step_fused, the peer forms) hold the table in registers / LDS and stay at n_per_rank <= 2^21. */
size_t gmx_shard_stats_bytes(int64_t n_per_rank);
int gmx_shard_totals(const void* stats_all_d /* [world] blocks */, int world, int64_t n_per_rank,
                     uint64_t* totals_d /* [world] */, float* max_d /* [1] */, gmx_stream stream);
int gmx_shard_step_tiles(int kind, const uint32_t key[2], const uint64_t* totals_d, int64_t* plan_d,
                         uint64_t* total_out_d, const float* lw_d, const void* stats_own_d, const float* max_d,
                         int shift, int rank, int world, int64_t n_per_rank, int64_t capacity,
                         const void* state_d, void* send_d, int32_t* next_idx_d, gmx_stream stream);
/* gmx_shard_totals + gmx_shard_step_tiles as ONE launch, straight from the all-gathered table (this rank's own block
 * of it serves as stats_own_d; *max_out_d receives the global max): a sharded SMC step is then site program ->
 * all-gather -> this -> all-to-all.  world <= 64. */
int gmx_shard_step_fused(int kind, const uint32_t key[2], const void* stats_all_d, int64_t* plan_d,
                         uint64_t* total_out_d /* [1] or NULL */, const float* lw_d, float* max_out_d, int shift,
                         int rank, int world, int64_t n_per_rank, int64_t capacity, const void* state_d,
                         void* send_d, int32_t* next_idx_d, gmx_stream stream);

/* ------------------------------------------------------------------------
 * Fused peer exchange (GENMI_COMM=peer): a sharded SMC step with NO collective launch.  No reference counterpart (the
 * reference is single-device; SURVEY.md 8e asks for "direct P2P ..., not ring").  Two launches per step:
 *   site program      its epilogue writes the tile's (m_b, A_b) to the local table as always AND puts them into the
 *                     landing table of every OTHER rank (gmx_run_args.peer);
 *   gmx_shard_step_peer  gmx_shard_step_fused reading the other ranks' statistics from its own landing table as they
 *                     arrive; a slot another rank owns gets the ancestor's state put straight into THAT rank's landing
 *                     block; at its end, each of this rank's slots whose ancestor is remote waits for that one value
 *                     and stores it in the local extended state (ordinary memory: the next site program is unchanged).
 * Everything that crosses ranks is an 8-byte GRANULE {32 bits of data, 32-bit tag} written by one write-through store
 * and read by polling with system-scope loads until the tag is the step's: naturally aligned 8-byte accesses are
 * single-copy atomic, so there is no flag, no counter, no fence (no L2 write-back / invalidate) and no ordering
 * requirement between granules.  Tags are *tag_base_d + t: they grow by one per step ACROSS sweeps (gmx_peer_bump adds
 * T at the head of every sweep, inside the captured graph), parity = tag & 1 selects one of two landing halves, and a
 * rank can only overwrite a half two steps later — by when every reader of it has finished (its own next launch needs
 * every peer's next statistics).  A poll gives up after a bounded number of spins and sets status_d[0]; the caller
 * reads it once per sweep.  Landing block of a rank (bytes; the same layout on every rank, fine-grained memory from
 * gmx_p2p_alloc, zeroed):
 *   statistics  [2 halves][world][tiles][3] u64: {A_b low | tag}, {A_b high | tag}, {m_b bits | tag}
 *   states      [2 halves][leaves][world][capacity] u64: {state bits | tag}   (block s: what rank s ships here)
 * ---------------------------------------------------------------------- */
size_t gmx_peer_landing_bytes(int world, int64_t n_per_rank, int64_t capacity, int leaves);
/* *tag_base_d += T (one thread): at the head of every sweep, before its first site program */
int gmx_peer_bump(uint32_t* tag_base_d, int32_t T, gmx_stream stream);
/* The verdict of a sharded sweep, folded on the device at its end (one thread; inside a captured sweep too):
 * *verdict_d = 2 if any of the n_status (<= 4) status words is non-zero (a wait that gave up: the results are not
 * valid), else the overflow word *overflow_d of the plan (non-zero: re-run with full capacity).  The host then reads ONE
 * word per sweep (after a MAX over ranks) instead of every status word in turn.  No reference counterpart. */
int gmx_sweep_verdict(const int64_t* overflow_d, const uint64_t* const* status_h, int32_t n_status, int64_t* verdict_d,
                      gmx_stream stream);
/* Put a statistics block that a launch without the epilogue wrote (gmx_tile_stats; an interpreted site program) */
int gmx_peer_put_stats(const void* stats_own_d, gmx_peer peer, int64_t n_per_rank, gmx_stream stream);
/* state_rows_h / tail_rows_h: HOST arrays [peer.leaves] of device pointers — leaf l's local states [n_per_rank] and the
 * tail [world * capacity] of its extended state that receives what the other ranks ship. */
int gmx_shard_step_peer(int kind, const uint32_t key[2], const void* stats_own_d, gmx_peer peer, int64_t* plan_d,
                        uint64_t* total_out_d /* [1] or NULL */, const float* lw_d, float* max_out_d, int shift,
                        int64_t n_per_rank, const void* const* state_rows_h, void* const* tail_rows_h,
                        int32_t* next_idx_d, gmx_stream stream);

/* ------------------------------------------------------------------------
 * MH accept + select.  Replaces the user idiom
 *   check = log(uniform.sample(k, 0, 1)) < w; tr = tree_map(where(check, new, old))
 * (tests/inference/test_requests.py:131-137, 186-191).
 * accept_d[i] = log(uniform(keys_d[i])) < log_alpha_d[i].
 * ---------------------------------------------------------------------- */
int gmx_mh_accept(const uint32_t* keys_d /* [n,2] */, const float* log_alpha_d, int64_t n,
                  uint8_t* accept_d, gmx_stream stream);
int gmx_select(const uint8_t* mask_d, const void* const* a_d, const void* const* b_d,
               void* const* out_d, const int32_t* elem_bytes, int32_t n_leaves, int64_t n,
               gmx_stream stream);

/* ------------------------------------------------------------------------
 * DEPRECATED as a collective (gmx_p2p_exchange: one launch per collective; superseded by the fused peer exchange above —
 * gmx_run_args.peer + gmx_shard_step_peer, no collective launch at all).  gmx_p2p_alloc / _open / _close / _free stay:
 * they are how the fused exchange's landing blocks are allocated and mapped.
 * Peer-mapped exchange over xGMI (GENMI_COMM=p2p; SURVEY.md 8e: "direct P2P all-to-all, not ring").  No reference
 * counterpart (the reference is single-device).  One process per GPU; every rank allocates its receive buffers and a
 * flag row with gmx_p2p_alloc (fine-grained device memory + an IPC handle), the handles travel once through
 * torch.distributed, every rank maps its peers' buffers with gmx_p2p_open.  A collective is then ONE launch and one
 * rendezvous, no library in between:
 *   gmx_p2p_exchange   workgroup d copies `bytes` from src_d + d * src_stride (src_stride = 0: the same block for
 *                      every peer = all-gather; `bytes`: equal-split all-to-all) into PEER d's LANDING buffer
 *                      land_peers_d[d] + half + rank * bytes, makes the copy visible at system scope, writes this
 *                      launch's epoch into peer d's flag word flag_peers_d[d][rank], waits until its own flag word
 *                      flags_local_d[d] carries the epoch (peer d's block has landed), and copies slot d of its own
 *                      landing buffer land_local_d into out_d + d * bytes — ordinary device memory: only the small
 *                      landing buffers (2 * world * bytes: `half` alternates with the epoch's parity, so a fast peer's
 *                      next exchange cannot overwrite a block still being copied out) and the flags are fine-grained.
 *                      The epoch lives on the device (state_d[0], advanced by the last workgroup), so a captured
 *                      graph replays correctly.
 * A wait gives up after 4 s of the device's wall clock and sets state_d[1] (sticky error word: a peer that never arrives
 * must not hang the GPU); the caller checks it once per sweep.  Unmeasured across GPUs (no multi-GPU box in the build
 * loop): exercised at world size 1 on the device and, through the tests' CPU mirror with process-shared memory, at
 * world sizes 2 and 4.
 * ---------------------------------------------------------------------- */
#define GMX_P2P_HANDLE_BYTES 64
int gmx_p2p_alloc(size_t bytes, void** ptr_out, void* handle_out /* GMX_P2P_HANDLE_BYTES */);
int gmx_p2p_open(const void* handle /* GMX_P2P_HANDLE_BYTES */, void** ptr_out);
int gmx_p2p_close(void* ptr);
int gmx_p2p_free(void* ptr);
int gmx_p2p_exchange(const void* src_d, size_t src_stride, void* const* land_peers_d /* [world] */,
                     const void* land_local_d /* 2 * world * bytes */, void* out_d /* world * bytes */,
                     uint64_t* const* flag_peers_d /* [world] */, uint64_t* flags_local_d /* [world] */,
                     uint64_t* state_d /* [3 + world], zeroed: epoch, error, tickets */, int rank, int world, size_t bytes,
                     gmx_stream stream);

/* ------------------------------------------------------------------------
 * Graph capture helpers (launch-bound sweeps: T steps x few kernels).
 * ---------------------------------------------------------------------- */
typedef struct gmx_graph gmx_graph;
int gmx_capture_begin(gmx_stream stream);
int gmx_capture_end(gmx_stream stream, gmx_graph** out);
int gmx_graph_launch(gmx_graph* g, gmx_stream stream);
int gmx_graph_destroy(gmx_graph* g);

/* Timing helper for bench.py: HIP events on the caller's stream. */
typedef struct gmx_timer gmx_timer;
int gmx_timer_create(gmx_timer** out);
int gmx_timer_start(gmx_timer* t, gmx_stream stream);
int gmx_timer_stop(gmx_timer* t, gmx_stream stream);
int gmx_timer_elapsed_ms(gmx_timer* t, float* ms_out); /* synchronises on stop */
int gmx_timer_destroy(gmx_timer* t);

#ifdef __cplusplus
}
#endif
#endif /* GENMI_H */
