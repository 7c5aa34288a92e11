// hostsim.cpp — TEST HARNESS ONLY.  A sequential CPU implementation of the
// C-ABI in include/genmi.h so the Python host layer (tracer, program encoder,
// binding plans, SMC combinators) can be exercised on machines without a GPU.
//
// The site-program interpreter, samplers and log-densities are the PRODUCT's
// own headers (genjax_amd/csrc/gmx_vm.h ...) compiled for the host, so CPU
// tests diff the product's device functions against the independent oracle.
// The scan / search / gather entry points are plain re-implementations (the
// gfx950 kernels themselves are validated by the `-m gpu` tests).
//
// Nothing in genjax_amd/ refers to this file; tests install it through
// genjax_amd._lib.install().  It is never shipped as a fallback.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "../../genjax_amd/csrc/gmx_vm.h"  // build with -I include

static thread_local char g_err[512] = "";
static int fail(const char* m) { snprintf(g_err, sizeof(g_err), "%s", m); return 1; }

extern "C" int gmx_version(void) { return GMX_ABI_VERSION; }
extern "C" size_t gmx_run_args_bytes(void) { return sizeof(gmx_run_args); }
extern "C" const char* gmx_last_error(void) { return g_err; }
extern "C" void gmx_threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
  gmx_threefry2x32(k0, k1, c0, c1, &out[0], &out[1]);
}

extern "C" int gmx_split(const uint32_t key[2], int64_t n, int64_t off, uint32_t* out, gmx_stream) {
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  for (int64_t i = 0; i < n; ++i) { gmx_key o = gmx_split_child(k, (uint64_t)(off + i)); out[2*i] = o.k0; out[2*i+1] = o.k1; }
  return 0;
}
extern "C" int gmx_split_rows(const uint32_t* keys, int64_t rows, int64_t inner, uint32_t* out, gmx_stream) {
  for (int64_t r = 0; r < rows; ++r) {
    gmx_key k; k.k0 = keys[2*r]; k.k1 = keys[2*r+1];
    for (int64_t j = 0; j < inner; ++j) { gmx_key o = gmx_split_child(k, (uint64_t)j); out[2*(r*inner+j)] = o.k0; out[2*(r*inner+j)+1] = o.k1; }
  }
  return 0;
}
extern "C" int gmx_fold_in(const uint32_t* keys, uint32_t data, int64_t n, uint32_t* out, gmx_stream) {
  for (int64_t i = 0; i < n; ++i) { gmx_key k; k.k0 = keys[2*i]; k.k1 = keys[2*i+1]; gmx_key o = gmx_fold_in(k, data); out[2*i] = o.k0; out[2*i+1] = o.k1; }
  return 0;
}
extern "C" int gmx_random_bits(const uint32_t* keys, int64_t n, int64_t m, uint32_t* out, gmx_stream) {
  for (int64_t i = 0; i < n; ++i) { gmx_key k; k.k0 = keys[2*i]; k.k1 = keys[2*i+1]; for (int64_t j = 0; j < m; ++j) out[i*m+j] = gmx_bits32(k, (uint64_t)j); }
  return 0;
}


// ---- programs ----
struct gmx_program { std::vector<uint32_t> code, consts; uint32_t n_instr, n_regs, n_in, n_out, n_uni, n_tab, n_const, n_dyn; uint32_t n_redmax = 0, n_redlse = 0; bool fuse_rs = false; };

struct HostCtx {
  const uint32_t* code; const gmx_run_args* A;
  std::vector<float>* red; int64_t i; int kind;
  void fetch(uint32_t pc, uint32_t* w0, uint32_t* w1) const { *w0 = code[2 * pc]; *w1 = code[2 * pc + 1]; }
  uint32_t pool(uint32_t k) const { return A->uni[k]; }
  const void* in_ptr(uint32_t s) const { return A->in_d[s]; }
  void* out_ptr(uint32_t s) const { return A->out_d[s]; }
  const void* tab_ptr(uint32_t s) const { return A->tab_d[s]; }
  void red_max(float x, bool active) { kind = 1; (*red)[i] = active ? x : -gmx_inf(); }
  void red_lse(float x, bool active) { kind = 2; (*red)[i] = active ? x : -gmx_inf(); }
};

extern "C" int gmx_program_create(const uint32_t* blob, size_t n_words, gmx_program** out) {
  if (!blob || !out || n_words < GMX_PROG_HEADER_WORDS) return fail("program_create: bad blob");
  if (blob[0] != GMX_PROG_MAGIC || blob[1] != GMX_PROG_VERSION) return fail("program_create: bad magic/version");
  if (n_words != GMX_PROG_HEADER_WORDS + 2ull * blob[2] + blob[8]) return fail("program_create: bad length");
  if (blob[3] == 0 || blob[3] > GMX_MAX_REGS) return fail("program_create: n_regs out of range");
  gmx_program* p = new gmx_program;
  p->n_instr = blob[2]; p->n_regs = blob[3]; p->n_in = blob[4]; p->n_out = blob[5]; p->n_uni = blob[6]; p->n_tab = blob[7];
  p->n_const = blob[8]; p->n_dyn = blob[9];
  p->code.assign(blob + GMX_PROG_HEADER_WORDS, blob + GMX_PROG_HEADER_WORDS + 2ull * blob[2]);
  p->consts.assign(blob + GMX_PROG_HEADER_WORDS + 2ull * blob[2], blob + n_words);
  // the same index validation the HIP library performs
  for (uint32_t pc = 0; pc < p->n_instr; ++pc) {
    uint32_t w0 = p->code[2*pc]; uint32_t op = w0 & 0xff;
    if (op >= OP__COUNT) { delete p; return fail("program_create: invalid opcode"); }
    uint32_t dst = (w0 >> 8) & 0xff, a = (w0 >> 16) & 0xff, b = w0 >> 24;
    bool regdst = !(op == OP_STOUT || op == OP_END || op == OP_REDMAX || op == OP_REDLSE || op == OP_LOOP || op == OP_ENDLOOP);
    if (regdst && dst >= p->n_regs) { delete p; return fail("program_create: register out of range"); }
    (void)a; (void)b;
  }
  for (uint32_t pc = 0; pc < p->n_instr; ++pc) {
    const uint32_t op = p->code[2 * pc] & 0xffu;
    if (op == OP_REDMAX) ++p->n_redmax;
    if (op == OP_REDLSE) ++p->n_redlse;
  }
  *out = p; return 0;
}
extern "C" int gmx_program_destroy(gmx_program* p) { delete p; return 0; }
extern "C" int gmx_program_specialize(gmx_program*) { return fail("hostsim: no specialisation"); }
extern "C" int gmx_program_is_specialized(const gmx_program*) { return 0; }
extern "C" uint64_t gmx_program_code_hash(const gmx_program*) { return 0; }
extern "C" int gmx_program_despecialize(gmx_program*, const char*) { return 0; }        // (never specialised here)
extern "C" int64_t gmx_jit_rejected_count(void) { return 0; }
extern "C" int gmx_program_set_background(gmx_program* p, uint32_t lds_pad) {     // scheduling only: nothing to mirror
  if (!p) return fail("gmx_program_set_background: null program");
  if (lds_pad > 160u * 1024u) return fail("gmx_program_set_background: lds_pad above the 160 KB of a CU");
  return 0;
}
// Like the specialised 4-particles-per-thread kernel, a program with exactly one OP_REDMAX runs in workgroups of
// 1024 particles (one partial row each) and can leave the CDF tile statistics; GENMI_HOSTSIM_TILE_STATS=0 gives the
// interpreter's shape instead (256-particle groups, no statistics).
static bool hs_tile_mode(const gmx_program* p) {
  const char* e = getenv("GENMI_HOSTSIM_TILE_STATS");
  return p && p->n_redmax == 1 && p->n_redlse == 0 && !(e && e[0] == '0');
}
extern "C" int gmx_program_writes_tile_stats(const gmx_program* p) { return hs_tile_mode(p) ? 1 : 0; }
// the resample-first form of a gathering program (gmx_run_args.rs): the mirror resamples, TAGS the ancestors as the
// device leaves them, and gathers through the masked indices
static bool hs_gathers(const gmx_program* p) {
  for (uint32_t pc = 0; pc < p->n_instr; ++pc) {
    const uint32_t w0 = p->code[2 * pc];
    if ((w0 & 0xffu) == OP_LDIN && ((w0 >> 24) & GMX_F_GATHER)) return true;
  }
  return false;
}
extern "C" int gmx_program_set_fuse_resample(gmx_program* p) {
  if (!p) return fail("gmx_program_set_fuse_resample: null program");
  p->fuse_rs = true;
  return 0;
}
extern "C" int gmx_program_set_fuse_shard_step(gmx_program* p) { return p ? 0 : 1; }       // (the mirror keeps the two-launch sharded step)
extern "C" int gmx_program_fuses_shard_step(const gmx_program*) { return 0; }
extern "C" int64_t gmx_program_resident_particles(const gmx_program* p) { return p ? (int64_t)1 << 20 : 0; }   // (the mirror runs workgroups in turn)
extern "C" int gmx_program_set_fuse_resample_loop(gmx_program* p) { return gmx_program_set_fuse_resample(p); }   // (the mirror runs tiles in turn anyway)
extern "C" int gmx_program_fuses_resample(const gmx_program* p) { return p && p->fuse_rs && hs_tile_mode(p) && hs_gathers(p) ? 1 : 0; }
extern "C" int64_t gmx_program_grid(const gmx_program* p, int64_t n) { return hs_tile_mode(p) ? (n + 1023) / 1024 : (n + 255) / 256; }
static uint64_t hs_weight_fixed(float lw, float ref, float scale);

static float butterfly_sum64(const float* v) {
  float t[64]; memcpy(t, v, sizeof(t));
  for (int m = 32; m >= 1; m >>= 1) { float u[64]; for (int l = 0; l < 64; ++l) u[l] = t[l] + t[l ^ m]; memcpy(t, u, sizeof(t)); }
  return t[0];
}

#include "../../genjax_amd/csrc/gmx_peer.h"
extern "C" int gmx_resample_tiles(int kind, const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                  const uint64_t* agg, float* max_d, uint64_t* total, int32_t* anc, gmx_stream st);
static void hs_peer_put_tile(uint64_t* land, uint32_t tag, int world, int tiles, int src, int tile, uint64_t agg, float tmax) {
  uint64_t* row = land + gmx_peer_stats_at(tag, world, tiles, src, tile);
  __atomic_store_n(row + 0, gmx_granule((uint32_t)agg, tag), __ATOMIC_RELAXED);
  __atomic_store_n(row + 1, gmx_granule((uint32_t)(agg >> 32), tag), __ATOMIC_RELAXED);
  __atomic_store_n(row + 2, gmx_granule(gmx_f2u(tmax), tag), __ATOMIC_RELAXED);
}
extern "C" int gmx_program_run(const gmx_program* p, int64_t n, const gmx_run_args* A_in, gmx_stream) {
  if (!p || !A_in) return fail("program_run: null");
  gmx_run_args patched = *A_in;
  for (uint32_t k = 0; k < p->n_const; ++k) patched.uni[p->n_dyn + k] = p->consts[k];
  const gmx_run_args* A = &patched;
  for (uint32_t s = 0; s < p->n_in; ++s) if (!A->in_d[s]) return fail("program_run: null input slot");
  for (uint32_t s = 0; s < p->n_out; ++s) if (!A->out_d[s]) return fail("program_run: null output slot");
  std::vector<int32_t> rs_anc;
  if (A->rs.lw_d) {          // resample the previous step first: gmx_resample_tiles, tagged ancestors, gather through them
    const gmx_resample_in& q = A->rs;
    if (!gmx_program_fuses_resample(p)) return fail("program_run: rs is set but this program cannot resample in its own launch");
    if (!q.tile_max_d || !q.tile_agg_d || !q.max_out_d || !q.total_out_d || !q.status_d || !A->ancestors_d)
      return fail("program_run: rs has a null pointer");
    if (q.tag < 1u || q.tag > 255u) return fail("program_run: rs.tag must be in [1, 255]");
    if ((n + 1023) / 1024 > 1024) return fail("program_run: rs: n <= 2^20");
    if (A->tile_agg_d == q.tile_agg_d || (const float*)A->red_out_d == q.tile_max_d)
      return fail("program_run: rs reads the tile statistics this launch writes (use two sets)");
    for (uint32_t s_ = 0; s_ < p->n_out; ++s_)
      if ((const void*)A->out_d[s_] == (const void*)q.lw_d) return fail("program_run: rs.lw_d is also an output of this launch");
    const uint32_t key[2] = {q.key0, q.key1};
    rs_anc.resize((size_t)n);
    if (gmx_resample_tiles(GMX_RESAMPLE_SYSTEMATIC, key, q.lw_d, n, q.shift, q.tile_max_d, q.tile_agg_d, q.max_out_d,
                           q.total_out_d, rs_anc.data(), nullptr)) return 1;
    int32_t* tagged = const_cast<int32_t*>(A->ancestors_d);
    for (int64_t i = 0; i < n; ++i) tagged[i] = (int32_t)((uint32_t)rs_anc[(size_t)i] | (q.tag << 24));
    patched.ancestors_d = rs_anc.data();       // the program's gathers see the indices
  }
  const bool tile = hs_tile_mode(p);
  if (A->tile_agg_d && !tile) return fail("program_run: tile_agg_d is set but this program cannot write tile statistics");
  const int G = tile ? 1024 : 256;                 // particles per workgroup
  int64_t grid = (n + G - 1) / G;
  std::vector<float> red((size_t)G);
  for (int64_t blk = 0; blk < grid; ++blk) {
    int kind = 0;
    for (int t = 0; t < G; ++t) {
      int64_t i = blk * G + t;
      HostCtx ctx; ctx.code = p->code.data(); ctx.A = A; ctx.red = &red; ctx.i = t; ctx.kind = 0;
      gmx_vm_run<gmx_regs_vgpr<GMX_MAX_REGS>, true, -1, HostCtx>(p->n_instr, i, i < n, *A, ctx);
      if (ctx.kind) kind = ctx.kind;
    }
    if (kind && A->red_out_d) {
      float m = -gmx_inf();
      for (int t = 0; t < G; ++t) m = gmx_rmax(m, red[t]);
      A->red_out_d[blk] = m;
      if (kind == 2) {
        float e[256];
        for (int t = 0; t < 256; ++t) e[t] = (red[t] > -gmx_inf() && m > -gmx_inf()) ? gmx_expf(red[t] - m) : 0.0f;
        float w0 = butterfly_sum64(e), w1 = butterfly_sum64(e + 64), w2 = butterfly_sum64(e + 128), w3 = butterfly_sum64(e + 192);
        A->red_out_d[grid + blk] = (w0 + w1) + (w2 + w3);
      }
      if (tile && A->tile_agg_d) {
        const float ref = gmx_tile_ref(gmx_tile_exp(m));
        uint64_t sum = 0;
        for (int t = 0; t < G; ++t) sum += gmx_exp_fixed(red[t] - ref, A->tile_shift);
        A->tile_agg_d[blk] = sum;
        if (A->peer.land_d) {        // the epilogue's put: this tile's statistics into every other rank's landing table
          const gmx_peer& P = A->peer;
          const uint32_t tag = *P.tag_base_d + (uint32_t)P.step;
          for (int d = 0; d < P.world; ++d)
            if (d != P.rank) hs_peer_put_tile((uint64_t*)P.land_d[d], tag, P.world, P.tiles, P.rank, (int)blk, sum, m);
        }
      }
    }
  }
  return 0;
}

// ---- logsumexp (sequential; float tolerance vs the device tree) ----
extern "C" size_t gmx_logsumexp_workspace(int64_t, int64_t) { return 16; }
extern "C" int gmx_logsumexp(const float* lw, int64_t rows, int64_t cols, float* out, float* out_max, void*, gmx_stream) {
  for (int64_t r = 0; r < rows; ++r) {
    const float* x = lw + r * cols;
    float m = -gmx_inf();
    for (int64_t j = 0; j < cols; ++j) m = gmx_rmax(m, x[j]);
    float s = 0.0f;
    if (m > -gmx_inf()) for (int64_t j = 0; j < cols; ++j) s += gmx_expf(x[j] - m);
    out[r] = (m > -gmx_inf()) ? m + gmx_logf(s) : m;
    if (out_max) out_max[r] = m;
  }
  return 0;
}

// ---- row sums in the device's fixed tree (gmx_sum_rows) ----
static float hs_block_sum(const float* v256) {
  const float w0 = butterfly_sum64(v256), w1 = butterfly_sum64(v256 + 64), w2 = butterfly_sum64(v256 + 128), w3 = butterfly_sum64(v256 + 192);
  return (w0 + w1) + (w2 + w3);
}
extern "C" int gmx_sum_rows_inorder(const float* x, int64_t rows, int64_t cols, int64_t sr, int64_t sc, float* out, gmx_stream) {
  if (rows <= 0) return 0;
  if (cols < 0 || !x || !out) return fail("sum_rows_inorder: bad argument");
  for (int64_t r = 0; r < rows; ++r) {
    float acc = 0.0f;
    for (int64_t c = 0; c < cols; ++c) acc += x[r * sr + c * sc];
    out[r] = acc;
  }
  return 0;
}
extern "C" size_t gmx_sum_rows_workspace(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 16;
  return (size_t)(rows * ((cols + 4095) / 4096) * sizeof(float)) + 16;
}
extern "C" int gmx_sum_rows(const float* x, int64_t rows, int64_t cols, float* out, void* ws, gmx_stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || !x || !out || !ws) return fail("sum_rows: bad argument");
  const int64_t tiles = (cols + 4095) / 4096;
  float* part = (float*)ws;
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t tile = 0; tile < tiles; ++tile) {
      float v[256];
      for (int t = 0; t < 256; ++t) {
        float s = 0.0f;
        for (int k = 0; k < 16; ++k) { const int64_t j = tile * 4096 + (int64_t)k * 256 + t; s += j < cols ? x[r * cols + j] : 0.0f; }
        v[t] = s;
      }
      part[r * tiles + tile] = hs_block_sum(v);
    }
    float v[256];
    for (int t = 0; t < 256; ++t) { float s = 0.0f; for (int64_t j = t; j < tiles; j += 256) s += part[r * tiles + j]; v[t] = s; }
    out[r] = hs_block_sum(v);
  }
  return 0;
}
// ---- weights / cdf / ancestors ----
extern "C" int gmx_ancestors(int kind, const uint32_t key[2], const uint64_t* cdf, int64_t n_in, uint64_t off,
                             const uint64_t* total_d, int64_t n_out_total, int64_t slot_offset, int64_t n_slots,
                             int32_t* anc, gmx_stream);
extern "C" int gmx_reduce_max(const float* parts, int64_t n, float* max_d, gmx_stream) {
  float m = -gmx_inf(); for (int64_t j = 0; j < n; ++j) m = gmx_rmax(m, parts[j]); *max_d = m; return 0;
}
extern "C" size_t gmx_weight_cdf_workspace(int64_t n) { return 8 + (size_t)((n + 1023) / 1024) * 8; }
// the two-level integer CDF (include/genmi.h "Resampling"): tiles of 1024 consecutive indices
static const int64_t HS_TILE = 1024;
// the device's own function (csrc/gmx_math.h): the mirror checks it against the oracle's floor(exp(.) * 2^shift)
static uint64_t hs_weight_fixed(float lw, float ref, float scale) {
  return gmx_exp_fixed(lw - ref, (int)(gmx_f2u(scale) >> 23) - 127);
}
extern "C" int gmx_weight_cdf(const float* lw, int64_t n, int shift, const float* parts, int64_t n_parts, float* max_d,
                              uint64_t* cdf, uint64_t* total, void*, gmx_stream) {
  if (n <= 0) return fail("weight_cdf: n");
  int need = 0; while (((int64_t)1 << need) < n) ++need;
  if (shift + need > 62) return fail("weight_cdf: shift too large");
  if (parts) { float m = -gmx_inf(); for (int64_t j = 0; j < n_parts; ++j) m = gmx_rmax(m, parts[j]); *max_d = m; }
  const float M = *max_d, scale = gmx_pow2i(shift);
  const int32_t K = gmx_tile_exp(M);
  uint64_t prefix = 0;
  for (int64_t lo = 0; lo < n; lo += HS_TILE) {
    const int64_t hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    float m = -gmx_inf();
    for (int64_t i = lo; i < hi; ++i) m = gmx_rmax(m, lw[i]);
    const int32_t k = gmx_tile_exp(m);
    const float ref = gmx_tile_ref(k);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) {
      run += hs_weight_fixed(lw[i], ref, scale);
      cdf[i] = prefix + gmx_tile_scale(run, k, K);
    }
    prefix += gmx_tile_scale(run, k, K);
  }
  *total = prefix;
  return 0;
}
typedef unsigned __int128 u128;
extern "C" int gmx_ancestors(int kind, const uint32_t key[2], const uint64_t* cdf, int64_t n_in, uint64_t off,
                             const uint64_t* total_d, int64_t n_out_total, int64_t slot_offset, int64_t n_slots,
                             int32_t* anc, gmx_stream) {
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  uint64_t total = *total_d;
  for (int64_t s = 0; s < n_slots; ++s) {
    int64_t j = slot_offset + s;
    u128 D, P;
    if (kind == GMX_RESAMPLE_MULTINOMIAL) {
      uint64_t u = gmx_bits32(k, (uint64_t)j) >> 9;
      D = (u128)1 << 23; P = (u128)total * (((uint64_t)1 << 23) - u);
      if (P > 0) P -= 1;
    } else {
      uint64_t u = (kind == GMX_RESAMPLE_SYSTEMATIC) ? (gmx_bits32(k, 0) >> 9) : (gmx_bits32(k, (uint64_t)j) >> 9);
      D = (u128)((uint64_t)n_out_total << 23); P = (u128)(((uint64_t)j << 23) + u) * total;
    }
    int64_t lo = 0, hi = n_in;
    while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if ((u128)(cdf[mid] + off) * D > P) hi = mid; else lo = mid + 1; }
    if (lo >= n_in) lo = n_in - 1;
    anc[s] = (int32_t)lo;
  }
  return 0;
}
extern "C" size_t gmx_resample_workspace(int64_t) { return 2048 * 12; }
extern "C" int gmx_tile_stats(const float* lw, int64_t n, int shift, float* tmax, uint64_t* agg, gmx_stream) {
  if (n <= 0 || !lw || !tmax || !agg) return fail("tile_stats: bad argument");
  const float scale = gmx_pow2i(shift);
  for (int64_t lo = 0, b = 0; lo < n; lo += HS_TILE, ++b) {
    const int64_t hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    float m = -gmx_inf();
    for (int64_t i = lo; i < hi; ++i) m = gmx_rmax(m, lw[i]);
    const float ref = gmx_tile_ref(gmx_tile_exp(m));
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) run += hs_weight_fixed(lw[i], ref, scale);
    tmax[b] = m; agg[b] = run;
  }
  return 0;
}
extern "C" size_t gmx_multinomial_workspace(int64_t n_in) { return (size_t)(n_in + 4) * 4; }
// the mirror states the definition: the per-slot search of gmx_ancestors
extern "C" int gmx_multinomial(const uint32_t key[2], const uint64_t* cdf, int64_t n_in, const uint64_t* total, int64_t n_out,
                               int32_t* anc, void* ws, gmx_stream st) {
  if (!ws) return fail("multinomial: null workspace");
  return gmx_ancestors(GMX_RESAMPLE_MULTINOMIAL, key, cdf, n_in, 0, total, n_out, 0, n_out, anc, st);
}
// the mirror rebuilds the global CDF from log-weights + tile stats and searches it per slot
extern "C" int gmx_resample_tiles(int kind, const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                  const uint64_t* agg, float* max_d, uint64_t* total, int32_t* anc, gmx_stream st) {
  if (kind == GMX_RESAMPLE_MULTINOMIAL) return fail("resample: kind");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE;
  float M = -gmx_inf();
  for (int64_t b = 0; b < tiles; ++b) M = gmx_rmax(M, tmax[b]);
  const float scale = gmx_pow2i(shift);
  const int32_t K = gmx_tile_exp(M);
  std::vector<uint64_t> cdf((size_t)n);
  uint64_t prefix = 0;
  for (int64_t b = 0; b < tiles; ++b) {
    const int64_t lo = b * HS_TILE, hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    const int32_t k = gmx_tile_exp(tmax[b]);
    const float ref = gmx_tile_ref(k);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) {
      run += hs_weight_fixed(lw[i], ref, scale);
      cdf[(size_t)i] = prefix + gmx_tile_scale(run, k, K);
    }
    prefix += gmx_tile_scale(agg[b], k, K);
  }
  *max_d = M; *total = prefix;
  return gmx_ancestors(kind, key, cdf.data(), n, 0, total, n, 0, n, anc, st);
}
// tile statistics -> tile prefixes (sequential statement of gmx_block.h: gmx_tile_prefix_block)
extern "C" size_t gmx_tile_prefix_words(int64_t n) {       // prefixes | total | M, K | pad to 16
  return (size_t)(((n + HS_TILE - 1) / HS_TILE + 2 + 15) / 16) * 16;
}
extern "C" int gmx_tile_prefix(const float* tmax, const uint64_t* agg, int64_t n, uint64_t* pref, gmx_stream) {
  if (n <= 0 || !tmax || !agg || !pref) return fail("tile_prefix: bad argument");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE;
  if (n > 0x7fffffffLL) return fail("tile_prefix: n out of range");
  float M = -gmx_inf();
  for (int64_t b = 0; b < tiles; ++b) M = gmx_rmax(M, tmax[b]);
  const int32_t K = gmx_tile_exp(M);
  uint64_t run = 0;
  for (int64_t b = 0; b < tiles; ++b) { pref[b] = run; run += gmx_tile_scale(agg[b], gmx_tile_exp(tmax[b]), K); }
  pref[tiles] = run;
  pref[tiles + 1] = (uint64_t)gmx_f2u(M) | ((uint64_t)(uint32_t)K << 32);
  return 0;
}
// gmx_resample_tiles reading the prefixes instead of reducing the table: the mirror checks them against the
// statistics-free definition (the CDF rebuilt from the log-weights alone) on the way
extern "C" int gmx_resample_tiles_p(int kind, const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                    const uint64_t* pref, float* max_d, uint64_t* total, int32_t* anc, gmx_stream st) {
  if (kind == GMX_RESAMPLE_MULTINOMIAL) return fail("resample: kind");
  if (!lw || !tmax || !pref || !max_d || !total || !anc) return fail("resample_tiles_p: bad argument");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE;
  const float M = gmx_u2f((uint32_t)pref[tiles + 1]);
  const int32_t K = (int32_t)(uint32_t)(pref[tiles + 1] >> 32);
  if (K != gmx_tile_exp(M)) return fail("resample_tiles_p: (M, K) of the prefix block disagree");
  const float scale = gmx_pow2i(shift);
  std::vector<uint64_t> cdf((size_t)n);
  for (int64_t b = 0; b < tiles; ++b) {
    const int64_t lo = b * HS_TILE, hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    const int32_t k = gmx_tile_exp(tmax[b]);
    const float ref = gmx_tile_ref(k);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) {
      run += hs_weight_fixed(lw[i], ref, scale);
      cdf[(size_t)i] = pref[b] + gmx_tile_scale(run, k, K);
    }
    const uint64_t next = pref[b] + gmx_tile_scale(run, k, K);
    if (next != pref[b + 1]) return fail("resample_tiles_p: the prefixes do not match the log-weights");
  }
  *max_d = M; *total = pref[tiles];
  return gmx_ancestors(kind, key, cdf.data(), n, 0, total, n, 0, n, anc, st);
}
// two-stage multinomial resampling (include/genmi.h: gmx_multinomial_tiled) — the sequential statement
extern "C" size_t gmx_multinomial_tiled_workspace(int64_t n) { return (size_t)((n + HS_TILE - 1) / HS_TILE + 16) * 8; }
extern "C" int gmx_multinomial_tiled(const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                     const uint64_t* agg, const uint32_t* u_d, float* max_d, uint64_t* total_d,
                                     int32_t* anc, void* ws, int phase, gmx_stream) {
  if (!key || !lw || !tmax || !agg || !max_d || !total_d || !anc || !ws || n <= 0 || shift < 1 || phase < -1 || phase > 1)
    return fail("multinomial_tiled: bad argument");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE;
  // the two-buffer protocol of the device entry, checked: the buffer of this phase must arrive zero
  uint32_t* buf = (uint32_t*)ws + (phase == 1 ? tiles + 16 : 0);
  uint32_t* other = (uint32_t*)ws + (phase == 1 ? 0 : tiles + 16);
  if (phase >= 0)
    for (int64_t b = 0; b < tiles; ++b)
      if (buf[b]) return fail("multinomial_tiled: the count buffer of this phase is not zero (alternate the phases)");
  struct Leave { uint32_t* o; int64_t t; ~Leave() { for (int64_t b = 0; b < t; ++b) o[b] = 0; } } leave{other, tiles};
  float M = -gmx_inf();
  for (int64_t b = 0; b < tiles; ++b) M = gmx_rmax(M, tmax[b]);
  const int32_t K = gmx_tile_exp(M);
  std::vector<uint64_t> cend((size_t)tiles), G((size_t)tiles);
  uint64_t total = 0;
  for (int64_t b = 0; b < tiles; ++b) { G[b] = gmx_tile_scale(agg[b], gmx_tile_exp(tmax[b]), K); total += G[b]; cend[b] = total; }
  *max_d = M; *total_d = total;
  if (total == 0) { for (int64_t i = 0; i < n; ++i) anc[i] = (int32_t)(n - 1); return 0; }
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  const gmx_key k1 = gmx_split_child(k, 0), k2 = gmx_split_child(k, 1);
  uint32_t* counts = buf;
  for (int64_t b = 0; b < tiles; ++b) counts[b] = 0;
  for (int64_t j = 0; j < n; ++j) {
    const uint32_t u = gmx_bits32(k1, (uint64_t)j) >> 9;
    if (u_d && u_d[j] != u) return fail("multinomial_tiled: u_d is not split(key, 2)[0]'s slot uniforms");
    const uint64_t P = (uint64_t)(((u128)u * (u128)total) >> 23);
    int64_t lo = 0, hi = tiles;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (cend[mid] > P) hi = mid; else lo = mid + 1; }
    ++counts[lo];
  }
  const float scale = gmx_pow2i(shift);
  int64_t pos = 0;
  for (int64_t b = 0; b < tiles; ++b) {
    const int64_t lo_i = b * HS_TILE, hi_i = lo_i + HS_TILE < n ? lo_i + HS_TILE : n;
    const int32_t kb_e = gmx_tile_exp(tmax[b]);
    const float ref = gmx_tile_ref(kb_e);
    std::vector<uint64_t> loc((size_t)(hi_i - lo_i));
    uint64_t run = 0;
    for (int64_t i = lo_i; i < hi_i; ++i) { run += hs_weight_fixed(lw[i], ref, scale); loc[(size_t)(i - lo_i)] = gmx_tile_scale(run, kb_e, K); }
    const gmx_key kb = gmx_fold_in(k2, (uint32_t)b);
    for (uint32_t r = 0; r < counts[b]; ++r) {
      const uint32_t v = gmx_bits32(kb, (uint64_t)r) >> 9;
      const uint64_t Q = (uint64_t)(((u128)v * (u128)G[b]) >> 23);
      int64_t a = 0, z = hi_i - lo_i;
      while (a < z) { int64_t mid = (a + z) >> 1; if (loc[(size_t)mid] > Q) z = mid; else a = mid + 1; }
      anc[pos++] = (int32_t)(lo_i + a);
    }
  }
  if (pos != n) return fail("multinomial_tiled: the tile counts do not add up to n");
  return 0;
}
// the stratified resampler's uniforms ahead of time, and the resampler that reads them (the mirror checks that what it
// is handed IS what the resampling would draw, then resamples as usual)
extern "C" int gmx_slot_uniforms(const uint32_t* keys, int rows, int64_t n, uint32_t* out, int lds_pad, gmx_stream) {
  if (!keys || !out || rows < 1 || n <= 0 || lds_pad < 0) return fail("slot_uniforms: bad argument");
  for (int r = 0; r < rows; ++r) {
    gmx_key k; k.k0 = keys[2 * r]; k.k1 = keys[2 * r + 1];
    for (int64_t j = 0; j < n; ++j) out[(int64_t)r * n + j] = gmx_bits32(k, (uint64_t)j) >> 9;
  }
  return 0;
}
extern "C" int gmx_resample_tiles_u(int kind, const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                    const uint64_t* agg, const uint32_t* u, float* max_d, uint64_t* total, int32_t* anc,
                                    gmx_stream st) {
  if (kind != GMX_RESAMPLE_STRATIFIED) return fail("resample_tiles_u: stratified only");
  if (!key || !u) return fail("resample_tiles_u: bad argument");
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  for (int64_t j = 0; j < n; ++j)
    if (u[j] != (gmx_bits32(k, (uint64_t)j) >> 9)) return fail("resample_tiles_u: u_d is not this key's slot uniforms");
  return gmx_resample_tiles(kind, key, lw, n, shift, tmax, agg, max_d, total, anc, st);
}
// multinomial resampling with sorted uniforms (csrc/gmx_sorted.h): the table from its definition, sequentially; the
// resampler rebuilds the CDF and merges it with the sums read back from the TABLE (so a wrong table gives wrong ancestors),
// after checking the table against the key when it was handed one
#include "../../genjax_amd/csrc/gmx_sorted.h"
extern "C" size_t gmx_sorted_uniforms_words(int64_t n) { return n > 0 ? gmx_sorted_layout_of(n).words : 0; }
static void hs_sorted_row(gmx_key k, int64_t n, uint32_t* row) {
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  uint64_t* tsum = (uint64_t*)(row + L.off_tsum);
  uint64_t* toff = (uint64_t*)(row + L.off_toff);
  std::vector<uint64_t> S((size_t)n);
  uint64_t run = 0;
  for (int64_t t = 0; t < L.tiles; ++t) tsum[t] = 0;
  for (int64_t j = 0; j < n; ++j) {
    if (j % GMX_SORTED_TILE == 0) toff[j / GMX_SORTED_TILE] = run;
    const uint32_t e = gmx_sorted_exp(k, (uint64_t)j);
    run += e; tsum[j / GMX_SORTED_TILE] += e;
    S[(size_t)j] = run;
    row[j] = (uint32_t)run;
  }
  for (int64_t j = n; j < L.tiles * GMX_SORTED_TILE; ++j) row[j] = 0;
  const uint64_t stot = run + gmx_sorted_exp(k, (uint64_t)n);
  toff[L.tiles] = stot;
  const uint32_t sh = gmx_sorted_shift(stot, L.ng);
  row[L.off_sh] = sh;
  uint32_t* guide = row + L.off_guide;
  int64_t j = 0;
  for (int64_t g = 0; g <= (int64_t)(stot >> sh) + 1; ++g) {        // guide[g] = #{ j : (S_j >> sh) < g }
    while (j < n && (int64_t)(S[(size_t)j] >> sh) < g) ++j;
    guide[g] = (uint32_t)j;
  }
}
extern "C" int gmx_sorted_uniforms(const uint32_t* keys, int rows, int64_t n, uint32_t* out, int lds_pad, gmx_stream) {
  if (!keys || !out || rows < 1 || n <= 0 || lds_pad < 0) return fail("sorted_uniforms: bad argument");
  const size_t words = gmx_sorted_layout_of(n).words;
  for (int r = 0; r < rows; ++r) {
    gmx_key k; k.k0 = keys[2 * r]; k.k1 = keys[2 * r + 1];
    hs_sorted_row(k, n, out + (size_t)r * words);
  }
  return 0;
}
// agg: the tile statistics A_b — or (prefixes = true) the prefix block of gmx_tile_prefix
static int hs_resample_sorted(const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                              const uint64_t* agg, bool prefixes, uint32_t* table, int table_ready, float* max_d, uint64_t* total,
                              int32_t* anc) {
  if (!key || !lw || !tmax || !agg || !table || !max_d || !total || !anc || n <= 0 || (!prefixes && n > 2048 * HS_TILE))
    return fail("resample_sorted: bad argument");
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  if (table_ready) {
    std::vector<uint32_t> want(L.words, 0u);
    hs_sorted_row(k, n, want.data());
    const uint64_t stot = ((const uint64_t*)(want.data() + L.off_toff))[L.tiles];
    const size_t gtop = (size_t)(stot >> want[L.off_sh]) + 1;
    if (memcmp(want.data(), table, (size_t)L.tiles * GMX_SORTED_TILE * 4) ||
        memcmp(want.data() + L.off_guide, table + L.off_guide, (gtop + 1) * 4) ||
        memcmp(want.data() + L.off_toff, table + L.off_toff, ((size_t)L.tiles + 1) * 8) || want[L.off_sh] != table[L.off_sh])
      return fail("resample_sorted: table_d is not this key's order-statistics table");
  } else {
    hs_sorted_row(k, n, table);
  }
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE;
  float M = -gmx_inf();
  for (int64_t b = 0; b < tiles; ++b) M = gmx_rmax(M, tmax[b]);
  const float scale = gmx_pow2i(shift);
  const int32_t K = gmx_tile_exp(M);
  std::vector<uint64_t> cdf((size_t)n);
  uint64_t prefix = 0;
  for (int64_t b = 0; b < tiles; ++b) {
    const int64_t lo = b * HS_TILE, hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    const int32_t kb = gmx_tile_exp(tmax[b]);
    const float ref = gmx_tile_ref(kb);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) {
      run += hs_weight_fixed(lw[i], ref, scale);
      cdf[(size_t)i] = prefix + gmx_tile_scale(run, kb, K);
    }
    prefix = prefixes ? (b + 1 < tiles ? agg[b + 1] : agg[tiles]) : prefix + gmx_tile_scale(agg[b], kb, K);
  }
  *max_d = M; *total = prefix;
  if (prefix == 0) { for (int64_t j = 0; j < n; ++j) anc[j] = (int32_t)(n - 1); return 0; }
  const uint64_t* toff = (const uint64_t*)(table + L.off_toff);
  const uint64_t stot = toff[L.tiles];
  int64_t i = 0;
  for (int64_t j = 0; j < n; ++j) {
    const uint64_t off = toff[j / GMX_SORTED_TILE];
    const uint64_t S = off + (uint64_t)(uint32_t)(table[j] - (uint32_t)off);
    const u128 P = (u128)S * (u128)prefix;
    while (i < n - 1 && !((u128)cdf[(size_t)i] * (u128)stot > P)) ++i;
    anc[j] = (int32_t)i;
  }
  return 0;
}
extern "C" int gmx_resample_sorted(const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                   const uint64_t* agg, uint32_t* table, int table_ready, float* max_d, uint64_t* total,
                                   int32_t* anc, gmx_stream) {
  return hs_resample_sorted(key, lw, n, shift, tmax, agg, false, table, table_ready, max_d, total, anc);
}
extern "C" int gmx_resample_sorted_p(const uint32_t key[2], const float* lw, int64_t n, int shift, const float* tmax,
                                     const uint64_t* pref, uint32_t* table, int table_ready, float* max_d, uint64_t* total,
                                     int32_t* anc, gmx_stream) {
  return hs_resample_sorted(key, lw, n, shift, tmax, pref, true, table, table_ready, max_d, total, anc);
}
extern "C" int gmx_resample(int kind, const uint32_t key[2], const float* lw, int64_t n, int shift, const float*,
                            int64_t, float* max_d, uint64_t* total, int32_t* anc, void* ws, gmx_stream st) {
  uint64_t* agg = (uint64_t*)ws; float* tmax = (float*)(agg + 2048);
  if (gmx_tile_stats(lw, n, shift, tmax, agg, st)) return 1;
  return gmx_resample_tiles(kind, key, lw, n, shift, tmax, agg, max_d, total, anc, st);
}
// ---- global resampling across ranks: destination-centric restatement ----
static int64_t hs_slots_below(int kind, gmx_key k, uint64_t c, uint64_t total, int64_t N) {
  // #{ j : (j*2^23 + u_j) * total < c * N * 2^23 }  (thresholds increase with j)
  if (total == 0) return 0;
  u128 X = (u128)c * (u128)((uint64_t)N << 23);
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    uint64_t u = (kind == GMX_RESAMPLE_SYSTEMATIC) ? (gmx_bits32(k, 0) >> 9) : (gmx_bits32(k, (uint64_t)mid) >> 9);
    u128 P = (u128)(((uint64_t)mid << 23) + u) * total;
    if (P < X) lo = mid + 1; else hi = mid;
  }
  return lo;
}
extern "C" size_t gmx_shard_plan_words(int world) { return (size_t)(GMX_PLAN_BOUNDS + world + 1); }
extern "C" int gmx_shard_plan(int kind, const uint32_t key[2], const uint64_t* totals, int rank, int world, int64_t n,
                              int64_t* plan, uint64_t* total_out, gmx_stream) {
  if (kind != GMX_RESAMPLE_SYSTEMATIC && kind != GMX_RESAMPLE_STRATIFIED) return fail("shard_plan: kind");
  gmx_key k; k.k0 = key[0]; k.k1 = key[1];
  int64_t N = n * world;
  uint64_t total = 0, off = 0;
  for (int s = 0; s < world; ++s) total += totals[s];
  for (int s = 0; s < world; ++s) {
    if (s == rank) plan[GMX_PLAN_OFFSET] = (int64_t)off;
    plan[GMX_PLAN_BOUNDS + s] = hs_slots_below(kind, k, off, total, N);
    off += totals[s];
  }
  plan[GMX_PLAN_BOUNDS + world] = N;
  plan[GMX_PLAN_TOTAL] = (int64_t)total;
  if (total_out) *total_out = total;
  return 0;
}
extern "C" int gmx_shard_route(int kind, const uint32_t key[2], int64_t* plan, const uint64_t* cdf, int rank, int world,
                               int64_t n, int64_t cap, const void* state_v, void* send_v, int32_t* next_idx, gmx_stream st) {
  if (cap < 1 || cap > n) return fail("shard_route: capacity");
  const uint32_t* state = (const uint32_t*)state_v; uint32_t* send = (uint32_t*)send_v;
  const int64_t* bounds = plan + GMX_PLAN_BOUNDS;
  const int64_t N = n * world, base = (int64_t)rank * n, S = bounds[rank], E = bounds[rank + 1];
  uint64_t total = (uint64_t)plan[GMX_PLAN_TOTAL];
  // slots whose ancestor is here: per-slot binary search (gmx_ancestors), then route
  for (int64_t j = S; j < E; ++j) {
    int32_t a;
    if (gmx_ancestors(kind, key, cdf, n, (uint64_t)plan[GMX_PLAN_OFFSET], &total, N, j, 1, &a, st)) return 1;
    int64_t d = j / n;
    if (d == rank) next_idx[j - base] = a;
    else {
      int64_t first = S > d * n ? S : d * n, k = j - first;
      if (k < cap) send[d * cap + k] = state[a]; else plan[GMX_PLAN_OVERFLOW] = 1;
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    int64_t jj = base + i;
    int s = 0; while (s + 1 < world && bounds[s + 1] <= jj) ++s;
    if (s == rank) continue;
    int64_t first = bounds[s] > base ? bounds[s] : base, k = jj - first;
    if (k < cap) next_idx[i] = (int32_t)(n + (int64_t)s * cap + k); else { next_idx[i] = 0; plan[GMX_PLAN_OVERFLOW] = 1; }
  }
  return 0;
}
extern "C" int gmx_shard_step(int kind, const uint32_t key[2], const uint64_t* totals, int64_t* plan, uint64_t* total_out,
                              const uint64_t* cdf, int rank, int world, int64_t n, int64_t cap, const void* state,
                              void* send, int32_t* next_idx, gmx_stream st) {
  int64_t flag = plan[GMX_PLAN_OVERFLOW];
  if (gmx_shard_plan(kind, key, totals, rank, world, n, plan, total_out, st)) return 1;
  plan[GMX_PLAN_OVERFLOW] = flag;
  return gmx_shard_route(kind, key, plan, cdf, rank, world, n, cap, state, send, next_idx, st);
}
// sorted multinomial across ranks, destination-centric: slot j of the N global slots sits at S_j / S_total; its ancestor
// is the first GLOBAL source i with cdf_i * S_total > S_j * total
extern "C" int gmx_shard_step_sorted(const uint32_t* table, const uint64_t* totals, int64_t* plan, uint64_t* total_out,
                                     const uint64_t* cdf, int rank, int world, int64_t n, int64_t cap,
                                     const void* state_v, void* send_v, int32_t* next_idx, gmx_stream) {
  if (cap < 1 || cap > n) return fail("shard_step_sorted: capacity");
  const int64_t N = n * world, base = (int64_t)rank * n;
  const gmx_sorted_layout L = gmx_sorted_layout_of(N);
  const uint64_t* toff = (const uint64_t*)(table + L.off_toff);
  const uint64_t stot = toff[L.tiles];
  auto S_of = [&](int64_t j) { const uint64_t off = toff[j / GMX_SORTED_TILE]; return off + (uint64_t)(uint32_t)(table[j] - (uint32_t)off); };
  uint64_t total = 0, off = 0, mine = 0;
  for (int s = 0; s < world; ++s) total += totals[s];
  auto below = [&](uint64_t c) {           // #{ j : S_j * total < c * S_total }
    if (total == 0) return (int64_t)0;
    int64_t lo = 0, hi = N;
    while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if ((u128)S_of(mid) * total < (u128)c * stot) lo = mid + 1; else hi = mid; }
    return lo;
  };
  for (int s = 0; s < world; ++s) {
    if (s == rank) mine = off;
    plan[GMX_PLAN_BOUNDS + s] = below(off);
    off += totals[s];
  }
  plan[GMX_PLAN_BOUNDS + world] = N; plan[GMX_PLAN_TOTAL] = (int64_t)total; plan[GMX_PLAN_OFFSET] = (int64_t)mine;
  if (total_out) *total_out = total;
  const int64_t* bounds = plan + GMX_PLAN_BOUNDS;
  const uint32_t* state = (const uint32_t*)state_v; uint32_t* send = (uint32_t*)send_v;
  const int64_t S = bounds[rank], E = bounds[rank + 1];
  int64_t i = 0;
  for (int64_t j = S; j < E; ++j) {
    if (total == 0) i = n - 1;
    else { const u128 P = (u128)S_of(j) * total; while (i < n - 1 && !((u128)(cdf[i] + mine) * stot > P)) ++i; }
    int64_t d = j / n;
    if (d == rank) next_idx[j - base] = (int32_t)i;
    else { int64_t first = S > d * n ? S : d * n, k = j - first; if (k < cap) send[d * cap + k] = state[i]; else plan[GMX_PLAN_OVERFLOW] = 1; }
  }
  for (int64_t q = 0; q < n; ++q) {
    int64_t jj = base + q;
    int s = 0; while (s + 1 < world && bounds[s + 1] <= jj) ++s;
    if (s == rank) continue;
    int64_t first = bounds[s] > base ? bounds[s] : base, k = jj - first;
    if (k < cap) next_idx[q] = (int32_t)(n + (int64_t)s * cap + k); else { next_idx[q] = 0; plan[GMX_PLAN_OVERFLOW] = 1; }
  }
  return 0;
}
extern "C" size_t gmx_shard_stats_bytes(int64_t n) { int64_t t = (n + HS_TILE - 1) / HS_TILE; t += t & 1; return (size_t)t * 12; }
extern "C" int gmx_shard_totals(const void* stats_all, int world, int64_t n, uint64_t* totals, float* max_d, gmx_stream) {
  if (!stats_all || !totals || !max_d) return fail("shard_totals: null argument");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE, pad = tiles + (tiles & 1);
  const size_t stride = gmx_shard_stats_bytes(n);
  float M = -gmx_inf();
  for (int r = 0; r < world; ++r) {
    const float* tm = (const float*)((const uint8_t*)stats_all + r * stride + pad * 8);
    for (int64_t t = 0; t < tiles; ++t) M = gmx_rmax(M, tm[t]);
  }
  const int32_t K = gmx_tile_exp(M);
  for (int r = 0; r < world; ++r) {
    const uint64_t* ag = (const uint64_t*)((const uint8_t*)stats_all + r * stride);
    const float* tm = (const float*)((const uint8_t*)stats_all + r * stride + pad * 8);
    uint64_t sum = 0;
    for (int64_t t = 0; t < tiles; ++t) sum += gmx_tile_scale(ag[t], gmx_tile_exp(tm[t]), K);
    totals[r] = sum;
  }
  *max_d = M;
  return 0;
}
// the mirror rebuilds this rank's CDF array from log-weights + its statistics and runs the array form
extern "C" int gmx_shard_step_tiles(int kind, const uint32_t key[2], const uint64_t* totals, int64_t* plan, uint64_t* total_out,
                                    const float* lw, const void* stats_own, const float* max_d, int shift, int rank, int world,
                                    int64_t n, int64_t cap, const void* state, void* send, int32_t* next_idx, gmx_stream st) {
  if (!lw || !stats_own || !max_d) return fail("shard_step_tiles: null argument");
  const int64_t tiles = (n + HS_TILE - 1) / HS_TILE, pad = tiles + (tiles & 1);
  const uint64_t* ag = (const uint64_t*)stats_own;
  const float* tm = (const float*)((const uint8_t*)stats_own + pad * 8);
  const int32_t K = gmx_tile_exp(*max_d);
  const float scale = gmx_pow2i(shift);
  std::vector<uint64_t> cdf((size_t)n);
  uint64_t prefix = 0;
  for (int64_t b = 0; b < tiles; ++b) {
    const int64_t lo = b * HS_TILE, hi = lo + HS_TILE < n ? lo + HS_TILE : n;
    const int32_t k = gmx_tile_exp(tm[b]);
    const float ref = gmx_tile_ref(k);
    uint64_t run = 0;
    for (int64_t i = lo; i < hi; ++i) { run += hs_weight_fixed(lw[i], ref, scale); cdf[(size_t)i] = prefix + gmx_tile_scale(run, k, K); }
    prefix += gmx_tile_scale(ag[b], k, K);
  }
  return gmx_shard_step(kind, key, totals, plan, total_out, cdf.data(), rank, world, n, cap, state, send, next_idx, st);
}
// the one-launch form: the totals from the gathered table, then the step with this rank's own block of it
extern "C" int gmx_shard_step_fused(int kind, const uint32_t key[2], const void* stats_all, int64_t* plan, uint64_t* total_out,
                                    const float* lw, float* max_out, int shift, int rank, int world, int64_t n, int64_t cap,
                                    const void* state, void* send, int32_t* next_idx, gmx_stream st) {
  if (!stats_all || !max_out || world < 1 || world > 64) return fail("shard_step_fused: bad argument");
  std::vector<uint64_t> totals((size_t)world);
  if (gmx_shard_totals(stats_all, world, n, totals.data(), max_out, st)) return 1;
  const void* own = (const uint8_t*)stats_all + (size_t)rank * gmx_shard_stats_bytes(n);
  return gmx_shard_step_tiles(kind, key, totals.data(), plan, total_out, lw, own, max_out, shift, rank, world, n, cap, state,
                              send, next_idx, st);
}
// ---- fused peer exchange (include/genmi.h): granules in process-shared landing blocks; every wait is bounded ----
#include <sched.h>
#include <time.h>
extern "C" size_t gmx_peer_landing_bytes(int world, int64_t n, int64_t cap, int leaves) {
  if (world < 1 || n < 1 || cap < 1 || leaves < 1) return 0;
  return (gmx_peer_stats_words(world, (int)((n + HS_TILE - 1) / HS_TILE)) + gmx_peer_state_words(world, cap, leaves)) * 8;
}
extern "C" int gmx_peer_bump(uint32_t* tag_base, int32_t T, gmx_stream) {
  if (!tag_base || T < 1) return fail("peer_bump: bad argument");
  *tag_base += (uint32_t)T;
  return 0;
}
extern "C" int gmx_sweep_verdict(const int64_t* overflow, const uint64_t* const* status, int32_t n_status, int64_t* verdict,
                                 gmx_stream) {
  if (!overflow || !verdict || n_status < 0 || n_status > 4 || (n_status && !status)) return fail("sweep_verdict: bad argument");
  bool bad = false;
  for (int k = 0; k < n_status; ++k) bad = bad || (status[k] && *status[k] != 0);
  *verdict = bad ? 2 : *overflow;
  return 0;
}
static int hs_peer_check(const gmx_peer& P, int64_t n) {
  if (!P.land_d || !P.tag_base_d || !P.status_d) return fail("peer: null pointer");
  if (P.world < 1 || P.world > 64 || P.rank < 0 || P.rank >= P.world || P.step < 0) return fail("peer: rank / world / step out of range");
  if (n <= 0 || P.tiles != (int32_t)((n + HS_TILE - 1) / HS_TILE)) return fail("peer: tiles must be ceil(n / 1024)");
  if (P.capacity < 1 || P.capacity > n || P.leaves < 1 || P.leaves > GMX_PEER_MAX_LEAVES) return fail("peer: capacity / leaves out of range");
  return 0;
}
extern "C" int gmx_peer_put_stats(const void* stats_own, gmx_peer P, int64_t n, gmx_stream) {
  if (!stats_own) return fail("peer_put_stats: null argument");
  if (hs_peer_check(P, n)) return 1;
  const int pad = P.tiles + (P.tiles & 1);
  const uint64_t* ag = (const uint64_t*)stats_own;
  const float* tm = (const float*)((const uint8_t*)stats_own + (size_t)pad * 8);
  const uint32_t tag = *P.tag_base_d + (uint32_t)P.step;
  for (int d = 0; d < P.world; ++d)
    if (d != P.rank)
      for (int b = 0; b < P.tiles; ++b) hs_peer_put_tile((uint64_t*)P.land_d[d], tag, P.world, P.tiles, P.rank, b, ag[b], tm[b]);
  return 0;
}
// wait (bounded) until granule *g carries `tag`; its data
static bool hs_peer_wait(const uint64_t* g, uint32_t tag, uint32_t* data, const timespec& t0) {
  for (;;) {
    const uint64_t v = __atomic_load_n(g, __ATOMIC_RELAXED);
    if ((uint32_t)(v >> 32) == tag) { *data = (uint32_t)v; return true; }
    sched_yield();
    timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    if (t1.tv_sec - t0.tv_sec > 60) return false;
  }
}
extern "C" int gmx_shard_step_peer(int kind, const uint32_t key[2], const void* stats_own, gmx_peer P, int64_t* plan,
                                   uint64_t* total_out, const float* lw, float* max_out, int shift, int64_t n,
                                   const void* const* state_rows, void* const* tail_rows, int32_t* next_idx, gmx_stream st) {
  if (!stats_own || !plan || !lw || !max_out || !state_rows || !tail_rows || !next_idx) return fail("shard_step_peer: null argument");
  if (hs_peer_check(P, n)) return 1;
  const int W = P.world, me = P.rank, tiles = P.tiles, pad = tiles + (tiles & 1);
  const int64_t cap = P.capacity;
  const uint32_t tag = *P.tag_base_d + (uint32_t)P.step;
  const size_t stride = gmx_shard_stats_bytes(n);
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  // the gathered table: own block from the local table, the others' rows from the landing block as they arrive
  std::vector<uint8_t> all((size_t)W * stride, 0);
  memcpy(all.data() + (size_t)me * stride, stats_own, stride);
  const uint64_t* land_own = (const uint64_t*)P.land_d[me];
  for (int r = 0; r < W; ++r) {
    if (r == me) continue;
    uint64_t* ag = (uint64_t*)(all.data() + (size_t)r * stride);
    float* tm = (float*)(all.data() + (size_t)r * stride + (size_t)pad * 8);
    for (int b = 0; b < tiles; ++b) {
      const uint64_t* row = land_own + gmx_peer_stats_at(tag, W, tiles, r, b);
      uint32_t lo, hi, mb;
      if (!hs_peer_wait(row, tag, &lo, t0) || !hs_peer_wait(row + 1, tag, &hi, t0) || !hs_peer_wait(row + 2, tag, &mb, t0)) {
        P.status_d[0] = 1; plan[GMX_PLAN_OVERFLOW] = 1;
        return fail("shard_step_peer: a peer's statistics did not arrive within 60 s");
      }
      ag[b] = (uint64_t)lo | ((uint64_t)hi << 32);
      tm[b] = gmx_u2f(mb);
    }
  }
  std::vector<uint64_t> totals((size_t)W);
  if (gmx_shard_totals(all.data(), W, n, totals.data(), max_out, st)) return 1;
  std::vector<uint32_t> send((size_t)W * (size_t)cap);
  for (int l = 0; l < P.leaves; ++l) {
    if (!state_rows[l] || !tail_rows[l]) return fail("shard_step_peer: a leaf pointer is null");
    if (gmx_shard_step_tiles(kind, key, totals.data(), plan, total_out, lw, stats_own, max_out, shift, me, W, n, cap,
                             state_rows[l], send.data(), next_idx, st)) return 1;
    // what this rank ships to d: the slots of d's shard whose ancestor lives here, [max(bounds[me], d n), min(bounds[me+1], (d+1) n))
    const int64_t lo_me = plan[GMX_PLAN_BOUNDS + me], hi_me = plan[GMX_PLAN_BOUNDS + me + 1];
    for (int d = 0; d < W; ++d) {
      if (d == me) continue;
      const int64_t a = lo_me > d * n ? lo_me : d * n, b = hi_me < (d + 1) * n ? hi_me : (d + 1) * n;
      const int64_t cnt = b > a ? (b - a < cap ? b - a : cap) : 0;
      uint64_t* land_d = (uint64_t*)P.land_d[d];
      for (int64_t k = 0; k < cnt; ++k)
        __atomic_store_n(land_d + gmx_peer_state_at(tag, W, tiles, cap, P.leaves, l, me, k),
                         gmx_granule(send[(size_t)d * (size_t)cap + (size_t)k], tag), __ATOMIC_RELAXED);
    }
  }
  // what arrives here from s: the slots of MY shard whose ancestor lives on s
  for (int l = 0; l < P.leaves; ++l) {
    uint32_t* tail = (uint32_t*)tail_rows[l];
    for (int s = 0; s < W; ++s) {
      if (s == me) continue;
      const int64_t lo_s = plan[GMX_PLAN_BOUNDS + s], hi_s = plan[GMX_PLAN_BOUNDS + s + 1];
      const int64_t a = lo_s > me * n ? lo_s : me * n, b = hi_s < (me + 1) * n ? hi_s : (me + 1) * n;
      const int64_t cnt = b > a ? (b - a < cap ? b - a : cap) : 0;
      for (int64_t k = 0; k < cnt; ++k) {
        uint32_t v;
        if (!hs_peer_wait(land_own + gmx_peer_state_at(tag, W, tiles, cap, P.leaves, l, s, k), tag, &v, t0)) {
          P.status_d[0] = 1;
          return fail("shard_step_peer: a peer's states did not arrive within 60 s");
        }
        tail[(size_t)s * (size_t)cap + (size_t)k] = v;
      }
    }
  }
  return 0;
}
// ---- peer-mapped exchange: the same protocol over PROCESS-SHARED memory (POSIX shm), so that the gloo ranks of the CPU
// tests really write into each other's buffers and wait on each other's flags ----
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
struct hs_shm { char name[48]; uint64_t bytes; };
static_assert(sizeof(hs_shm) <= GMX_P2P_HANDLE_BYTES, "handle");
#include <map>
static std::map<void*, hs_shm> g_shm;
extern "C" int gmx_p2p_alloc(size_t bytes, void** ptr_out, void* handle_out) {
  if (!ptr_out || !handle_out || !bytes) return fail("p2p_alloc: bad argument");
  static int counter = 0;
  hs_shm h; memset(&h, 0, sizeof(h));
  snprintf(h.name, sizeof(h.name), "/gmx_p2p_%d_%d_%ld", (int)getpid(), counter++, (long)time(nullptr));
  h.bytes = bytes;
  int fd = shm_open(h.name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return fail("p2p_alloc: shm_open / ftruncate failed");
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("p2p_alloc: mmap failed");
  memset(p, 0, bytes);
  memset(handle_out, 0, GMX_P2P_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  g_shm[p] = h;
  *ptr_out = p;
  return 0;
}
extern "C" int gmx_p2p_open(const void* handle, void** ptr_out) {
  if (!handle || !ptr_out) return fail("p2p_open: null argument");
  hs_shm h; memcpy(&h, handle, sizeof(h));
  int fd = shm_open(h.name, O_RDWR, 0600);
  if (fd < 0) return fail("p2p_open: shm_open failed");
  void* p = mmap(nullptr, h.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("p2p_open: mmap failed");
  hs_shm m = h; m.name[0] = 0;                   // a mapping of somebody else's segment: not ours to unlink
  g_shm[p] = m;
  *ptr_out = p;
  return 0;
}
extern "C" int gmx_p2p_close(void* p) {
  auto it = g_shm.find(p);
  if (it != g_shm.end()) { munmap(p, it->second.bytes); g_shm.erase(it); }
  return 0;
}
extern "C" int gmx_p2p_free(void* p) {
  auto it = g_shm.find(p);
  if (it != g_shm.end()) { if (it->second.name[0]) shm_unlink(it->second.name); munmap(p, it->second.bytes); g_shm.erase(it); }
  return 0;
}
extern "C" int gmx_p2p_exchange(const void* src, size_t src_stride, void* const* land_peers, const void* land_local, void* out,
                                uint64_t* const* flag_peers, uint64_t* flags_local, uint64_t* state, int rank, int world,
                                size_t bytes, gmx_stream) {
  if (!src || !land_peers || !land_local || !out || !flag_peers || !flags_local || !state) return fail("p2p_exchange: null argument");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail("p2p_exchange: rank / world out of range");
  if (!bytes) return 0;
  const uint64_t epoch = state[0] + 1;
  const size_t half = (size_t)(epoch & 1) * (size_t)world * bytes;
  for (int d = 0; d < world; ++d) {                                   // put into the peer's landing buffer, then announce
    memcpy((uint8_t*)land_peers[d] + half + (size_t)rank * bytes, (const uint8_t*)src + (size_t)d * src_stride, bytes);
    __atomic_store_n(flag_peers[d] + rank, epoch, __ATOMIC_RELEASE);
  }
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int d = 0; d < world; ++d) {                                   // wait for every peer's block (bounded), copy it out
    while (__atomic_load_n(flags_local + d, __ATOMIC_ACQUIRE) < epoch) {
      sched_yield();
      timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
      if (t1.tv_sec - t0.tv_sec > 60) { state[1] = 1; return fail("p2p_exchange: a peer did not arrive within 60 s"); }
    }
    memcpy((uint8_t*)out + (size_t)d * bytes, (const uint8_t*)land_local + half + (size_t)d * bytes, bytes);
  }
  state[0] = epoch;
  return 0;
}
extern "C" int gmx_gather(const void* const* src, void* const* dst, const int32_t* bytes, int32_t n_leaves,
                          const int32_t* anc, int64_t n_out, gmx_stream) {
  for (int32_t l = 0; l < n_leaves; ++l)
    for (int64_t j = 0; j < n_out; ++j)
      memcpy((char*)dst[l] + j * bytes[l], (const char*)src[l] + (int64_t)anc[j] * bytes[l], bytes[l]);
  return 0;
}
extern "C" int gmx_select(const uint8_t* mask, const void* const* a, const void* const* b, void* const* out,
                          const int32_t* bytes, int32_t n_leaves, int64_t n, gmx_stream) {
  for (int32_t l = 0; l < n_leaves; ++l)
    for (int64_t j = 0; j < n; ++j)
      memcpy((char*)out[l] + j * bytes[l], (const char*)(mask[j] ? a[l] : b[l]) + j * bytes[l], bytes[l]);
  return 0;
}
extern "C" int gmx_categorical_rows(const uint32_t* keys, const float* logits, int64_t rows, int64_t cols, int32_t* out, gmx_stream) {
  for (int64_t r = 0; r < rows; ++r) {
    gmx_key k; k.k0 = keys[2*r]; k.k1 = keys[2*r+1];
    gmx_cat_state s; s.best = 0; s.idx = 0;
    for (int64_t j = 0; j < cols; ++j) gmx_cat_step(&s, k, (uint64_t)j, (int)j, logits[r*cols+j]);
    out[r] = s.idx;
  }
  return 0;
}
extern "C" int gmx_mh_accept(const uint32_t* keys, const float* la, int64_t n, uint8_t* acc, gmx_stream) {
  for (int64_t i = 0; i < n; ++i) { gmx_key k; k.k0 = keys[2*i]; k.k1 = keys[2*i+1]; float u = gmx_uniform_sample(k, 0, 0.0f, 1.0f); acc[i] = gmx_logf(u) < la[i] ? 1 : 0; }
  return 0;
}
// graphs / timers: pass-through stubs (no streams on the host)
struct gmx_graph { int dummy; };
extern "C" int gmx_capture_begin(gmx_stream) { return fail("hostsim: graph capture unavailable"); }
extern "C" int gmx_capture_end(gmx_stream, gmx_graph**) { return fail("hostsim: graph capture unavailable"); }
extern "C" int gmx_graph_launch(gmx_graph*, gmx_stream) { return fail("hostsim: graph capture unavailable"); }
extern "C" int gmx_graph_destroy(gmx_graph*) { return 0; }
struct gmx_timer { int dummy; };
extern "C" int gmx_timer_create(gmx_timer** out) { *out = new gmx_timer; return 0; }
extern "C" int gmx_timer_start(gmx_timer*, gmx_stream) { return 0; }
extern "C" int gmx_timer_stop(gmx_timer*, gmx_stream) { return 0; }
extern "C" int gmx_timer_elapsed_ms(gmx_timer*, float* ms) { *ms = 0.0f; return 0; }
extern "C" int gmx_timer_destroy(gmx_timer* t) { delete t; return 0; }
