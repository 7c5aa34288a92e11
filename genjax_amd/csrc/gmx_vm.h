// gmx_vm.h — the per-particle site-program interpreter.
//
// One GPU thread runs gmx_vm_run for one particle.  The instruction stream is
// launch-uniform (every lane of every wave executes the same op with the same
// operand codes), so decoding is scalar work and there is no divergence.
//
// What keeps the interpreter cheap on gfx950:
//   * the program, the constant/uniform pool and the slot pointer tables are
//     loaded ONCE per wave into VGPR lanes (lane l holds entry l) and fetched
//     with v_readlane — no memory latency inside the op loop;
//   * the register file is one ext_vector in VGPRs indexed with s_set_gpr_idx;
//     every op produces its result in scalars and there is a single write
//     after the switch, so the vector is updated in place (no copies);
//   * operands may name a pool entry directly (codes >= 64), so constants and
//     launch uniforms cost no instruction;
//   * a LITE build of the switch (no Beta/categorical/lgamma/trig) keeps VGPR
//     use low for the common Normal/Bernoulli/Uniform programs.
// The only HBM traffic is what OP_LDIN / OP_STOUT / OP_LDTAB name.
//
// The same template is instantiated (a) in gmx_kernels.hip with DevCtx for
// gfx950 and (b) by tests/hostsim (g++, HostCtx) so the product's own device
// functions can be diffed against the oracle without a GPU.  (b) is a test
// harness, never a fallback: the product library only contains (a).
#pragma once
#include "genmi.h"
#include "gmx_dist.h"
#include "gmx_program.h"

// Register file held in VGPRs; N <= 32 keeps hipcc's dynamic indexing on
// s_set_gpr_idx (larger vectors fall back to scratch on gfx950).
template <int N>
struct gmx_regs_vgpr {
#if defined(__HIPCC__)
  typedef uint32_t vec_t __attribute__((ext_vector_type(N)));
  vec_t v;
  GMX_HDM void init() { v = (vec_t)(0u); }
#else
  uint32_t v[N];
  GMX_HDM void init() { for (int k = 0; k < N; ++k) v[k] = 0u; }
#endif
  GMX_HDM uint32_t get(uint32_t i) const { return v[i]; }
  GMX_HDM void set(uint32_t i, uint32_t x) { v[i] = x; }
};

// The row of keys a particle of a GMX_KEY_ROWSPLIT launch draws from.  A 2-D launch (gmx_program_run: background
// programs, one row of keys per blockIdx.y) knows it; everything else (1-D launches, the tests' CPU mirror) divides.
GMX_HD int64_t gmx_rowsplit_row(int64_t i, int64_t inner) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (gridDim.y > 1) return (int64_t)blockIdx.y;
#endif
  return i / inner;
}

GMX_HD float gmx_asf(uint32_t u) { return gmx_u2f(u); }
GMX_HD uint32_t gmx_asu(float f) { return gmx_f2u(f); }

// Ctx must provide (all arguments launch-uniform):
//   void     fetch(pc, &w0, &w1)     instruction words
//   uint32_t pool(i)                 constant / uniform pool entry
//   const void* in_ptr(slot); void* out_ptr(slot); const void* tab_ptr(slot)
//   void red_max(float x, bool active); void red_lse(float x, bool active)
// An instruction word pair: run-time (interpreter) or compile-time (specialised
// programs: op / operand codes are constant expressions, so the compiler emits
// only the one switch case an instruction needs).
struct gmx_rword {
  uint32_t a, b;
  GMX_HDM uint32_t w0() const { return a; }
  GMX_HDM uint32_t w1() const { return b; }
};
template <uint32_t A0, uint32_t B0>
struct gmx_cword {
  static constexpr uint32_t w0() { return A0; }
  static constexpr uint32_t w1() { return B0; }
};

// One instruction.  All of w0 / w1 are launch-uniform.
template <class Regs, bool FULL, class W, class Ctx>
GMX_HD void gmx_vm_step(Regs& R, const W w, int64_t i, bool active, const gmx_run_args& A, Ctx& ctx, uint32_t t = 0u,
                        uint32_t tf = 0u) {
  const uint32_t w0 = w.w0(), w1 = w.w1();
#define SRC(x) ((x) < GMX_POOL_BASE ? R.get(x) : ctx.pool((x) - GMX_POOL_BASE))
#define FSRC(x) gmx_asf(SRC(x))
#define KEY(x) gmx_key k; k.k0 = R.get(x); k.k1 = R.get((x) + 1u)
#define ELEM() (e == GMX_ELEM_INDEX ? (uint32_t)(A.index_offset + i) : (e == GMX_ELEM_LOOP ? t : e))
  {
    const uint32_t op = w0 & 0xffu, dst = (w0 >> 8) & 0xffu, a = (w0 >> 16) & 0xffu, b = w0 >> 24;
    const uint32_t c = w1 & 0xffu, e = w1 >> 8;
    uint32_t r0 = 0u, r1 = 0u;
    int wr = 1;            // registers written after the switch: 0, 1 (dst) or 2 (dst, dst+1)
    switch (op) {
      case OP_CONST: r0 = w1; break;
      case OP_UNI: r0 = ctx.pool(w1 & (GMX_MAX_UNI - 1)); break;
      case OP_LDIN: {
        if (b & GMX_F_BCAST) {              // one launch-uniform element: every lane alike -> a scalar load
          const void* p = ctx.in_ptr(a);
          r0 = (b & GMX_F_U8) ? (uint32_t)((const uint8_t*)p)[0] : ((const uint32_t*)p)[0];
        } else if (active) {
          int64_t row = i;
          if (b & GMX_F_GATHER) row = (int64_t)A.ancestors_d[i];
          if (b & GMX_F_BCAST) row = 0;
          if (b & GMX_F_IDX) row += (int64_t)((int32_t)SRC(c) + (int32_t)e) * A.step_stride;        // element r[c] + e: chosen at run time
          else if (b & GMX_F_STEP) row += (int64_t)(((b & GMX_F_FLAT) ? tf : t) + w1) * A.step_stride;   // element t + imm of a [T, n] leaf
          const void* p = ctx.in_ptr(a);
          if (b & GMX_F_U8) r0 = (uint32_t)((const uint8_t*)p)[row];
          else r0 = ((const uint32_t*)p)[row];
        }
      } break;
      case OP_LDTAB: {
        int32_t idx = (int32_t)SRC(b) + (int32_t)w1;
        // A pool operand is launch-uniform: the load is then done by every lane alike, outside any divergent
        // branch, so a specialised kernel gets a scalar load (s_load_dword through the scalar cache) instead of
        // 64 lanes fetching one address — a K = 64 mixture reads 128 table entries per datapoint.
        if (b >= GMX_POOL_BASE) r0 = ((const uint32_t*)ctx.tab_ptr(a))[idx];
        else if (active) r0 = ((const uint32_t*)ctx.tab_ptr(a))[idx];
      } break;
      case OP_STOUT: {
        uint32_t v = SRC(b);
#if defined(GMX_JIT_FAULT)
        if (!(dst & GMX_F_U8)) v ^= 1u;          // (GENMI_JIT_FAULT=1: a deliberately wrong specialised kernel, tests only)
#endif
        if (active) {
          void* p = ctx.out_ptr(a);
          const int64_t orow = (dst & GMX_F_STEP) ? i + (int64_t)(((dst & GMX_F_FLAT) ? tf : t) + w1) * A.step_stride : i;
          if (dst & GMX_F_U8) ((uint8_t*)p)[orow] = (uint8_t)(v != 0u);
          else ((uint32_t*)p)[orow] = v;
        }
        wr = 0;
      } break;
      case OP_LDKEY: {
        gmx_key k; k.k0 = A.key0; k.k1 = A.key1;
        if (A.key_mode == GMX_KEY_ARRAY) {
          if (active) { k.k0 = A.keys_d[2 * i]; k.k1 = A.keys_d[2 * i + 1]; }
        } else if (A.key_mode == GMX_KEY_SPLIT) {
          k = gmx_split_child(k, (uint64_t)(A.index_offset + i));
        } else if (A.key_mode == GMX_KEY_ROWSPLIT) {
          int64_t row, j;
#if defined(__HIP_DEVICE_COMPILE__)
          if (gridDim.y > 1) {
            // a 2-D launch: the row is the workgroup's (blockIdx.y), so the row key is LAUNCH-UNIFORM per workgroup —
            // loaded unconditionally through the scalar unit, and the key schedule of the split (ks2 = k0 ^ k1 ^ C,
            // the injections) stays on the scalar unit as in GMX_KEY_SPLIT mode.  Per-lane key loads made this mode
            // cost 75 vector instructions per wave more than GMX_KEY_SPLIT (SQ_INSTS_VALU 1211 vs 1136, noise program).
            row = (int64_t)blockIdx.y; j = i - row * A.key_inner;
            k.k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.keys_d[2 * row]);
            k.k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.keys_d[2 * row + 1]);
          } else
#endif
          {
            row = i / A.key_inner; j = i - row * A.key_inner;
            if (active) { k.k0 = A.keys_d[2 * row]; k.k1 = A.keys_d[2 * row + 1]; }
          }
          k = gmx_split_child(k, (uint64_t)(A.index_offset + j));     // a shard's block of every row's split
        }
        r0 = k.k0; r1 = k.k1; wr = 2;
      } break;
      case OP_KDERIVE: case OP_KDERIVER: {
        KEY(a);
        uint32_t d = (op == OP_KDERIVE) ? w1 : SRC(b);
        gmx_key o = gmx_fold_in(k, d);
        r0 = o.k0; r1 = o.k1; wr = 2;
      } break;
      case OP_LDIDX: r0 = (uint32_t)(A.index_offset + i); break;
      case OP_LDT: r0 = t; break;
      case OP_KSPLITU: {
        gmx_key k; k.k0 = SRC(a); k.k1 = SRC(b);
        k = gmx_split_child(k, (uint64_t)(A.index_offset + i));
        r0 = k.k0; r1 = k.k1; wr = 2;
      } break;
      case OP_MOV: r0 = SRC(a); break;
      // ---- f32 binary ----
      case OP_ADD: r0 = gmx_asu(FSRC(a) + FSRC(b)); break;
      case OP_SUB: r0 = gmx_asu(FSRC(a) - FSRC(b)); break;
      case OP_MUL: r0 = gmx_asu(FSRC(a) * FSRC(b)); break;
      case OP_DIV: r0 = gmx_asu(FSRC(a) / FSRC(b)); break;
      case OP_MIN: r0 = gmx_asu(gmx_fmin(FSRC(a), FSRC(b))); break;
      case OP_MAX: r0 = gmx_asu(gmx_fmax(FSRC(a), FSRC(b))); break;
      // ---- f32 unary ----
      case OP_NEG: r0 = SRC(a) ^ 0x80000000u; break;
      case OP_ABS: r0 = SRC(a) & 0x7fffffffu; break;
      case OP_EXP: r0 = gmx_asu(gmx_expf(FSRC(a))); break;
      case OP_LOG: r0 = gmx_asu(gmx_logf(FSRC(a))); break;
      case OP_LOG1P: r0 = gmx_asu(gmx_log1pf(FSRC(a))); break;
      case OP_SQRT: r0 = gmx_asu(gmx_sqrtf(FSRC(a))); break;
      case OP_FLOOR: r0 = gmx_asu(__builtin_floorf(FSRC(a))); break;
      case OP_CEIL: r0 = gmx_asu(__builtin_ceilf(FSRC(a))); break;
      case OP_ROUND: r0 = gmx_asu(__builtin_rintf(FSRC(a))); break;
      case OP_SQUARE: { float x = FSRC(a); r0 = gmx_asu(x * x); } break;
      case OP_RECIP: r0 = gmx_asu(1.0f / FSRC(a)); break;
      case OP_SIGMOID: r0 = gmx_asu(gmx_sigmoidf(FSRC(a))); break;
      // ---- comparisons -> i32 0/1 ----
      case OP_FLT: r0 = FSRC(a) < FSRC(b) ? 1u : 0u; break;
      case OP_FLE: r0 = FSRC(a) <= FSRC(b) ? 1u : 0u; break;
      case OP_FGT: r0 = FSRC(a) > FSRC(b) ? 1u : 0u; break;
      case OP_FGE: r0 = FSRC(a) >= FSRC(b) ? 1u : 0u; break;
      case OP_FEQ: r0 = FSRC(a) == FSRC(b) ? 1u : 0u; break;
      case OP_FNE: r0 = FSRC(a) != FSRC(b) ? 1u : 0u; break;
      case OP_IEQ: r0 = SRC(a) == SRC(b) ? 1u : 0u; break;
      case OP_INE: r0 = SRC(a) != SRC(b) ? 1u : 0u; break;
      case OP_ILT: r0 = (int32_t)SRC(a) < (int32_t)SRC(b) ? 1u : 0u; break;
      case OP_ILE: r0 = (int32_t)SRC(a) <= (int32_t)SRC(b) ? 1u : 0u; break;
      case OP_IGT: r0 = (int32_t)SRC(a) > (int32_t)SRC(b) ? 1u : 0u; break;
      case OP_IGE: r0 = (int32_t)SRC(a) >= (int32_t)SRC(b) ? 1u : 0u; break;
      case OP_AND: r0 = (SRC(a) != 0u && SRC(b) != 0u) ? 1u : 0u; break;
      case OP_OR: r0 = (SRC(a) != 0u || SRC(b) != 0u) ? 1u : 0u; break;
      case OP_XOR: r0 = ((SRC(a) != 0u) != (SRC(b) != 0u)) ? 1u : 0u; break;
      case OP_NOT: r0 = SRC(a) == 0u ? 1u : 0u; break;
      case OP_SEL: r0 = SRC(c) != 0u ? SRC(a) : SRC(b); break;
      case OP_I2F: r0 = gmx_asu((float)(int32_t)SRC(a)); break;
      case OP_F2I: r0 = (uint32_t)(int32_t)FSRC(a); break;
      case OP_IADD: r0 = SRC(a) + SRC(b); break;
      case OP_ISUB: r0 = SRC(a) - SRC(b); break;
      case OP_IMUL: r0 = (uint32_t)((int32_t)SRC(a) * (int32_t)SRC(b)); break;
      case OP_INEG: r0 = (uint32_t)(-(int32_t)SRC(a)); break;
      // ---- samplers ----
      case OP_S_NORMAL: { KEY(c); r0 = gmx_asu(gmx_normal_sample(k, ELEM(), FSRC(a), FSRC(b))); } break;
      case OP_S_UNIFORM: { KEY(c); r0 = gmx_asu(gmx_uniform_sample(k, ELEM(), FSRC(a), FSRC(b))); } break;
      case OP_S_FLIP: { KEY(c); r0 = (uint32_t)gmx_flip_sample(k, ELEM(), FSRC(a)); } break;
      case OP_S_BERNL: { KEY(c); r0 = (uint32_t)gmx_bernoulli_logits_sample(k, ELEM(), FSRC(a)); } break;
      // ---- log densities ----
      case OP_L_NORMAL: r0 = gmx_asu(gmx_normal_logpdf(FSRC(c), FSRC(a), FSRC(b))); break;
      case OP_L_UNIFORM: r0 = gmx_asu(gmx_uniform_logpdf(FSRC(c), FSRC(a), FSRC(b))); break;
      case OP_L_FLIP: r0 = gmx_asu(gmx_flip_logpdf((int)SRC(c), FSRC(a))); break;
      case OP_L_BERNL: r0 = gmx_asu(gmx_bernoulli_logits_logpdf((int)SRC(c), FSRC(a))); break;
      // ---- block reductions ----
      case OP_REDMAX: ctx.red_max(FSRC(a), active); wr = 0; break;
      case OP_REDLSE: ctx.red_lse(FSRC(a), active); wr = 0; break;
      default:
        wr = 0;
        if (FULL) {
          wr = 1;
          switch (op) {   // ops only the FULL build carries (gmx_op_needs_full)
            case OP_POW: r0 = gmx_asu(gmx_powf(FSRC(a), FSRC(b))); break;
            case OP_SIN: r0 = gmx_asu(gmx_sinf(FSRC(a))); break;
            case OP_COS: r0 = gmx_asu(gmx_cosf(FSRC(a))); break;
            case OP_TANH: r0 = gmx_asu(gmx_tanhf(FSRC(a))); break;
            case OP_SOFTPLUS: r0 = gmx_asu(gmx_softplusf(FSRC(a))); break;
            case OP_LGAMMA: r0 = gmx_asu(gmx_lgammaf(FSRC(a))); break;
            case OP_S_BETA: { KEY(c); r0 = gmx_asu(gmx_beta_sample(k, ELEM(), FSRC(a), FSRC(b))); } break;
            case OP_S_LOGGAMMA: { KEY(c); r0 = gmx_asu(gmx_log_gamma_sample(gmx_split_child(k, ELEM()), FSRC(a))); } break;
            case OP_S_CATSTEP: {
              KEY(c);
              gmx_cat_state s; s.best = gmx_asf(R.get(dst)); s.idx = (int)R.get(dst + 1u);
              gmx_cat_step(&s, k, (uint64_t)SRC(b), (int)e, FSRC(a));
              r0 = gmx_asu(s.best); r1 = (uint32_t)s.idx; wr = 2;
            } break;
            case OP_L_BETA: r0 = gmx_asu(gmx_beta_logpdf(FSRC(c), FSRC(a), FSRC(b))); break;
            default: wr = 0; break;   // OP_END / unknown (rejected at program_create)
          }
        }
        break;
    }
    // OP_STOUT keeps its FLAGS in the dst field (GMX_F_STEP | GMX_F_FLAT = 24): not a register.  The device
    // interpreter's write-back is an indexed VGPR move that the compiler predicates by VALUE, not by skipping it — with
    // an index past the 16-element file it lands in a neighbouring register (measured on gfx950: a wild pointer), so
    // the index itself is kept in range whenever nothing is written.
    const uint32_t dreg = (op == OP_STOUT) ? 0u : dst;
    if (wr >= 1) R.set(dreg, r0);
    if (wr == 2) R.set(dreg + 1u, r1);
  }
#undef SRC
#undef FSRC
#undef KEY
#undef ELEM
}

// NI < 0: interpret n_instr_rt instructions fetched through ctx at run time.
template <class Regs, bool FULL, int NI, class Ctx>
GMX_HD void gmx_vm_run(uint32_t n_instr_rt, int64_t i, bool active, const gmx_run_args& A, Ctx& ctx) {
  Regs R;
  R.init();
  // counted loops, at most three deep: launch-uniform control flow.  t = the innermost loop's iteration number,
  // tf = the row-major index over ALL enclosing loops (GMX_F_FLAT): (t0 * n1 + t1) * n2 + t2 three deep.  Few live
  // scalars (the switch below is at the SGPR limit): the heads of the two outer loops packed in one word, the third in
  // a second; the trip counts are re-read from the LOOP instructions at the loop ends, and the enclosing loop's
  // iteration number is recovered from tf when a loop finishes (tf / n = the parent's flat index, whose remainder by
  // the parent's own count is its iteration number).  (No arrays indexed at run time: the device interpreter's
  // register file already owns the GPR-index mode.)
  uint32_t t = 0u, tf = 0u, heads = 0u, head3 = 0u;   // heads: pc of the outer LOOP | pc of the second << 16
  int depth = 0;
  for (uint32_t pc = 0; pc < n_instr_rt; ++pc) {
    gmx_rword w;
    ctx.fetch(pc, &w.a, &w.b);
    const uint32_t op = w.a & 0xffu;
    if (op == OP_LOOP) {
      if (depth == 0) { heads = pc; depth = 1; t = 0u; tf = 0u; }
      else if (depth == 1) { heads = (heads & 0xffffu) | (pc << 16); depth = 2; tf = tf * w.b; t = 0u; }
      else { head3 = pc; depth = 3; tf = tf * w.b; t = 0u; }
      continue;
    }
    if (op == OP_ENDLOOP) {
      gmx_rword h;
      const uint32_t hp = depth == 3 ? head3 : (depth == 2 ? (heads >> 16) : (heads & 0xffffu));
      ctx.fetch(hp, &h.a, &h.b);                         // h.b = this loop's trip count
      if (t + 1u < h.b) { ++t; ++tf; pc = hp; }
      else if (depth == 3) {
        gmx_rword h2;
        ctx.fetch(heads >> 16, &h2.a, &h2.b);            // the second loop's trip count
        depth = 2; tf = tf / h.b; t = tf % h2.b;
      }
      else if (depth == 2) { depth = 1; t = tf / h.b; tf = t; }
      else { depth = 0; t = 0u; tf = 0u; }
      continue;
    }
    gmx_vm_step<Regs, FULL, gmx_rword, Ctx>(R, w, i, active, A, ctx, t, tf);
  }
}

// ops that need the FULL interpreter build
GMX_HD bool gmx_op_needs_full(uint32_t op) {
  return op == OP_POW || op == OP_SIN || op == OP_COS || op == OP_TANH || op == OP_SOFTPLUS ||
         op == OP_LGAMMA || op == OP_S_BETA || op == OP_S_CATSTEP || op == OP_L_BETA || op == OP_S_LOGGAMMA;
}
