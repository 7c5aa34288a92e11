// gmx_vm.h — the per-particle site-program interpreter.
//
// One GPU thread runs gmx_vm_run for one particle.  The instruction stream is
// launch-uniform (every lane of every wave executes the same op with the same
// register indices), so decoding is scalar work (SALU + s_load) and register
// access is s_set_gpr_idx + v_mov: no divergence, no scratch.  The only HBM
// traffic is what OP_LDIN / OP_STOUT / OP_LDTAB name.
//
// The same template is instantiated (a) in gmx_kernels.hip with DevCtx for
// gfx950 and (b) by tests/hostsim (g++, HostCtx) so the product's own device
// functions can be diffed against the oracle without a GPU.  (b) is a test
// harness, never a fallback: the product library only contains (a).
#pragma once
#include "../../include/genmi.h"
#include "gmx_dist.h"
#include "gmx_program.h"

// Register file held in VGPRs; N <= 32 keeps hipcc's dynamic indexing on
// s_set_gpr_idx (larger vectors fall back to scratch on gfx950).
template <int N>
struct gmx_regs_vgpr {
#if defined(__HIPCC__)
  // an explicit vector type: dynamic element access lowers straight to
  // s_set_gpr_idx/v_mov (an array would depend on promote-alloca heuristics)
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

GMX_HD float gmx_asf(uint32_t u) { return gmx_u2f(u); }
GMX_HD uint32_t gmx_asu(float f) { return gmx_f2u(f); }

// Ctx must provide:
//   uint32_t uniform(uint32_t)              broadcast of a wave-uniform value
//   void red_max(float x, bool active)      block partial -> red_out[block][0]
//   void red_lse(float x, bool active)      block partial -> red_out[block][0..1]
template <class Regs, class Ctx>
GMX_HD void gmx_vm_run(const uint32_t* __restrict__ code, uint32_t n_instr, int64_t i,
                       bool active, const gmx_run_args& A, Ctx& ctx) {
  Regs R;
  R.init();
  for (uint32_t pc = 0; pc < n_instr; ++pc) {
    const uint32_t w0 = ctx.uniform(code[2u * pc]);
    const uint32_t w1 = ctx.uniform(code[2u * pc + 1u]);
    const uint32_t op = w0 & 0xffu, dst = (w0 >> 8) & 0xffu, a = (w0 >> 16) & 0xffu,
                   b = w0 >> 24;
    const uint32_t c = w1 & 0xffu, e = w1 >> 8;
    switch (op) {
      case OP_CONST: R.set(dst, w1); break;
      case OP_UNI: R.set(dst, A.uni[w1 & (GMX_MAX_UNI - 1)]); break;
      case OP_LDIN: {
        uint32_t v = 0u;
        if (active) {
          int64_t row = i;
          if (b & GMX_F_GATHER) row = (int64_t)A.ancestors_d[i];
          if (b & GMX_F_BCAST) row = 0;
          if (b & GMX_F_U8) v = (uint32_t)((const uint8_t*)A.in_d[a])[row];
          else v = ((const uint32_t*)A.in_d[a])[row];
        }
        R.set(dst, v);
      } break;
      case OP_LDTAB: {
        int32_t idx = (int32_t)R.get(b) + (int32_t)w1;
        uint32_t v = 0u;
        if (active) v = ((const uint32_t*)A.tab_d[a])[idx];
        R.set(dst, v);
      } break;
      case OP_STOUT: {
        uint32_t v = R.get(b);
        if (active) {
          if (dst & GMX_F_U8) ((uint8_t*)A.out_d[a])[i] = (uint8_t)(v != 0u);
          else ((uint32_t*)A.out_d[a])[i] = v;
        }
      } break;
      case OP_LDKEY: {
        gmx_key k; k.k0 = A.key0; k.k1 = A.key1;
        if (A.key_mode == GMX_KEY_ARRAY) {
          if (active) { k.k0 = A.keys_d[2 * i]; k.k1 = A.keys_d[2 * i + 1]; }
        } else if (A.key_mode == GMX_KEY_SPLIT) {
          k = gmx_split_child(k, (uint64_t)(A.index_offset + i));
        } else if (A.key_mode == GMX_KEY_ROWSPLIT) {
          int64_t row = i / A.key_inner, j = i - row * A.key_inner;
          if (active) { k.k0 = A.keys_d[2 * row]; k.k1 = A.keys_d[2 * row + 1]; }
          k = gmx_split_child(k, (uint64_t)j);
        }
        R.set(dst, k.k0); R.set(dst + 1u, k.k1);
      } break;
      case OP_KDERIVE: case OP_KDERIVER: {
        gmx_key k; k.k0 = R.get(a); k.k1 = R.get(a + 1u);
        uint32_t d = (op == OP_KDERIVE) ? w1 : R.get(b);
        gmx_key o = gmx_fold_in(k, d);
        R.set(dst, o.k0); R.set(dst + 1u, o.k1);
      } break;
      case OP_MOV: R.set(dst, R.get(a)); break;
      // ---- f32 binary ----
      case OP_ADD: R.set(dst, gmx_asu(gmx_asf(R.get(a)) + gmx_asf(R.get(b)))); break;
      case OP_SUB: R.set(dst, gmx_asu(gmx_asf(R.get(a)) - gmx_asf(R.get(b)))); break;
      case OP_MUL: R.set(dst, gmx_asu(gmx_asf(R.get(a)) * gmx_asf(R.get(b)))); break;
      case OP_DIV: R.set(dst, gmx_asu(gmx_asf(R.get(a)) / gmx_asf(R.get(b)))); break;
      case OP_MIN: R.set(dst, gmx_asu(gmx_fmin(gmx_asf(R.get(a)), gmx_asf(R.get(b))))); break;
      case OP_MAX: R.set(dst, gmx_asu(gmx_fmax(gmx_asf(R.get(a)), gmx_asf(R.get(b))))); break;
      case OP_POW: R.set(dst, gmx_asu(gmx_powf(gmx_asf(R.get(a)), gmx_asf(R.get(b))))); break;
      // ---- f32 unary ----
      case OP_NEG: R.set(dst, R.get(a) ^ 0x80000000u); break;
      case OP_ABS: R.set(dst, R.get(a) & 0x7fffffffu); break;
      case OP_EXP: R.set(dst, gmx_asu(gmx_expf(gmx_asf(R.get(a))))); break;
      case OP_LOG: R.set(dst, gmx_asu(gmx_logf(gmx_asf(R.get(a))))); break;
      case OP_LOG1P: R.set(dst, gmx_asu(gmx_log1pf(gmx_asf(R.get(a))))); break;
      case OP_SQRT: R.set(dst, gmx_asu(gmx_sqrtf(gmx_asf(R.get(a))))); break;
      case OP_SIN: R.set(dst, gmx_asu(gmx_sinf(gmx_asf(R.get(a))))); break;
      case OP_COS: R.set(dst, gmx_asu(gmx_cosf(gmx_asf(R.get(a))))); break;
      case OP_TANH: R.set(dst, gmx_asu(gmx_tanhf(gmx_asf(R.get(a))))); break;
      case OP_SIGMOID: R.set(dst, gmx_asu(gmx_sigmoidf(gmx_asf(R.get(a))))); break;
      case OP_SOFTPLUS: R.set(dst, gmx_asu(gmx_softplusf(gmx_asf(R.get(a))))); break;
      case OP_FLOOR: R.set(dst, gmx_asu(__builtin_floorf(gmx_asf(R.get(a))))); break;
      case OP_CEIL: R.set(dst, gmx_asu(__builtin_ceilf(gmx_asf(R.get(a))))); break;
      case OP_ROUND: R.set(dst, gmx_asu(__builtin_rintf(gmx_asf(R.get(a))))); break;
      case OP_LGAMMA: R.set(dst, gmx_asu(gmx_lgammaf(gmx_asf(R.get(a))))); break;
      case OP_SQUARE: { float x = gmx_asf(R.get(a)); R.set(dst, gmx_asu(x * x)); } break;
      case OP_RECIP: R.set(dst, gmx_asu(1.0f / gmx_asf(R.get(a)))); break;
      // ---- comparisons -> i32 0/1 ----
      case OP_FLT: R.set(dst, gmx_asf(R.get(a)) < gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_FLE: R.set(dst, gmx_asf(R.get(a)) <= gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_FGT: R.set(dst, gmx_asf(R.get(a)) > gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_FGE: R.set(dst, gmx_asf(R.get(a)) >= gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_FEQ: R.set(dst, gmx_asf(R.get(a)) == gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_FNE: R.set(dst, gmx_asf(R.get(a)) != gmx_asf(R.get(b)) ? 1u : 0u); break;
      case OP_IEQ: R.set(dst, R.get(a) == R.get(b) ? 1u : 0u); break;
      case OP_INE: R.set(dst, R.get(a) != R.get(b) ? 1u : 0u); break;
      case OP_ILT: R.set(dst, (int32_t)R.get(a) < (int32_t)R.get(b) ? 1u : 0u); break;
      case OP_ILE: R.set(dst, (int32_t)R.get(a) <= (int32_t)R.get(b) ? 1u : 0u); break;
      case OP_IGT: R.set(dst, (int32_t)R.get(a) > (int32_t)R.get(b) ? 1u : 0u); break;
      case OP_IGE: R.set(dst, (int32_t)R.get(a) >= (int32_t)R.get(b) ? 1u : 0u); break;
      case OP_AND: R.set(dst, (R.get(a) != 0u && R.get(b) != 0u) ? 1u : 0u); break;
      case OP_OR: R.set(dst, (R.get(a) != 0u || R.get(b) != 0u) ? 1u : 0u); break;
      case OP_XOR: R.set(dst, ((R.get(a) != 0u) != (R.get(b) != 0u)) ? 1u : 0u); break;
      case OP_NOT: R.set(dst, R.get(a) == 0u ? 1u : 0u); break;
      case OP_SEL: R.set(dst, R.get(c) != 0u ? R.get(a) : R.get(b)); break;
      case OP_I2F: R.set(dst, gmx_asu((float)(int32_t)R.get(a))); break;
      case OP_F2I: R.set(dst, (uint32_t)(int32_t)gmx_asf(R.get(a))); break;
      case OP_IADD: R.set(dst, R.get(a) + R.get(b)); break;
      case OP_ISUB: R.set(dst, R.get(a) - R.get(b)); break;
      case OP_IMUL: R.set(dst, (uint32_t)((int32_t)R.get(a) * (int32_t)R.get(b))); break;
      case OP_INEG: R.set(dst, (uint32_t)(-(int32_t)R.get(a))); break;
      // ---- samplers ----
      case OP_S_NORMAL: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        R.set(dst, gmx_asu(gmx_normal_sample(k, e, gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
      } break;
      case OP_S_UNIFORM: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        R.set(dst, gmx_asu(gmx_uniform_sample(k, e, gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
      } break;
      case OP_S_FLIP: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        R.set(dst, (uint32_t)gmx_flip_sample(k, e, gmx_asf(R.get(a))));
      } break;
      case OP_S_BERNL: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        R.set(dst, (uint32_t)gmx_bernoulli_logits_sample(k, e, gmx_asf(R.get(a))));
      } break;
      case OP_S_BETA: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        R.set(dst, gmx_asu(gmx_beta_sample(k, e, gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
      } break;
      case OP_S_CATSTEP: {
        gmx_key k; k.k0 = R.get(c); k.k1 = R.get(c + 1u);
        gmx_cat_state s; s.best = gmx_asf(R.get(dst)); s.idx = (int)R.get(dst + 1u);
        gmx_cat_step(&s, k, (uint64_t)R.get(b), (int)e, gmx_asf(R.get(a)));
        R.set(dst, gmx_asu(s.best)); R.set(dst + 1u, (uint32_t)s.idx);
      } break;
      // ---- log densities ----
      case OP_L_NORMAL:
        R.set(dst, gmx_asu(gmx_normal_logpdf(gmx_asf(R.get(c)), gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
        break;
      case OP_L_UNIFORM:
        R.set(dst, gmx_asu(gmx_uniform_logpdf(gmx_asf(R.get(c)), gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
        break;
      case OP_L_FLIP:
        R.set(dst, gmx_asu(gmx_flip_logpdf((int)R.get(c), gmx_asf(R.get(a)))));
        break;
      case OP_L_BERNL:
        R.set(dst, gmx_asu(gmx_bernoulli_logits_logpdf((int)R.get(c), gmx_asf(R.get(a)))));
        break;
      case OP_L_BETA:
        R.set(dst, gmx_asu(gmx_beta_logpdf(gmx_asf(R.get(c)), gmx_asf(R.get(a)), gmx_asf(R.get(b)))));
        break;
      // ---- block reductions ----
      case OP_REDMAX: ctx.red_max(gmx_asf(R.get(a)), active); break;
      case OP_REDLSE: ctx.red_lse(gmx_asf(R.get(a)), active); break;
      default: break;  // OP_END and unknown ops (rejected at program_create)
    }
  }
}
