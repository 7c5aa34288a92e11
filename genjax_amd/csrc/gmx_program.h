// gmx_program.h — encoding of a site program (the unit gmx_program_run executes).
//
// A program is a flat array of uint32 words:
//   [0] GMX_PROG_MAGIC   [1] GMX_PROG_VERSION   [2] n_instr   [3] n_regs
//   [4] n_in  [5] n_out  [6] n_uni (pool entries used)  [7] n_tab
//   [8] n_const  [9] n_dyn (launch uniforms; n_uni = n_dyn + n_const)
//   then n_instr instructions of two words each:
//     w0 = op | dst << 8 | a << 16 | b << 24          w1 = imm32
//   then n_const constant words: pool entries n_dyn .. n_dyn + n_const - 1
//   (gmx_program_run writes them into gmx_run_args.uni itself).
//   For three-operand ops c = imm & 0xff and e = imm >> 8 (24 bits).
//   A SOURCE operand code x names register x when x < GMX_POOL_BASE and pool
//   entry (x - GMX_POOL_BASE) otherwise.  The pool is gmx_run_args.uni: launch
//   uniforms first, then the program's constants (the host fills both), so a
//   constant or a uniform costs no instruction.  dst and key operands are
//   always registers.
//
// Registers are untyped 32-bit cells r[0..n_regs); an op reads them as f32 or
// i32.  A PRNG key occupies two consecutive registers (k, k+1).  Booleans are
// i32 0/1.  There is no data-dependent control flow: a `@gen` static function has a fixed
// site list (static.py "Language restrictions"), `jax.lax.cond`/`where` on
// values become OP_SEL.  The ONE loop form is a counted, launch-uniform repetition of a block
// (OP_LOOP count ... OP_ENDLOOP; at most three deep): what `jax.lax.scan` is to the reference's Scan combinator
// (combinators/scan.py:221, 278).  Values carried across iterations live in registers the block reads and
// overwrites (the encoder emits the copies); OP_LDT yields the iteration number; loads / stores flagged
// GMX_F_STEP address element t of a [T, n] leaf.  All indices are launch-uniform, so on gfx950 the
// register file is indexed with s_set_gpr_idx (no scratch, no LDS).
//
// The Python encoder is genjax_amd/program.py; the independent CPU checker is
// oracle/ (which does NOT execute programs: it restates the reference's
// handlers directly on numpy arrays).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#else
#include "genmi.h"
#endif

#define GMX_PROG_MAGIC 0x50584D47u /* 'GMXP' */
#define GMX_PROG_VERSION 2u
#define GMX_PROG_HEADER_WORDS 10u
#define GMX_MAX_REGS 64 /* > 31: specialised kernels only (the interpreter's register file is 32 VGPRs, last one unused) */
#define GMX_POOL_BASE 64u /* source operand codes >= this name pool entries */

// LDIN / STOUT flag bits (field b for LDIN, field dst for STOUT)
/* a sampler's element counter (the 24-bit immediate of OP_S_*): this value means "the particle's GLOBAL index"
 * (index_offset + i) — a vector-valued site of n elements run with its elements on the launch axis: element i draws with
 * counter i from the ONE site key, as the unrolled form's element i does with immediate i */
#define GMX_ELEM_INDEX 0xffffffu
/* ... and this one "the iteration number of the innermost enclosing OP_LOOP": a vector-valued site of many elements under
 * a particle batch runs as a counted loop per particle, iteration j drawing with counter j from the one site key */
#define GMX_ELEM_LOOP 0xfffffeu
#define GMX_F_GATHER 1u /* row = ancestors[i] instead of i            */
#define GMX_F_U8 2u     /* element is 1 byte (bool) <-> i32 0/1        */
#define GMX_F_BCAST 4u  /* row = 0: one device-resident scalar for all  */
#define GMX_F_STEP 8u   /* row += (t + imm) * gmx_run_args.step_stride: element t + imm of a [T, n] leaf (t = the
                           iteration number of the INNERMOST enclosing OP_LOOP, 0 outside; STOUT: element t, inside a
                           loop only) */
#define GMX_F_FLAT 16u  /* with GMX_F_STEP, inside nested loops: t is the ROW-MAJOR index over all enclosing loops —
                           t0 * n1 + t1 two deep (element (t0, t1) of a [T0, T1, n] leaf: a plate of scans),
                           (t0 * n1 + t1) * n2 + t2 three deep.  In an outermost loop it is the same as GMX_F_STEP alone. */

#define GMX_F_IDX 32u   /* row += ((int) r[imm & 0xff] + (imm >> 8)) * step_stride: element of a [T, n] leaf chosen by a
                           REGISTER (or pool entry) at run time — `means[z]` with z a draw — at any loop depth; the loop's
                           iteration number takes no part (not combined with GMX_F_STEP / GMX_F_FLAT).  The encoder clamps
                           the index to the leaf (jax clamps a gather's indices; and no lane may read outside its rows). */

enum gmx_op {
  OP_END = 0,
  OP_CONST = 1,   // r[dst] = imm
  OP_UNI = 2,     // r[dst] = uni[imm]
  OP_LDIN = 3,    // r[dst] = in[a][row], flags in b (GMX_F_STEP: imm = element offset; GMX_F_IDX: imm = index reg | offset << 8)
  OP_LDTAB = 4,   // r[dst] = tab[a][(int)r[b] + (int)imm]
  OP_STOUT = 5,   // out[a][i] = r[b], flags in dst (GMX_F_STEP: element t + imm of a [T, n] leaf — outside a loop t = 0,
                  // so imm alone names the element: K registers spilled into one [K, n] scratch leaf)
  OP_LDKEY = 6,   // (r[dst], r[dst+1]) = particle key
  OP_KDERIVE = 7, // (r[dst], r[dst+1]) = threefry(key r[a..a+1], ctr (0, imm))  == fold_in / split child
  OP_KDERIVER = 8,// same with ctr (0, (uint)r[b])
  OP_LDIDX = 9,   // r[dst] = (i32) global particle index (index_offset + i)
  OP_MOV = 10,
  OP_ADD = 11, OP_SUB = 12, OP_MUL = 13, OP_DIV = 14, OP_MIN = 15, OP_MAX = 16, OP_POW = 17,
  OP_NEG = 20, OP_ABS = 21, OP_EXP = 22, OP_LOG = 23, OP_LOG1P = 24, OP_SQRT = 25,
  OP_SIN = 26, OP_COS = 27, OP_TANH = 28, OP_SIGMOID = 29, OP_SOFTPLUS = 30,
  OP_FLOOR = 31, OP_LGAMMA = 32, OP_SQUARE = 33, OP_RECIP = 34, OP_CEIL = 35, OP_ROUND = 36,
  OP_FLT = 40, OP_FLE = 41, OP_FGT = 42, OP_FGE = 43, OP_FEQ = 44, OP_FNE = 45,
  OP_IEQ = 46, OP_INE = 47, OP_ILT = 48, OP_ILE = 49, OP_IGT = 50, OP_IGE = 51,
  OP_AND = 52, OP_OR = 53, OP_NOT = 54, OP_XOR = 55,
  OP_SEL = 56,    // r[dst] = r[c] ? r[a] : r[b]
  OP_I2F = 57, OP_F2I = 58,
  OP_IADD = 60, OP_ISUB = 61, OP_IMUL = 62, OP_INEG = 63,
  // samplers: key = r[c..c+1], element counter e = imm >> 8
  OP_S_NORMAL = 70,   // a = loc, b = scale
  OP_S_UNIFORM = 71,  // a = low, b = high
  OP_S_FLIP = 72,     // a = p            -> i32
  OP_S_BERNL = 73,    // a = logits       -> i32
  OP_S_BETA = 74,     // a = c1, b = c0
  OP_S_LOGGAMMA = 76, // a = concentration; log of a Gamma(a, 1) draw from key split_child(r[c..c+1], e) (Dirichlet)
  OP_S_CATSTEP = 75,  // state (r[dst] best f32, r[dst+1] idx i32); a = logit; b = ctr reg (i32); key c; category e
  // log-densities: value x = r[c]
  OP_L_NORMAL = 80, OP_L_UNIFORM = 81, OP_L_FLIP = 82, OP_L_BERNL = 83, OP_L_BETA = 84,
  // block reductions into red_out[blockIdx][0..1]
  OP_REDMAX = 90,     // column 0 = max over the block of r[a]
  OP_REDLSE = 91,     // column 0 = max, column 1 = sum exp(r[a] - max)
  // counted loop (launch-uniform trip count >= 1)
  OP_LOOP = 100,      // imm = trip count: the block up to the matching OP_ENDLOOP runs imm times, t = 0 .. imm-1 (loops
                      // nest up to three deep: a long scan inside a large plate, a plate of plates of plates)
  OP_ENDLOOP = 101,
  OP_LDT = 102,       // r[dst] = (i32) t, the iteration number (0 outside a loop)
  // a SECOND per-particle key in one program (two generative-function calls chained into one launch, each with its
  // own launch key): (r[dst], r[dst+1]) = split((r[a], r[b]), *)[index_offset + i] — what OP_LDKEY does in
  // GMX_KEY_SPLIT mode for gmx_run_args.key0 / key1, here for a key whose two words are launch values
  OP_KSPLITU = 103,
  OP__COUNT = 104
};
