// gmx_kernels.hip — gfx950 kernels + the C-ABI of include/genmi.h.
//
// Kernel inventory (roofline notes in DESIGN.md §4):
//   k_vm<Regs>            site-program interpreter, 1 thread / particle            (ALU + HBM)
//   gmx_jit_kernel        the same program specialised by hiprtc (gmx_jit.h)       (VALU issue)
//   k_lse_tiles/_final    deterministic two-stage log-sum-exp                      (HBM)
//   k_lse_rows            one wave per row (many short rows)                       (HBM)
//   k_weight_cdf          block-floating-point integer CDF, single-pass chained scan (HBM)
//   k_offspring           source-centric offspring ranges from a CDF array         (HBM, latency)
//   k_ancestors           128-bit exact inverse-CDF search (multinomial, sharded)  (L2 latency)
//   k_tile_stats          (max, fixed-point weight sum) per 1024-particle tile     (HBM)
//   k_offspring_tile      ancestors from log-weights + tile statistics, no CDF array (VALU issue)
//   k_shard_plan/_route/_step<TILES>, k_shard_totals   global resampling across ranks
//   k_gather/k_select     multi-leaf row gather / masked select                    (HBM)
//   k_categorical_rows    Gumbel-max per row                                       (ALU)
//   small key kernels     split / fold_in / random_bits / mh_accept
//
// Wavefront = 64 everywhere; blocks are 256 threads (4 waves, one per SIMD).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <new>
#include <atomic>
#include <string>
#include <vector>

#include "gmx_block.h"
#include "gmx_vm.h"
#include "gmx_resample.h"
#include "gmx_peer.h"
#include "gmx_sorted.h"
#include "gmx_offspring.h"
#include "gmx_shard_fill.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int gmx_fail(const char* fmt, const char* a = "", long long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return 1;
}
#define GMX_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return gmx_fail("%s (hip error %lld)", hipGetErrorString(_e), (long long)_e); \
  } while (0)

extern "C" int gmx_version(void) { return GMX_ABI_VERSION; }
extern "C" size_t gmx_run_args_bytes(void) { return sizeof(gmx_run_args); }
extern "C" const char* gmx_last_error(void) { return g_err; }

extern "C" void gmx_threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                      uint32_t out[2]) {
  gmx_threefry2x32(k0, k1, c0, c1, &out[0], &out[1]);
}

// ---------------------------------------------------------------------------
// site-program interpreter
// ---------------------------------------------------------------------------
struct DevCtx {
  // lane l of each of these holds table entry l (filled once per wave)
  uint32_t prog_a0, prog_b0, prog_a1, prog_b1;   // instruction words 0..63 / 64..127
  uint32_t pool_v;                               // gmx_run_args.uni[l]
  uint32_t in_lo, in_hi, out_lo, out_hi;         // slot pointers
  const uint32_t* code;
  const gmx_run_args* A;
  float* red_out;
  float* lds4;
  __device__ __forceinline__ void init(const uint32_t* code_, uint32_t n_instr, const gmx_run_args* A_,
                                        float* lds4_) {
    code = code_; A = A_; red_out = A_->red_out_d; lds4 = lds4_;
    const uint32_t lane = threadIdx.x & 63u;
    const uint2* c2 = reinterpret_cast<const uint2*>(code_);
    uint2 w = (lane < n_instr) ? c2[lane] : make_uint2(0u, 0u);
    prog_a0 = w.x; prog_b0 = w.y;
    uint2 w1 = (lane + 64u < n_instr) ? c2[lane + 64u] : make_uint2(0u, 0u);
    prog_a1 = w1.x; prog_b1 = w1.y;
    pool_v = A_->uni[lane];
    uint64_t pi = (uint64_t)A_->in_d[lane], po = (uint64_t)A_->out_d[lane];
    in_lo = (uint32_t)pi; in_hi = (uint32_t)(pi >> 32);
    out_lo = (uint32_t)po; out_hi = (uint32_t)(po >> 32);
  }
  __device__ __forceinline__ void fetch(uint32_t pc, uint32_t* w0, uint32_t* w1) const {
    if (pc < 64u) {
      *w0 = (uint32_t)__builtin_amdgcn_readlane(prog_a0, pc);
      *w1 = (uint32_t)__builtin_amdgcn_readlane(prog_b0, pc);
    } else if (pc < 128u) {
      *w0 = (uint32_t)__builtin_amdgcn_readlane(prog_a1, pc - 64u);
      *w1 = (uint32_t)__builtin_amdgcn_readlane(prog_b1, pc - 64u);
    } else {
      *w0 = __builtin_amdgcn_readfirstlane(code[2u * pc]);
      *w1 = __builtin_amdgcn_readfirstlane(code[2u * pc + 1u]);
    }
  }
  __device__ __forceinline__ uint32_t pool(uint32_t i) const {
    return (uint32_t)__builtin_amdgcn_readlane(pool_v, i);
  }
  __device__ __forceinline__ const void* in_ptr(uint32_t s) const {
    // readlane returns int: go through uint32_t so the low half is not sign-extended
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(in_lo, s), hi = (uint32_t)__builtin_amdgcn_readlane(in_hi, s);
    return (const void*)(((uint64_t)hi << 32) | (uint64_t)lo);
  }
  __device__ __forceinline__ void* out_ptr(uint32_t s) const {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(out_lo, s), hi = (uint32_t)__builtin_amdgcn_readlane(out_hi, s);
    return (void*)(((uint64_t)hi << 32) | (uint64_t)lo);
  }
  __device__ __forceinline__ const void* tab_ptr(uint32_t s) const { return A->tab_d[s]; }
  __device__ __forceinline__ void red_max(float x, bool active) { gmx_red_max(red_out, lds4, blockIdx.x, x, active); }
  __device__ __forceinline__ void red_lse(float x, bool active) {
    gmx_red_lse(red_out, lds4, blockIdx.x, gridDim.x, x, active);
  }
};

template <class Regs, bool FULL>
__global__ void __launch_bounds__(GMX_BLOCK)
k_vm(const uint32_t* __restrict__ code, uint32_t n_instr, int64_t n, const gmx_run_args A) {
  __shared__ float lds4[4];
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  DevCtx ctx;
  ctx.init(code, n_instr, &A, lds4);
  gmx_vm_run<Regs, FULL, -1, DevCtx>(n_instr, i, i < n, A, ctx);
}
// The 16-register interpreter of the common opcodes, held to 7 waves per SIMD (72 VGPRs, 7 dwords of scratch in cold
// paths): it had drifted from 70 to 86 VGPRs (5 waves) as opcodes were added.  Measured on MI355X, everything interpreted
// (GENMI_JIT=0): config 5 3.74 -> 3.30 ms, config 3 242 -> 229 us / step, config 2 6.62 -> 6.50 ms.  The other three
// instantiations keep the compiler's allocation: under the same bound they spill 200-250 bytes per lane (config 4's
// importance launch: 3.7 -> 41 ms).
__global__ void __launch_bounds__(GMX_BLOCK) __attribute__((amdgpu_waves_per_eu(7, 8)))
k_vm_lean(const uint32_t* __restrict__ code, uint32_t n_instr, int64_t n, const gmx_run_args A) {
  __shared__ float lds4[4];
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  DevCtx ctx;
  ctx.init(code, n_instr, &A, lds4);
  gmx_vm_run<gmx_regs_vgpr<16>, false, -1, DevCtx>(n_instr, i, i < n, A, ctx);
}

struct gmx_program {
  uint32_t* code_d;
  uint32_t n_instr, n_regs, n_in, n_out, n_uni, n_tab, n_const, n_dyn;
  bool uses_key, uses_red, uses_lse, uses_gather, needs_full;
  bool uses_loop = false, uses_step = false;
  uint32_t n_redmax = 0;             // OP_REDMAX instructions (tile statistics need exactly one)
  std::vector<uint32_t> code_h;      // instruction words (host copy, for specialisation)
  std::vector<uint32_t> consts;      // pool entries n_dyn..
  hipModule_t jit_module = nullptr;  // specialised kernel (gmx_program_specialize)
  hipFunction_t jit_fn = nullptr;
  int jit_pp = 1;                    // particles per thread of the specialised kernel
  uint64_t jit_code_hash = 0;        // FNV-1a of the loaded code object (gmx_program_code_hash)
  bool fuse_rs = false;              // gmx_program_set_fuse_resample: the specialised kernel can resample first
  bool fuse_rs_loop = false;         // gmx_program_set_fuse_resample_loop: ... with fewer workgroups than tiles (n > 2^20)
  bool fuse_sh = false;              // gmx_program_set_fuse_shard_step: ... or route a sharded step first
  bool jit_gathers_pre = false;      // every gathered load of the specialised kernel goes through the prologue's ancestors
  int64_t jit_resident_blocks = 0;   // workgroups of the specialised kernel one device holds AT ONCE (occupancy x CUs)
  bool background = false;           // gmx_program_set_background: wave priority 0 ...
  unsigned lds_pad = 0;              // ... and this much unused dynamic LDS per workgroup (a residency cap)
};

static int parse_program(const uint32_t* blob, size_t n_words, gmx_program& P) {
  if (!blob) return gmx_fail("gmx_program_create: null argument%s");
  if (n_words < GMX_PROG_HEADER_WORDS) return gmx_fail("gmx_program_create: blob too short%s");
  if (blob[0] != GMX_PROG_MAGIC) return gmx_fail("gmx_program_create: bad magic%s");
  if (blob[1] != GMX_PROG_VERSION) return gmx_fail("gmx_program_create: bad version%s");
  uint32_t n_instr = blob[2], n_regs = blob[3];
  uint32_t n_const = blob[8], n_dyn = blob[9];
  if (n_words != GMX_PROG_HEADER_WORDS + 2ull * n_instr + n_const)
    return gmx_fail("gmx_program_create: length does not match n_instr / n_const%s");
  if ((uint64_t)n_dyn + n_const != blob[6])
    return gmx_fail("gmx_program_create: n_uni != n_dyn + n_const%s");
  if (n_regs == 0 || n_regs > GMX_MAX_REGS)
    return gmx_fail("gmx_program_create: n_regs must be in [1,64] (got %s%lld)", "", n_regs);
  if (blob[4] > GMX_MAX_IN || blob[5] > GMX_MAX_OUT || blob[6] > GMX_MAX_UNI || blob[7] > GMX_MAX_TAB)
    return gmx_fail("gmx_program_create: slot count exceeds ABI limits%s");
  P.code_d = nullptr;
  P.uses_key = P.uses_red = P.uses_lse = P.uses_gather = P.needs_full = false;
  P.n_instr = n_instr; P.n_regs = n_regs; P.n_const = n_const; P.n_dyn = n_dyn;
  P.n_in = blob[4]; P.n_out = blob[5]; P.n_uni = blob[6]; P.n_tab = blob[7];
  // validate every instruction: register / slot indices must be in range so
  // the kernel never needs a bounds check
  const uint32_t* ins = blob + GMX_PROG_HEADER_WORDS;
  bool in_loop = false;
  int loop_depth = 0;
  for (uint32_t pc = 0; pc < n_instr; ++pc) {
    uint32_t w0 = ins[2 * pc], w1 = ins[2 * pc + 1];
    uint32_t op = w0 & 0xff, dst = (w0 >> 8) & 0xff, a = (w0 >> 16) & 0xff, b = w0 >> 24;
    uint32_t c = w1 & 0xff;
    bool ok = true;
    auto D = [&](uint32_t r) { return r < n_regs; };                        // destination register
    auto R2 = [&](uint32_t r) { return r + 1 < n_regs; };                  // register pair (keys)
    auto R = [&](uint32_t r) {                                              // source: register or pool entry
      return r < n_regs || (r >= GMX_POOL_BASE && r - GMX_POOL_BASE < P.n_uni);
    };
    if (gmx_op_needs_full(op)) P.needs_full = true;
    switch (op) {
      case OP_END: break;
      case OP_CONST: case OP_LDIDX: case OP_LDT: ok = D(dst); break;
      case OP_LOOP: ok = loop_depth < 3 && w1 >= 1u; ++loop_depth; in_loop = true; P.uses_loop = true; break;   // counted, <= 3 deep
      case OP_ENDLOOP: ok = loop_depth > 0; --loop_depth; in_loop = loop_depth > 0; break;
      case OP_UNI: ok = D(dst) && w1 < P.n_uni; break;
      case OP_LDIN:
        ok = D(dst) && a < P.n_in && (!(b & GMX_F_IDX) || (R(c) && !(b & (GMX_F_STEP | GMX_F_FLAT | GMX_F_BCAST))));
        if (b & GMX_F_GATHER) P.uses_gather = true;
        break;
      case OP_LDTAB: ok = D(dst) && R(b) && a < P.n_tab; break;
      case OP_STOUT: ok = R(b) && a < P.n_out; break;
      case OP_LDKEY: ok = R2(dst); P.uses_key = true; break;
      case OP_KDERIVE: ok = R2(dst) && R2(a); break;
      case OP_KDERIVER: ok = R2(dst) && R2(a) && R(b); break;
      case OP_KSPLITU: ok = R2(dst) && R(a) && R(b); break;
      case OP_MOV: case OP_NEG: case OP_ABS: case OP_EXP: case OP_LOG: case OP_LOG1P:
      case OP_SQRT: case OP_SIN: case OP_COS: case OP_TANH: case OP_SIGMOID:
      case OP_SOFTPLUS: case OP_FLOOR: case OP_CEIL: case OP_ROUND: case OP_LGAMMA:
      case OP_SQUARE: case OP_RECIP: case OP_NOT: case OP_I2F: case OP_F2I: case OP_INEG:
        ok = D(dst) && R(a); break;
      case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_MIN: case OP_MAX: case OP_POW:
      case OP_FLT: case OP_FLE: case OP_FGT: case OP_FGE: case OP_FEQ: case OP_FNE:
      case OP_IEQ: case OP_INE: case OP_ILT: case OP_ILE: case OP_IGT: case OP_IGE:
      case OP_AND: case OP_OR: case OP_XOR: case OP_IADD: case OP_ISUB: case OP_IMUL:
        ok = D(dst) && R(a) && R(b); break;
      case OP_SEL: ok = D(dst) && R(a) && R(b) && R(c); break;
      case OP_S_NORMAL: case OP_S_UNIFORM: case OP_S_BETA:
        ok = D(dst) && R(a) && R(b) && R2(c); break;
      case OP_S_FLIP: case OP_S_BERNL: case OP_S_LOGGAMMA: ok = D(dst) && R(a) && R2(c); break;
      case OP_S_CATSTEP: ok = R2(dst) && R(a) && R(b) && R2(c); break;
      case OP_L_NORMAL: case OP_L_UNIFORM: case OP_L_BETA:
        ok = D(dst) && R(a) && R(b) && R(c); break;
      case OP_L_FLIP: case OP_L_BERNL: ok = D(dst) && R(a) && R(c); break;
      case OP_REDMAX: case OP_REDLSE:
        ok = R(a); P.uses_red = true;
        if (op == OP_REDLSE) P.uses_lse = true; else ++P.n_redmax;
        break;
      default: ok = false;
    }
    if ((op == OP_LDIN && (b & (GMX_F_STEP | GMX_F_IDX))) || (op == OP_STOUT && (dst & GMX_F_STEP))) {
      P.uses_step = true;      // (a step-indexed store outside a loop writes element imm: t = 0 there)
    }
    if (!ok) return gmx_fail("gmx_program_create: invalid instruction%s at pc %lld", "", pc);
  }
  if (in_loop) return gmx_fail("gmx_program_create: OP_LOOP without OP_ENDLOOP%s");
  P.code_h.assign(ins, ins + 2 * (size_t)n_instr);
  P.consts.assign(ins + 2 * (size_t)n_instr, ins + 2 * (size_t)n_instr + n_const);
  return 0;
}

extern "C" int gmx_program_create(const uint32_t* blob, size_t n_words, gmx_program** out) {
  if (!out) return gmx_fail("gmx_program_create: null argument%s");
  gmx_program P;
  if (parse_program(blob, n_words, P)) return 1;
  const uint32_t n_instr = P.n_instr;
  GMX_HIP(hipMalloc((void**)&P.code_d, sizeof(uint32_t) * 2 * (n_instr ? n_instr : 1)));
  if (n_instr)
    GMX_HIP(hipMemcpy(P.code_d, P.code_h.data(), sizeof(uint32_t) * 2 * n_instr, hipMemcpyHostToDevice));
  gmx_program* h = new (std::nothrow) gmx_program(P);
  if (!h) return gmx_fail("gmx_program_create: out of host memory%s");
  *out = h;
  return 0;
}

extern "C" int gmx_program_destroy(gmx_program* p) {
  if (!p) return 0;
  if (p->code_d) (void)hipFree(p->code_d);
  if (p->jit_module) (void)hipModuleUnload(p->jit_module);
  delete p;
  return 0;
}


// ---------------------------------------------------------------------------
// specialisation: the interpreter partially evaluated by hiprtc
// ---------------------------------------------------------------------------
#include "gmx_embed.inc"   // generated by __graft_entry__.build(): the device headers as strings

static bool jit_enabled() {
  const char* e = getenv("GENMI_JIT");
  return !(e && e[0] == '0');
}

// particles per thread: as much ILP as the register budget allows at 8 waves / SIMD
static int jit_pp_for(const gmx_program* p) {
  // Measured on MI355X (BASELINE config 2, 1e6 particles): 4 particles / thread do not raise the VALU
  // issue rate of the integer-heavy Threefry stream (isolated launch 11.8 -> 13.7 us) but the kernel
  // is 0.9 us SHORTER inside the sweep (a quarter of the workgroups to schedule against the cold
  // inputs the resampling kernels just wrote): sweep 2843 -> 2730 us; 2, 3 and 8 are slower.  Only
  // for small programs: registers scale with the particle count per thread.
  // OP_REDLSE partials (max, sum exp) are per 256-particle group by definition: one particle per thread.
  return (p->n_regs <= 16 && p->n_instr <= 96 && !p->uses_lse) ? 4 : 1;   // 96: an MH move chained with the extension (~65)
}

// prefetch plan of a specialised kernel (gmx_jit.h): the distinct (slot, flags) of the program's per-particle
// OP_LDIN instructions, at most JIT_MAX_PRE of them (PP registers each, held from the top of the kernel)
#define JIT_MAX_PRE 8
static std::string jit_source(const gmx_program* p, bool* gathers_prefetched = nullptr) {
  std::string s = p->background ? "#define GMX_JIT_BACKGROUND 1\n" : "";
  if (const char* f_ = getenv("GENMI_JIT_FAULT")) { if (f_[0] == '1') s += "#define GMX_JIT_FAULT 1\n"; }   // (tests: a kernel that
                                                        // stores every 32-bit word with its lowest bit flipped — what the first-launch cross-check must catch)
  if (p->fuse_rs) s += "#define GMX_JIT_RS 1\n";
  if (p->fuse_rs_loop) s += "#define GMX_JIT_RS_LOOP 1\n";
  if (p->fuse_sh) s += "#define GMX_JIT_SH 1\n";
  s += "#include \"gmx_jit.h\"\n";
  char buf[128];
  s += "__device__ static constexpr uint32_t GMX_JIT_CONST[] = {";
  for (size_t k = 0; k < p->consts.size(); ++k) { snprintf(buf, sizeof(buf), "0x%08xu,", p->consts[k]); s += buf; }
  s += "0u};\n";
  struct pre_t { uint32_t slot, flags; };
  std::vector<pre_t> pres;
  std::vector<int> pre_of(p->n_instr, -1);
  bool any_gather = false;
  uint32_t first_gather_pc = p->n_instr, first_key_pc = p->n_instr;
  bool fits = true, seen_loop = false;
  for (uint32_t pc = 0; pc < p->n_instr && fits; ++pc) {
    const uint32_t w0 = p->code_h[2 * pc], op = w0 & 0xffu, a = (w0 >> 16) & 0xffu, b = w0 >> 24;
    if ((op == OP_LDKEY || op == OP_KDERIVE) && first_key_pc == p->n_instr && !seen_loop) first_key_pc = pc;
    if (op == OP_LOOP) seen_loop = true;          // a key derived inside the loop is no place for one-off prefetches
    if (op != OP_LDIN || (b & (GMX_F_BCAST | GMX_F_STEP | GMX_F_IDX))) continue;     // step- / register-indexed rows: loaded in place
    int k = -1;
    for (size_t j = 0; j < pres.size(); ++j)
      if (pres[j].slot == a && pres[j].flags == b) k = (int)j;
    if (k < 0) {
      if (pres.size() == JIT_MAX_PRE) { fits = false; break; }
      k = (int)pres.size();
      pres.push_back({a, b});
    }
    pre_of[pc] = k;
    if (b & GMX_F_GATHER) { any_gather = true; if (pc < first_gather_pc) first_gather_pc = pc; }
  }
  if (const char* e_ = getenv("GENMI_JIT_NOPRE")) { if (e_[0] == '1' && !p->fuse_rs && !p->fuse_sh) fits = false; }   // (diagnosis)
  if (!fits) { pres.clear(); pre_of.assign(p->n_instr, -1); any_gather = false; }
  if (gathers_prefetched) {          // every gathered load goes through the prologue's ancestors (GMX_JIT_PRE_ANC)
    *gathers_prefetched = any_gather;
    for (uint32_t pc = 0; pc < p->n_instr; ++pc) {
      const uint32_t w0 = p->code_h[2 * pc];
      if ((w0 & 0xffu) == OP_LDIN && ((w0 >> 24) & GMX_F_GATHER) && pre_of[pc] < 0) *gathers_prefetched = false;
    }
  }
  // second-stage (gathered) loads go behind the first key derivation when that comes before their first use
  const uint32_t gpos = (any_gather && first_key_pc < first_gather_pc) ? first_key_pc + 1 : 0;
  snprintf(buf, sizeof(buf), "GMX_JIT_BEGIN(%u, %s, %u, %d, %d)\n", p->n_regs < 16 ? 16u : (p->n_regs < 32 ? 32u : 64u),
           p->needs_full ? "true" : "false", p->n_dyn, jit_pp_for(p), (int)pres.size());
  s += buf;
  if (any_gather) s += "  GMX_JIT_PRE_ANC\n";
  for (size_t k = 0; k < pres.size(); ++k)
    if (!(pres[k].flags & GMX_F_GATHER)) {
      snprintf(buf, sizeof(buf), "  GMX_JIT_PRE(%d, %u, %d)\n", (int)k, pres[k].slot, (pres[k].flags & GMX_F_U8) ? 1 : 0);
      s += buf;
    }
  auto gathers = [&]() {
    for (size_t k = 0; k < pres.size(); ++k)
      if (pres[k].flags & GMX_F_GATHER) {
        snprintf(buf, sizeof(buf), "  GMX_JIT_PRE_G(%d, %u, %d)\n", (int)k, pres[k].slot, (pres[k].flags & GMX_F_U8) ? 1 : 0);
        s += buf;
      }
    s += "  GMX_JIT_FENCE\n";
  };
  if (!pres.empty()) s += "  GMX_JIT_FENCE\n";
  if (any_gather && gpos == 0) gathers();
  int jit_depth = 0;
  for (uint32_t pc = 0; pc < p->n_instr; ++pc) {
    const uint32_t op_ = p->code_h[2 * pc] & 0xffu;
    if (op_ == OP_LOOP) {
      static const char* const open_[3] = {"  GMX_JIT_LOOP(%uu)\n", "  GMX_JIT_LOOP2(%uu)\n", "  GMX_JIT_LOOP3(%uu)\n"};
      snprintf(buf, sizeof(buf), open_[jit_depth < 2 ? jit_depth : 2], p->code_h[2 * pc + 1]);
      ++jit_depth;
    } else if (op_ == OP_ENDLOOP) {
      static const char* const close_[3] = {"  GMX_JIT_ENDLOOP\n", "  GMX_JIT_ENDLOOP2\n", "  GMX_JIT_ENDLOOP3\n"};
      --jit_depth;
      snprintf(buf, sizeof(buf), "%s", close_[jit_depth < 2 ? (jit_depth < 0 ? 0 : jit_depth) : 2]);
    }
    else if (pre_of[pc] >= 0)
      snprintf(buf, sizeof(buf), "  GMX_JIT_LDPRE(%u, %d)\n", (p->code_h[2 * pc] >> 8) & 0xffu, pre_of[pc]);
    else
      snprintf(buf, sizeof(buf), "  GMX_JIT_OP(0x%08xu, 0x%08xu)\n", p->code_h[2 * pc], p->code_h[2 * pc + 1]);
    s += buf;
    if (any_gather && gpos == pc + 1) gathers();
  }
  s += "GMX_JIT_END\n";
  return s;
}

extern "C" int gmx_program_is_specialized(const gmx_program* p) { return p && p->jit_fn ? 1 : 0; }
static std::atomic<int64_t> g_jit_rejected{0};
extern "C" int64_t gmx_jit_rejected_count(void) { return g_jit_rejected.load(); }
extern "C" int gmx_program_despecialize(gmx_program* p, const char* why) {
  if (!p) return gmx_fail("gmx_program_despecialize: null program%s");
  if (!p->jit_fn) return 0;
  (void)hipDeviceSynchronize();              // nothing of this module may still be running
  (void)hipModuleUnload(p->jit_module);
  p->jit_fn = nullptr; p->jit_module = nullptr; p->jit_code_hash = 0;
  ++g_jit_rejected;
  (void)gmx_fail("specialised kernel rejected, the interpreter takes over: %s", why ? why : "");
  return 0;
}
extern "C" uint64_t gmx_program_code_hash(const gmx_program* p) { return p && p->jit_fn ? p->jit_code_hash : 0ull; }
extern "C" int gmx_program_writes_tile_stats(const gmx_program* p) {
  return p && p->jit_fn && p->jit_pp == 4 && p->n_redmax == 1 && !p->uses_lse ? 1 : 0;
}

extern "C" int gmx_program_set_fuse_resample(gmx_program* p) {
  if (!p) return gmx_fail("gmx_program_set_fuse_resample: null program%s");
  if (p->jit_fn) return gmx_fail("gmx_program_set_fuse_resample: the program is already specialised%s");
  p->fuse_rs = true;
  return 0;
}
extern "C" int gmx_program_set_fuse_resample_loop(gmx_program* p) {
  if (!p) return gmx_fail("gmx_program_set_fuse_resample_loop: null program%s");
  if (p->jit_fn) return gmx_fail("gmx_program_set_fuse_resample_loop: the program is already specialised%s");
  p->fuse_rs = true;
  p->fuse_rs_loop = true;
  return 0;
}
// workgroups of a LOOPED launch: what the device holds at once, at most 1024 (4 per CU: the shape the one-tile-per-workgroup
// launch of 2^20 particles has always had) and at most the tiles there are
static int64_t rs_loop_grid(const gmx_program* p, int64_t tiles) {
  int64_t g = p->jit_resident_blocks > 0 ? p->jit_resident_blocks : 256;
  if (g > 1024) g = 1024;

  return tiles < g ? tiles : g;
}
extern "C" int gmx_program_fuses_resample(const gmx_program* p) {
  return p && p->jit_fn && p->fuse_rs && p->jit_pp == 4 && p->uses_gather && p->jit_gathers_pre ? 1 : 0;
}
extern "C" int gmx_program_set_fuse_shard_step(gmx_program* p) {
  if (!p) return gmx_fail("gmx_program_set_fuse_shard_step: null program%s");
  if (p->jit_fn) return gmx_fail("gmx_program_set_fuse_shard_step: the program is already specialised%s");
  p->fuse_sh = true;
  return 0;
}
extern "C" int gmx_program_fuses_shard_step(const gmx_program* p) {
  return p && p->jit_fn && p->fuse_sh && p->jit_pp == 4 && p->uses_gather && p->jit_gathers_pre ? 1 : 0;
}
extern "C" int64_t gmx_program_resident_particles(const gmx_program* p) {
  if (!p || !p->jit_fn) return 0;
  if (p->fuse_rs_loop && p->jit_pp == 4 && p->jit_resident_blocks > 0)           // a looped launch walks its tiles: 16 per workgroup
    return rs_loop_grid(p, 1024) * 16 * (int64_t)GMX_BLOCK * 4;
  return p->jit_resident_blocks * (int64_t)GMX_BLOCK * (int64_t)p->jit_pp;
}

// A BACKGROUND program: work that depends on nothing a dependent chain of launches produces (the standard-normal
// draws of the NEXT SMC steps: keys only), issued on a second stream beside that chain.  Its specialised kernel
//   * keeps wave priority 0 while the chain's kernels run at GMX_CHAIN_PRIO (gmx_block.h), and
//   * asks for `lds_pad` bytes of dynamic LDS it never touches: a cap on how many of its workgroups a CU holds
//     (160 KB of LDS per CU: 56 000 bytes -> two), so that the chain's kernels always find free wave slots —
//     without it a 29-VGPR noise kernel fills all 8 workgroup slots of every CU and the chain waits behind it.
// Set before gmx_program_specialize; the interpreter ignores it.
extern "C" int gmx_program_set_background(gmx_program* p, uint32_t lds_pad) {
  if (!p) return gmx_fail("gmx_program_set_background: null program%s");
  if (p->jit_fn) return gmx_fail("gmx_program_set_background: the program is already specialised%s");
  if (lds_pad > 160u * 1024u) return gmx_fail("gmx_program_set_background: lds_pad above the 160 KB of a CU%s");
  p->background = true;
  p->lds_pad = lds_pad;
  return 0;
}

// ---- on-disk cache of specialised code objects -----------------------------
// hiprtc needs 0.5-2 s per program; the code object depends only on the generated translation unit,
// the embedded device headers and the compiler, so it is kept under
//   $GENMI_JIT_CACHE  (default $XDG_CACHE_HOME/genjax_amd/jit or $HOME/.cache/genjax_amd/jit; "0" disables)
// as <fnv1a-64 of all three>.co, written to a temporary name and renamed (concurrent ranks are safe).
static uint64_t fnv1a(uint64_t h, const char* s, size_t n) {
  for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ull; }
  return h;
}

static std::string jit_cache_dir() {
  const char* e = getenv("GENMI_JIT_CACHE");
  if (e && e[0] == '0' && e[1] == '\0') return "";
  if (e && e[0]) return e;
  const char* x = getenv("XDG_CACHE_HOME");
  if (x && x[0]) return std::string(x) + "/genjax_amd/jit";
  const char* h = getenv("HOME");
  if (h && h[0]) return std::string(h) + "/.cache/genjax_amd/jit";
  return "";
}

static void mkdirs(const std::string& d) {
  for (size_t i = 1; i <= d.size(); ++i)
    if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0755);
}

static std::string jit_cache_path(const std::string& src) {
  std::string dir = jit_cache_dir();
  if (dir.empty()) return "";
  uint64_t h = fnv1a(0xcbf29ce484222325ull, src.data(), src.size());
  for (int k = 0; k < GMX_EMBED_COUNT; ++k) h = fnv1a(h, gmx_embed_src[k], strlen(gmx_embed_src[k]));
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  char buf[64];
  // (e2: entries written before the cache refused a profiled process's builds are not read again)
  snprintf(buf, sizeof(buf), "/%016llx_rtc%d.%d_abi%d_e2.co", (unsigned long long)h, major, minor, GMX_ABI_VERSION);
  return dir + buf;
}

// GENMI_JIT_OPT = -O1 / -O2 (diagnosis: a result that changes with the level names a miscompile) — never
// cached: the cache holds -O3 builds only
static const char* jit_opt_level() {
  const char* e = getenv("GENMI_JIT_OPT");
  return (e && (!strcmp(e, "-O1") || !strcmp(e, "-O2") || !strcmp(e, "-O0"))) ? e : "-O3";
}

static bool jit_cache_read(const std::string& path, std::vector<char>& code) {
  if (strcmp(jit_opt_level(), "-O3") != 0) return false;
  if (path.empty()) return false;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  if (fseek(f, 0, SEEK_END) == 0) {
    long n = ftell(f);
    if (n > 0 && n < (64l << 20) && fseek(f, 0, SEEK_SET) == 0) {
      code.resize((size_t)n);
      ok = fread(code.data(), 1, (size_t)n, f) == (size_t)n;
    }
  }
  fclose(f);
  return ok;
}

// hiprtc inside a process that rocprofv3 has preloaded its tool library into compiles the SAME source to different code
// (measured: a 1160-instruction program at 176 VGPRs / 0.60 ms against 60 VGPRs / 0.35 ms): such a build is used by the
// process that made it and never cached, so that the cache only ever holds what an unprofiled process compiles
static bool jit_under_profiler() {
  const char* e = getenv("ROCP_TOOL_LIBRARIES");
  if (e && e[0]) return true;
  e = getenv("LD_PRELOAD");
  return e && strstr(e, "rocprofiler") != nullptr;
}

static void jit_cache_write(const std::string& path, const std::vector<char>& code) {
  if (path.empty() || jit_under_profiler() || strcmp(jit_opt_level(), "-O3") != 0) return;
  if (const char* f_ = getenv("GENMI_JIT_FAULT")) { if (f_[0] == '1') return; }      // a deliberately wrong kernel is never cached
  mkdirs(path.substr(0, path.rfind('/')));
  char tmp[32];
  snprintf(tmp, sizeof(tmp), ".tmp%ld", (long)getpid());
  std::string t = path + tmp;
  FILE* f = fopen(t.c_str(), "wb");
  if (!f) return;                                   // a read-only cache directory is not an error
  bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}

static int jit_compile(const std::string& src, std::vector<char>& code) {
  hiprtcProgram prog;
  const char* hdr_src[GMX_EMBED_COUNT];
  const char* hdr_name[GMX_EMBED_COUNT];
  for (int k = 0; k < GMX_EMBED_COUNT; ++k) { hdr_src[k] = gmx_embed_src[k]; hdr_name[k] = gmx_embed_name[k]; }
  hiprtcResult rc = hiprtcCreateProgram(&prog, src.c_str(), "gmx_jit_program.hip", GMX_EMBED_COUNT, hdr_src, hdr_name);
  if (rc != HIPRTC_SUCCESS) return gmx_fail("hiprtcCreateProgram: %s", hiprtcGetErrorString(rc));
  const char* opts[4] = {"--offload-arch=gfx950", jit_opt_level(), "-std=c++17", "-ffp-contract=off"};
  rc = hiprtcCompileProgram(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls + 1, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    if (log.size() > 400) log.resize(400);
    return gmx_fail("hiprtcCompileProgram failed: %s", log.c_str());
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code.resize(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return 0;
}

static int jit_load(gmx_program* p, const std::vector<char>& code) {
  hipModule_t mod;
  GMX_HIP(hipModuleLoadData(&mod, code.data()));
  hipFunction_t fn;
  hipError_t e = hipModuleGetFunction(&fn, mod, p->background ? "gmx_jit_background_kernel" : "gmx_jit_kernel");
  if (e != hipSuccess) { (void)hipModuleUnload(mod); return gmx_fail("hipModuleGetFunction: %s", hipGetErrorString(e)); }
  p->jit_module = mod;
  p->jit_fn = fn;
  p->jit_pp = jit_pp_for(p);
  p->jit_code_hash = fnv1a(0xcbf29ce484222325ull, code.data(), code.size());
  if (p->lds_pad > 48 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_pad);
    if (ea != hipSuccess) { (void)hipGetLastError(); p->lds_pad = 48 * 1024; }
  }
  // how many of this kernel's workgroups the device holds at once: what the resample-first launch (gmx_run_args.rs),
  // whose workgroups wait for each other, must fit into — asked of the runtime for THIS code object on THIS device
  // (a partitioned or smaller device, a heavier step model), never assumed
  p->jit_resident_blocks = 0;
  int per_cu = 0, dev = 0, cus = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, GMX_BLOCK, (size_t)p->lds_pad) == hipSuccess &&
      hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    p->jit_resident_blocks = (int64_t)per_cu * (int64_t)cus;
  else
    (void)hipGetLastError();
  return 0;
}

extern "C" int gmx_program_specialize(gmx_program* p) {
  if (!p) return gmx_fail("gmx_program_specialize: null program%s");
  if (p->jit_fn) return 0;
  if (!jit_enabled()) return gmx_fail("gmx_program_specialize: disabled by GENMI_JIT=0%s");
  if (p->n_instr == 0 || p->n_instr > 8192) return gmx_fail("gmx_program_specialize: program size out of range%s");
  const std::string src = jit_source(p, &p->jit_gathers_pre);
  const std::string path = jit_cache_path(src);
  std::vector<char> code;
  if (jit_cache_read(path, code)) {
    if (jit_load(p, code) == 0) return 0;
    (void)hipGetLastError();                 // a damaged entry just recompiles (and leaves no sticky error behind)
  }
  if (jit_compile(src, code)) return 1;
  if (jit_load(p, code)) return 1;
  jit_cache_write(path, code);
  return 0;
}

// offline entry point used by tests / tools on machines without a GPU: compile
// the specialised kernel and return its code-object size (0 on failure).
// flags: 1 = gmx_program_set_fuse_resample, 2 = gmx_program_set_background
extern "C" size_t gmx_specialize_dryrun2(const uint32_t* blob, size_t n_words, int flags, char* log_out, size_t log_cap,
                                         char* code_out, size_t code_cap);
extern "C" size_t gmx_specialize_dryrun(const uint32_t* blob, size_t n_words, char* log_out, size_t log_cap,
                                        char* code_out, size_t code_cap) {
  return gmx_specialize_dryrun2(blob, n_words, 0, log_out, log_cap, code_out, code_cap);
}
extern "C" size_t gmx_specialize_dryrun2(const uint32_t* blob, size_t n_words, int flags, char* log_out, size_t log_cap,
                                         char* code_out, size_t code_cap) {
  gmx_program P;
  if (parse_program(blob, n_words, P)) { if (log_out && log_cap) snprintf(log_out, log_cap, "%s", g_err); return 0; }
  P.fuse_rs = (flags & 1) != 0 || (flags & 8) != 0;
  P.fuse_rs_loop = (flags & 8) != 0;
  P.background = (flags & 2) != 0;
  P.fuse_sh = (flags & 4) != 0;
  const gmx_program* p = &P;
  std::string src = jit_source(p);
  hiprtcProgram prog;
  const char* hdr_src[GMX_EMBED_COUNT];
  const char* hdr_name[GMX_EMBED_COUNT];
  for (int k = 0; k < GMX_EMBED_COUNT; ++k) { hdr_src[k] = gmx_embed_src[k]; hdr_name[k] = gmx_embed_name[k]; }
  if (hiprtcCreateProgram(&prog, src.c_str(), "gmx_jit_program.hip", GMX_EMBED_COUNT, hdr_src, hdr_name) != HIPRTC_SUCCESS)
    return 0;
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
  hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  if (log_out && log_cap) {
    std::string log(ls + 1, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    snprintf(log_out, log_cap, "%s", log.c_str());
  }
  size_t cs = 0;
  if (rc == HIPRTC_SUCCESS) {
    hiprtcGetCodeSize(prog, &cs);
    if (code_out && code_cap >= cs) hiprtcGetCode(prog, code_out);
  }
  hiprtcDestroyProgram(&prog);
  return cs;
}

// Rows of block partials one launch over n particles writes (plane 0 of red_out_d): one per
// 256-particle group — or, for a specialised kernel running PP particles per thread, one per
// workgroup (OP_REDMAX only: a max is the same under any grouping).  Ask AFTER specialising.
extern "C" int64_t gmx_program_grid(const gmx_program* p, int64_t n) {
  int64_t per = GMX_BLOCK;
  if (p && p->jit_fn && !p->uses_lse) per = (int64_t)GMX_BLOCK * p->jit_pp;
  return (n + per - 1) / per;
}

static int peer_check(const char* who, const gmx_peer& P, int64_t n_per_rank);
extern "C" int gmx_program_run(const gmx_program* p, int64_t n, const gmx_run_args* args,
                               gmx_stream stream) {
  if (!p || !args) return gmx_fail("gmx_program_run: null argument%s");
  if (n < 0) return gmx_fail("gmx_program_run: negative n%s");
  if (n == 0) return 0;
  if (n > (int64_t)0x7fffffff * GMX_BLOCK) return gmx_fail("gmx_program_run: n too large%s");
  // host-side shape checks: every slot the program names must be bound
  for (uint32_t s = 0; s < p->n_in; ++s)
    if (!args->in_d[s]) return gmx_fail("gmx_program_run: input slot %s%lld is null", "", s);
  for (uint32_t s = 0; s < p->n_out; ++s)
    if (!args->out_d[s]) return gmx_fail("gmx_program_run: output slot %s%lld is null", "", s);
  for (uint32_t s = 0; s < p->n_tab; ++s)
    if (!args->tab_d[s]) return gmx_fail("gmx_program_run: table slot %s%lld is null", "", s);
  if (p->uses_gather && !args->ancestors_d)
    return gmx_fail("gmx_program_run: program gathers but ancestors_d is null%s");
  const bool fused_rs = args->rs.lw_d != nullptr;
  if (fused_rs) {
    const gmx_resample_in& q = args->rs;
    if (!gmx_program_fuses_resample(p))
      return gmx_fail("gmx_program_run: rs is set but this program cannot resample in its own launch "
                      "(gmx_program_set_fuse_resample before specialising; 4 particles per thread; gathering)%s");
    if (!q.tile_max_d || !q.tile_agg_d || !q.max_out_d || !q.total_out_d || !q.status_d)
      return gmx_fail("gmx_program_run: rs has a null pointer%s");
    if ((uintptr_t)q.lw_d & 15) return gmx_fail("gmx_program_run: rs.lw_d must be 16-byte aligned%s");
    if ((!p->fuse_rs_loop && (n + RS_TILE - 1) / RS_TILE > 1024) || n > gmx_program_resident_particles(p) ||
        n > (int64_t)GMX_ANC_INDEX_MASK + 1)
      return gmx_fail("gmx_program_run: rs: every workgroup of the launch must be resident at once (n <= 2^20 and "
                      "n <= gmx_program_resident_particles(); a looped program — gmx_program_set_fuse_resample_loop — "
                      "walks up to 16 tiles per workgroup: n <= 2^24)%s");
    if (q.shift < 1 || q.shift > 62) return gmx_fail("gmx_program_run: rs.shift out of range%s");
    int need = 0;
    while (((int64_t)1 << need) < n) ++need;
    if (q.shift + need > 62) return gmx_fail("gmx_program_run: rs.shift too large for n (overflow)%s");
    if (q.tag < 1u || q.tag > GMX_ANC_TAG_MAX) return gmx_fail("gmx_program_run: rs.tag must be in [1, 255]%s");
    if (args->tile_agg_d == q.tile_agg_d || (const float*)args->red_out_d == q.tile_max_d)
      return gmx_fail("gmx_program_run: rs reads the tile statistics this launch writes (use two sets)%s");
    for (uint32_t s = 0; s < p->n_out; ++s)
      if ((const void*)args->out_d[s] == (const void*)q.lw_d)
        return gmx_fail("gmx_program_run: rs.lw_d is also an output of this launch (use two buffers)%s");
  }
  const bool fused_sh = args->sh.lw_d != nullptr;
  if (fused_sh) {
    const gmx_shard_in& q = args->sh;
    if (fused_rs) return gmx_fail("gmx_program_run: rs and sh are both set%s");
    if (!gmx_program_fuses_shard_step(p))
      return gmx_fail("gmx_program_run: sh is set but this program cannot route a sharded step in its own launch "
                      "(gmx_program_set_fuse_shard_step before specialising; 4 particles per thread; gathering)%s");
    if (!q.stats_own_d || !q.plan_d || !q.total_out_d || !q.max_out_d || !q.status_d)
      return gmx_fail("gmx_program_run: sh has a null pointer%s");
    if (((uintptr_t)q.lw_d & 15) || ((uintptr_t)q.stats_own_d & 7))
      return gmx_fail("gmx_program_run: sh.lw_d must be 16-byte and sh.stats_own_d 8-byte aligned%s");
    if (peer_check("gmx_program_run (sh)", q.peer, n)) return 1;
    if (q.peer.world > 8 || (int64_t)q.peer.world * q.peer.tiles > 4 * GMX_BLOCK)
      return gmx_fail("gmx_program_run: sh: world <= 8 and world * tiles <= 1024%s");
    if (n + (int64_t)q.peer.world * q.peer.capacity > (int64_t)GMX_ANC_INDEX_MASK + 1)
      return gmx_fail("gmx_program_run: sh: n + world * capacity must fit the 21-bit index of an ancestor word%s");
    if (n > gmx_program_resident_particles(p))
      return gmx_fail("gmx_program_run: sh: every workgroup of the launch must be resident at once "
                      "(n <= gmx_program_resident_particles())%s");
    if (q.shift < 1 || q.shift > 62) return gmx_fail("gmx_program_run: sh.shift out of range%s");
    if (q.tag < 1u || q.tag > GMX_ANC_TAG_MAX) return gmx_fail("gmx_program_run: sh.tag must be in [1, 255]%s");
    for (int l = 0; l < q.peer.leaves; ++l)
      if (!q.state_d[l] || !q.tail_d[l]) return gmx_fail("gmx_program_run: sh: a leaf pointer is null%s");
    for (uint32_t s_ = 0; s_ < p->n_out; ++s_)
      if ((const void*)args->out_d[s_] == (const void*)q.lw_d)
        return gmx_fail("gmx_program_run: sh.lw_d is also an output of this launch (use two buffers)%s");
  }
  if (p->uses_red && !args->red_out_d)
    return gmx_fail("gmx_program_run: program reduces but red_out_d is null%s");
  if (args->tile_agg_d) {
    if (!gmx_program_writes_tile_stats(p))
      return gmx_fail("gmx_program_run: tile_agg_d is set but this program cannot write tile statistics "
                      "(gmx_program_writes_tile_stats)%s");
    if (args->tile_shift < 1 || args->tile_shift > 62) return gmx_fail("gmx_program_run: tile_shift out of range%s");
  }
  if (p->uses_step && args->step_stride < n)
    return gmx_fail("gmx_program_run: step_stride must be at least n for a program with step-indexed leaves%s");
  if (p->uses_key) {
    int km = args->key_mode;
    if (km != GMX_KEY_ARRAY && km != GMX_KEY_SPLIT && km != GMX_KEY_ROWSPLIT && km != GMX_KEY_BCAST)
      return gmx_fail("gmx_program_run: program draws but key_mode is unset%s");
    if ((km == GMX_KEY_ARRAY || km == GMX_KEY_ROWSPLIT) && !args->keys_d)
      return gmx_fail("gmx_program_run: keys_d is null%s");
    if (km == GMX_KEY_ROWSPLIT && args->key_inner <= 0)
      return gmx_fail("gmx_program_run: key_inner must be positive%s");
  }
  dim3 grid((unsigned)((n + GMX_BLOCK - 1) / GMX_BLOCK)), block(GMX_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  // the program's constants live in the operand pool after the launch uniforms
  gmx_run_args patched;
  if (p->n_const || fused_rs || fused_sh) {
    patched = *args;
    for (uint32_t k = 0; k < p->n_const; ++k) patched.uni[p->n_dyn + k] = p->consts[k];
    if (fused_rs) {
      uint32_t b0, b1;
      gmx_threefry2x32(patched.rs.key0, patched.rs.key1, 0u, 0u, &b0, &b1);     // bits32(key, 0) on the host
      patched.rs.u0 = (b0 ^ b1) >> 9;
    }
    if (fused_sh) {
      uint32_t b0, b1;
      gmx_threefry2x32(patched.sh.key0, patched.sh.key1, 0u, 0u, &b0, &b1);
      patched.sh.u0 = (b0 ^ b1) >> 9;
    }
    args = &patched;
  }
  if (p->jit_fn) {
    if (n >= 0x7fffffffLL - 4 * GMX_BLOCK)
      return gmx_fail("gmx_program_run: a specialised kernel indexes particles with 32 bits (n < 2^31 - 1024)%s");
    struct { int64_t n; gmx_run_args A; } ka;
    ka.n = n; ka.A = *args;
    size_t ka_size = sizeof(ka);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ka_size,
                      HIP_LAUNCH_PARAM_END};
    // A background program over several ROWS of keys (GMX_KEY_ROWSPLIT: the draws of several SMC steps in one
    // launch) runs as a 2-D grid, one row per blockIdx.y: no division to find a particle's row, and one node
    // instead of `rows` in a captured graph.  Only programs that read nothing and reduce nothing.
    int64_t rows = 1, per_row = n;
    if (p->background && args->key_mode == GMX_KEY_ROWSPLIT && args->key_inner > 0 && n % args->key_inner == 0 &&
        n / args->key_inner > 1 && n / args->key_inner <= 65535 && p->n_in == 0 && !p->uses_red && !p->uses_step) {
      rows = n / args->key_inner;
      per_row = args->key_inner;
      ka.n = per_row;
    }
    unsigned jgrid = (unsigned)((per_row + (int64_t)GMX_BLOCK * p->jit_pp - 1) / ((int64_t)GMX_BLOCK * p->jit_pp));
    if (p->fuse_rs_loop && p->jit_pp == 4) jgrid = (unsigned)rs_loop_grid(p, (int64_t)jgrid);      // each workgroup walks its tiles
    const unsigned dyn_lds = p->lds_pad;
    GMX_HIP(hipModuleLaunchKernel(p->jit_fn, jgrid, (unsigned)rows, 1, GMX_BLOCK, 1, 1, dyn_lds, st, nullptr, config));
    return 0;
  }
  // The interpreter's register file is a 16- or 32-element vector indexed at run time, and the
  // two-register write (dst, dst + 1) is emitted for every instruction: the LAST element must stay
  // unused, or the (predicated-off) dst + 1 access leaves the vector — measured on gfx950: a
  // program writing r15 of the 16-element file stored nothing at all.  So 15 / 31 usable registers.
  if (p->n_regs > 31)
    return gmx_fail("gmx_program_run: a program with more than 31 live values runs only as a specialised kernel "
                    "(gmx_program_specialize; hiprtc unavailable or GENMI_JIT=0?)%s");
  if (p->n_regs <= 15) {
    if (p->needs_full)
      hipLaunchKernelGGL((k_vm<gmx_regs_vgpr<16>, true>), grid, block, 0, st, p->code_d, p->n_instr, n, *args);
    else
      hipLaunchKernelGGL(k_vm_lean, grid, block, 0, st, p->code_d, p->n_instr, n, *args);
  } else {
    if (p->needs_full)
      hipLaunchKernelGGL((k_vm<gmx_regs_vgpr<32>, true>), grid, block, 0, st, p->code_d, p->n_instr, n, *args);
    else
      hipLaunchKernelGGL((k_vm<gmx_regs_vgpr<32>, false>), grid, block, 0, st, p->code_d, p->n_instr, n, *args);
  }
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// key kernels
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GMX_BLOCK)
k_split(uint32_t k0, uint32_t k1, int64_t n, int64_t off, uint32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (i >= n) return;
  gmx_key k; k.k0 = k0; k.k1 = k1;
  gmx_key o = gmx_split_child(k, (uint64_t)(off + i));
  reinterpret_cast<uint2*>(out)[i] = make_uint2(o.k0, o.k1);
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_split_rows(const uint32_t* __restrict__ keys, int64_t rows, int64_t inner,
             uint32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (i >= rows * inner) return;
  int64_t r = i / inner, j = i - r * inner;
  uint2 kk = reinterpret_cast<const uint2*>(keys)[r];
  gmx_key k; k.k0 = kk.x; k.k1 = kk.y;
  gmx_key o = gmx_split_child(k, (uint64_t)j);
  reinterpret_cast<uint2*>(out)[i] = make_uint2(o.k0, o.k1);
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_fold_in(const uint32_t* __restrict__ keys, uint32_t data, int64_t n, uint32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint2 kk = reinterpret_cast<const uint2*>(keys)[i];
  gmx_key k; k.k0 = kk.x; k.k1 = kk.y;
  gmx_key o = gmx_fold_in(k, data);
  reinterpret_cast<uint2*>(out)[i] = make_uint2(o.k0, o.k1);
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_random_bits(const uint32_t* __restrict__ keys, int64_t n, int64_t m, uint32_t* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (t >= n * m) return;
  int64_t i = t / m, j = t - i * m;
  uint2 kk = reinterpret_cast<const uint2*>(keys)[i];
  gmx_key k; k.k0 = kk.x; k.k1 = kk.y;
  out[t] = gmx_bits32(k, (uint64_t)j);
}

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + GMX_BLOCK - 1) / GMX_BLOCK)); }

extern "C" int gmx_split(const uint32_t key[2], int64_t n, int64_t index_offset,
                         uint32_t* out_keys_d, gmx_stream stream) {
  if (!key || (n > 0 && !out_keys_d)) return gmx_fail("gmx_split: null argument%s");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_split, grid_for(n), dim3(GMX_BLOCK), 0, (hipStream_t)stream, key[0], key[1],
                     n, index_offset, out_keys_d);
  GMX_HIP(hipGetLastError());
  return 0;
}
extern "C" int gmx_split_rows(const uint32_t* keys_d, int64_t rows, int64_t inner,
                              uint32_t* out_keys_d, gmx_stream stream) {
  if (rows <= 0 || inner <= 0) return 0;
  if (!keys_d || !out_keys_d) return gmx_fail("gmx_split_rows: null argument%s");
  hipLaunchKernelGGL(k_split_rows, grid_for(rows * inner), dim3(GMX_BLOCK), 0, (hipStream_t)stream,
                     keys_d, rows, inner, out_keys_d);
  GMX_HIP(hipGetLastError());
  return 0;
}
extern "C" int gmx_fold_in(const uint32_t* keys_d, uint32_t data, int64_t n, uint32_t* out_d,
                           gmx_stream stream) {
  if (n <= 0) return 0;
  if (!keys_d || !out_d) return gmx_fail("gmx_fold_in: null argument%s");
  hipLaunchKernelGGL(k_fold_in, grid_for(n), dim3(GMX_BLOCK), 0, (hipStream_t)stream, keys_d, data, n,
                     out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}
extern "C" int gmx_random_bits(const uint32_t* keys_d, int64_t n, int64_t m, uint32_t* out_d,
                               gmx_stream stream) {
  if (n <= 0 || m <= 0) return 0;
  if (!keys_d || !out_d) return gmx_fail("gmx_random_bits: null argument%s");
  hipLaunchKernelGGL(k_random_bits, grid_for(n * m), dim3(GMX_BLOCK), 0, (hipStream_t)stream, keys_d,
                     n, m, out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// log-sum-exp
// ---------------------------------------------------------------------------
#define LSE_ITEMS 16
#define LSE_TILE (GMX_BLOCK * LSE_ITEMS)

// stage 1: one block per (row, tile) -> partial (max, sum exp(x - max))
__global__ void __launch_bounds__(GMX_BLOCK)
k_lse_tiles(const float* __restrict__ lw, int64_t cols, int64_t tiles_per_row,
            float* __restrict__ partials) {
  __shared__ float lds4[4];
  int64_t row = blockIdx.y, tile = blockIdx.x;
  const float* x = lw + row * cols;
  int64_t base = tile * LSE_TILE;
  float v[LSE_ITEMS];
  float m = -gmx_inf();
#pragma unroll
  for (int k = 0; k < LSE_ITEMS; ++k) {
    int64_t j = base + (int64_t)k * GMX_BLOCK + threadIdx.x;   // coalesced per k
    v[k] = (j < cols) ? x[j] : -gmx_inf();
    m = gmx_rmax(m, v[k]);
  }
  m = block_max(m, lds4);
  float s = 0.0f;
  if (m > -gmx_inf()) {
#pragma unroll
    for (int k = 0; k < LSE_ITEMS; ++k) s += gmx_expf(v[k] - m);
  }
  s = block_sum(s, lds4);
  if (threadIdx.x == 0) {
    float* p = partials + 2 * (row * tiles_per_row + tile);
    p[0] = m; p[1] = s;
  }
}
// stage 2: one block per row combines partials in a fixed order
__global__ void __launch_bounds__(GMX_BLOCK)
k_lse_final(const float* __restrict__ partials, int64_t n_part, float* __restrict__ out,
            float* __restrict__ out_max) {
  __shared__ float lds4[4];
  int64_t row = blockIdx.x;
  const float* p = partials + 2 * row * n_part;
  float m = -gmx_inf();
  for (int64_t j = threadIdx.x; j < n_part; j += GMX_BLOCK) m = gmx_rmax(m, p[2 * j]);
  m = block_max(m, lds4);
  float s = 0.0f;
  if (m > -gmx_inf())
    for (int64_t j = threadIdx.x; j < n_part; j += GMX_BLOCK)
      s += p[2 * j + 1] * gmx_expf(p[2 * j] - m);
  s = block_sum(s, lds4);
  if (threadIdx.x == 0) {
    out[row] = (m > -gmx_inf()) ? m + gmx_logf(s) : m;
    if (out_max) out_max[row] = m;
  }
}
// many short rows: one wave per row
__global__ void __launch_bounds__(GMX_BLOCK)
k_lse_rows(const float* __restrict__ lw, int64_t rows, int64_t cols, float* __restrict__ out,
           float* __restrict__ out_max) {
  int64_t row = (int64_t)blockIdx.x * (GMX_BLOCK / GMX_WAVE) + (threadIdx.x >> 6);
  if (row >= rows) return;
  int lane = threadIdx.x & 63;
  const float* x = lw + row * cols;
  float m = -gmx_inf();
  for (int64_t j = lane; j < cols; j += GMX_WAVE) m = gmx_rmax(m, x[j]);
  m = wave_max(m);
  float s = 0.0f;
  if (m > -gmx_inf())
    for (int64_t j = lane; j < cols; j += GMX_WAVE) s += gmx_expf(x[j] - m);
  s = wave_sum(s);
  if (lane == 0) {
    out[row] = (m > -gmx_inf()) ? m + gmx_logf(s) : m;
    if (out_max) out_max[row] = m;
  }
}

#define LSE_ROWS_PATH_MAX_COLS 4096
static inline bool lse_use_rows_path(int64_t rows, int64_t cols) {
  return rows >= 32 && cols <= LSE_ROWS_PATH_MAX_COLS;
}
extern "C" size_t gmx_logsumexp_workspace(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0 || lse_use_rows_path(rows, cols)) return 16;
  int64_t tiles = (cols + LSE_TILE - 1) / LSE_TILE;
  return (size_t)(rows * tiles * 2 * sizeof(float)) + 16;
}
extern "C" int gmx_logsumexp(const float* lw_d, int64_t rows, int64_t cols, float* out_d,
                             float* out_max_d, void* workspace_d, gmx_stream stream) {
  if (rows <= 0) return 0;
  if (cols <= 0) return gmx_fail("gmx_logsumexp: cols must be positive%s");
  if (!lw_d || !out_d) return gmx_fail("gmx_logsumexp: null argument%s");
  hipStream_t st = (hipStream_t)stream;
  if (lse_use_rows_path(rows, cols)) {
    int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(k_lse_rows, dim3((unsigned)blocks), dim3(GMX_BLOCK), 0, st, lw_d, rows, cols,
                       out_d, out_max_d);
  } else {
    if (!workspace_d) return gmx_fail("gmx_logsumexp: workspace required%s");
    if (rows > 65535) return gmx_fail("gmx_logsumexp: too many long rows%s");
    int64_t tiles = (cols + LSE_TILE - 1) / LSE_TILE;
    float* part = (float*)workspace_d;
    hipLaunchKernelGGL(k_lse_tiles, dim3((unsigned)tiles, (unsigned)rows), dim3(GMX_BLOCK), 0, st,
                       lw_d, cols, tiles, part);
    hipLaunchKernelGGL(k_lse_final, dim3((unsigned)rows), dim3(GMX_BLOCK), 0, st, part, tiles, out_d,
                       out_max_d);
  }
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// row sums in a FIXED tree (include/genmi.h: gmx_sum_rows): the plate score of a Vmap whose elements run on the
// launch axis.  Stage 1, one block per (row, tile of 4096): thread t adds its 16 items t, t + 256, ... in that
// order, the wave butterfly (xor 32, 16, ..., 1) and (w0 + w1) + (w2 + w3) give the tile's partial; stage 2, one
// block per row: thread t adds partials t, t + 256, ... in order, then the same block tree.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GMX_BLOCK)
k_sum_tiles(const float* __restrict__ x, int64_t cols, int64_t tiles_per_row, float* __restrict__ partials) {
  __shared__ float lds4[4];
  const int64_t row = blockIdx.y, tile = blockIdx.x;
  const float* xr = x + row * cols;
  const int64_t base = tile * LSE_TILE;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < LSE_ITEMS; ++k) {
    const int64_t j = base + (int64_t)k * GMX_BLOCK + threadIdx.x;
    s += (j < cols) ? xr[j] : 0.0f;
  }
  s = block_sum(s, lds4);
  if (threadIdx.x == 0) partials[row * tiles_per_row + tile] = s;
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_sum_final(const float* __restrict__ partials, int64_t n_part, float* __restrict__ out) {
  __shared__ float lds4[4];
  const int64_t row = blockIdx.x;
  const float* p = partials + row * n_part;
  float s = 0.0f;
  for (int64_t j = threadIdx.x; j < n_part; j += GMX_BLOCK) s += p[j];
  s = block_sum(s, lds4);
  if (threadIdx.x == 0) out[row] = s;
}
// out[r] = ((x[r, 0] + x[r, 1]) + x[r, 2]) + ... in ELEMENT ORDER (one thread per row): the score of a plate held per
// particle (the counted loop's own order), recomputed after one of its elements changed
__global__ void __launch_bounds__(GMX_BLOCK)
k_sum_rows_inorder(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t sr, int64_t sc, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (r >= rows) return;
  const float* p = x + r * sr;
  float acc = 0.0f;
  for (int64_t c = 0; c < cols; ++c) acc += p[c * sc];
  out[r] = acc;
}
// the same sum for LONG contiguous rows (a plate of 1e4..1e6 elements held per particle): one wave per row.  The chain of
// dependent adds is the floor (cols x one v_add_f32), so the job of the other 63 lanes is to keep it fed: they fetch the
// next INORDER_CHUNK elements with coalesced loads while every lane adds the current chunk out of LDS (all lanes read the
// same address, a broadcast, and hold the same accumulator; nothing diverges)
constexpr int INORDER_CHUNK = 1024;
constexpr int INORDER_LONG_MIN = 1024;
__global__ void __launch_bounds__(64)
k_sum_rows_inorder_long(const float* __restrict__ x, int64_t cols, int64_t sr, float* __restrict__ out) {
  __shared__ float4 buf[2][INORDER_CHUNK / 4];
  const float* p = x + (int64_t)blockIdx.x * sr;
  const int lane = threadIdx.x;
  const int64_t chunks = (cols + INORDER_CHUNK - 1) / INORDER_CHUNK;
  float v[INORDER_CHUNK / 64];
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < INORDER_CHUNK / 64; ++k) v[k] = (k * 64 + lane < cols) ? p[k * 64 + lane] : 0.0f;
  for (int64_t c = 0; c < chunks; ++c) {
    float* b = (float*)buf[c & 1];
#pragma unroll
    for (int k = 0; k < INORDER_CHUNK / 64; ++k) b[k * 64 + lane] = v[k];
    __syncthreads();
    if (c + 1 < chunks) {
      const int64_t base = (c + 1) * INORDER_CHUNK;
#pragma unroll
      for (int k = 0; k < INORDER_CHUNK / 64; ++k) v[k] = (base + k * 64 + lane < cols) ? p[base + k * 64 + lane] : 0.0f;
    }
    const int64_t left = cols - c * INORDER_CHUNK;
    if (left >= INORDER_CHUNK) {
      const float4* b4 = buf[c & 1];
#pragma unroll 8
      for (int i = 0; i < INORDER_CHUNK / 4; ++i) {
        const float4 t = b4[i];
        acc += t.x;
        acc += t.y;
        acc += t.z;
        acc += t.w;
      }
    } else {
      for (int i = 0; i < (int)left; ++i) acc += b[i];
    }
  }
  if (lane == 0) out[blockIdx.x] = acc;
}
extern "C" int gmx_sum_rows_inorder(const float* x_d, int64_t rows, int64_t cols, int64_t stride_row, int64_t stride_col,
                                    float* out_d, gmx_stream stream) {
  if (rows <= 0) return 0;
  if (cols < 0 || !x_d || !out_d) return gmx_fail("gmx_sum_rows_inorder: bad argument%s");
  if (stride_col == 1 && cols >= INORDER_LONG_MIN && rows <= 0x7fffffff) {
    hipLaunchKernelGGL(k_sum_rows_inorder_long, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, x_d, cols, stride_row,
                       out_d);
    GMX_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_sum_rows_inorder, grid_for(rows), dim3(GMX_BLOCK), 0, (hipStream_t)stream, x_d, rows, cols, stride_row,
                     stride_col, out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}
extern "C" size_t gmx_sum_rows_workspace(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 16;
  return (size_t)(rows * ((cols + LSE_TILE - 1) / LSE_TILE) * sizeof(float)) + 16;
}
extern "C" int gmx_sum_rows(const float* x_d, int64_t rows, int64_t cols, float* out_d, void* workspace_d,
                            gmx_stream stream) {
  if (rows <= 0) return 0;
  if (cols <= 0) return gmx_fail("gmx_sum_rows: cols must be positive%s");
  if (!x_d || !out_d || !workspace_d) return gmx_fail("gmx_sum_rows: null argument%s");
  if (rows > 65535) return gmx_fail("gmx_sum_rows: too many rows%s");
  const int64_t tiles = (cols + LSE_TILE - 1) / LSE_TILE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sum_tiles, dim3((unsigned)tiles, (unsigned)rows), dim3(GMX_BLOCK), 0, st, x_d, cols, tiles,
                     (float*)workspace_d);
  hipLaunchKernelGGL(k_sum_final, dim3((unsigned)rows), dim3(GMX_BLOCK), 0, st, (const float*)workspace_d, tiles, out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// fixed-point weights + chained inclusive scan (single pass, decoupled look-back)
//
// Tile = 4096 log-weights (256 threads x 4 sub-tiles x float4), so 1e6
// particles are 245 tiles: every tile's look-back window (256 predecessors,
// 4 descriptors per lane of wave 0, all loads in flight together) covers the
// whole prefix in ONE polling round — the scan is bounded by one HBM round
// trip for the loads plus one L2 round trip for the look-back, not by a
// serial chain.  The global max comes from the per-block partials a site
// program's OP_REDMAX left behind (no extra launch), and the workspace is
// returned zeroed by the last tile to finish (no memset launch).
// ---------------------------------------------------------------------------
#ifndef CDF_THREADS
#define CDF_THREADS 1024               /* 16 waves: 4 per SIMD, enough to hide the load + look-back latency */
#endif
#define CDF_WAVES (CDF_THREADS / GMX_WAVE)
#ifndef CDF_VEC
#define CDF_VEC 4                      /* consecutive items per thread (one float4) */
#endif
#define CDF_TILE (CDF_THREADS * CDF_VEC)
static_assert(CDF_THREADS % 256 == 0 && CDF_VEC * 256 == 1024, "a definition tile of the CDF is 1024 items = 256 threads");
#define CDF_ST_AGG 1ull
#define CDF_ST_INC 2ull
#define CDF_SPIN_LIMIT (1u << 22)
#define CDF_LOOK 4                     /* descriptors per lane per look-back round (window 256) */
#define CDF_RESIDENT_TILES 512         /* grids up to this size are co-resident: blockIdx is the tile id */

// workspace: [0] u32 ticket, [1] u32 error flag, [2] u32 done count, [3] pad,
// then one u64 descriptor per tile: value << 2 | status.  value < 2^62 by the
// choice of `shift`, so a descriptor is a single naturally aligned 8-byte
// granule: written and read with relaxed agent-scope atomics it needs no
// ordering against any other memory.  Contract: zero on entry, zero on exit.
struct cdf_ws { uint32_t ticket; uint32_t error; uint32_t done; uint32_t pad; uint64_t desc[1]; };

extern "C" size_t gmx_weight_cdf_workspace(int64_t n) {
  int64_t tiles = (n + CDF_TILE - 1) / CDF_TILE;
  if (tiles < 1) tiles = 1;
  return 16 + (size_t)tiles * 8;
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_reduce_max(const float* __restrict__ partials, int64_t n_part, float* __restrict__ max_out) {
  __shared__ float lds4[4];
  float m = -gmx_inf();
  for (int64_t j = threadIdx.x; j < n_part; j += GMX_BLOCK) m = gmx_rmax(m, partials[j]);
  m = block_max(m, lds4);
  if (threadIdx.x == 0) *max_out = m;
}

// max_mode: 0 = read *max_d; 1 = reduce column 0 of partials[n_part][2] (every tile
// does it: a few KB from L2) and tile 0 stores the result to *max_d.
__global__ void __launch_bounds__(CDF_THREADS)
k_weight_cdf(const float* __restrict__ lw, int64_t n, float scale, int max_mode,
             const float* __restrict__ partials, int64_t n_part, float* __restrict__ max_d,
             uint64_t* __restrict__ cdf, uint64_t* __restrict__ total_out, cdf_ws* ws) {
  __shared__ uint64_t s_part[CDF_WAVES];
  __shared__ uint64_t s_prefix;
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_last;
  __shared__ float s_max[CDF_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = (n + CDF_TILE - 1) / CDF_TILE;
  uint32_t tile = blockIdx.x;
  if (n_tiles > CDF_RESIDENT_TILES) {
    // dynamic tile id: a tile only ever waits on tiles whose blocks already run
    if (threadIdx.x == 0) s_tile = atomicAdd(&ws->ticket, 1u);
    __syncthreads();
    tile = s_tile;
  }
  if ((int64_t)tile >= n_tiles) return;
  const int64_t base = (int64_t)tile * CDF_TILE + (int64_t)threadIdx.x * CDF_VEC;
  // issue the tile's load first, reduce the max while it is in flight
  float x[CDF_VEC];
  if (base + CDF_VEC <= n) {
    float4 v = *reinterpret_cast<const float4*>(lw + base);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) x[c] = (base + c < n) ? lw[base + c] : -gmx_inf();
  }
  float M;
  if (max_mode == 1) {
    float m = -gmx_inf();
    for (int64_t j = threadIdx.x; j < n_part; j += CDF_THREADS) m = gmx_rmax(m, partials[j]);
    m = wave_max(m);
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int w = 1; w < CDF_WAVES; ++w) m = gmx_rmax(m, s_max[w]);
    M = m;
    if (tile == 0 && threadIdx.x == 0) *max_d = M;
  } else {
    M = *max_d;
  }
  // ---- the two-level CDF: every 256 threads (4 waves, 1024 items) are one definition tile ----
  constexpr int NG = CDF_WAVES / 4;
  const int grp = wave >> 2;
  float gm = -gmx_inf();
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) gm = gmx_rmax(gm, x[c]);
  gm = wave_max(gm);
  __syncthreads();                       // s_max may still be read by the max_mode == 1 reduction above
  if (lane == 0) s_max[wave] = gm;
  __syncthreads();
  const float m_b = gmx_rmax(gmx_rmax(s_max[4 * grp], s_max[4 * grp + 1]), gmx_rmax(s_max[4 * grp + 2], s_max[4 * grp + 3]));
  const int32_t K = gmx_tile_exp(M);
  const int32_t k_b = gmx_tile_exp(m_b);
  const float ref_b = gmx_tile_ref(k_b);
  uint64_t q[CDF_VEC];
  uint64_t run = 0;
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) {
    uint64_t w = (base + c < n) ? weight_fixed(x[c], ref_b, scale) : 0ull;
    run += w;
    q[c] = run;                        // thread-local inclusive
  }
  const uint64_t inc = wave_scan_u64(run);   // wave inclusive scan of thread totals (DPP)
  if (lane == 63) s_part[wave] = inc;
  __syncthreads();
  // per definition tile g: A_g (sum of its 4 waves) and G_g = A_g * 2^(k_g - K)
  uint64_t tile_agg = 0, grp_off = 0, wave_off = 0;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    uint64_t A = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (g == grp && 4 * g + w == wave) wave_off = A;
      A += s_part[4 * g + w];
    }
    const float mg = gmx_rmax(gmx_rmax(s_max[4 * g], s_max[4 * g + 1]), gmx_rmax(s_max[4 * g + 2], s_max[4 * g + 3]));
    if (g == grp) grp_off = tile_agg;
    tile_agg += gmx_tile_scale(A, gmx_tile_exp(mg), K);
  }
  // ---- chained scan across tiles (wave 0) ----
  if (wave == 0) {
    uint64_t prefix = 0;
    if (tile == 0) {
      if (lane == 0)
        __hip_atomic_store(&ws->desc[0], (tile_agg << 2) | CDF_ST_INC, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(&ws->desc[tile], (tile_agg << 2) | CDF_ST_AGG, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      int64_t look = (int64_t)tile - 1;     // nearest predecessor of the current window
      uint32_t spins = 0;
      bool done = false;
      while (!done) {
        // one round: CDF_LOOK x 64 predecessors, all loads in flight together
        uint64_t d[CDF_LOOK];
#pragma unroll
        for (int r = 0; r < CDF_LOOK; ++r) {
          int64_t t = look - (int64_t)r * GMX_WAVE - lane;
          d[r] = (t >= 0) ? __hip_atomic_load(&ws->desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : CDF_ST_INC;       // virtual tile -1: inclusive prefix 0
        }
        bool retry = false;
        uint64_t acc = 0;
        int consumed = 0;
#pragma unroll
        for (int r = 0; r < CDF_LOOK; ++r) {
          if (done || retry) continue;
          uint64_t st = d[r] & 3ull;
          unsigned long long inc_mask = __ballot(st == CDF_ST_INC);
          unsigned long long none_mask = __ballot(st == 0ull);
          int first_inc = inc_mask ? __builtin_ctzll(inc_mask) : 64;
          unsigned long long need = (first_inc >= 63) ? ~0ull : ((2ull << first_inc) - 1ull);
          if (none_mask & need) { retry = true; continue; }
          uint64_t contrib = (lane <= first_inc) ? (d[r] >> 2) : 0ull;
          acc += wave_sum_u64(contrib);
          consumed = r + 1;
          if (first_inc < 64) done = true;
        }
        prefix += acc;                        // sub-windows consumed so far stay valid
        look -= (int64_t)consumed * GMX_WAVE;
        if (retry) {
          if (++spins > CDF_SPIN_LIMIT) { if (lane == 0) ws->error = 1u; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (lane == 0)
        __hip_atomic_store(&ws->desc[tile], ((prefix + tile_agg) << 2) | CDF_ST_INC,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  const uint64_t tile_prefix = s_prefix;
  const uint64_t off = tile_prefix + grp_off;              // mass before this definition tile
  const uint64_t loc = wave_off + (inc - run);             // tile-local mass before this thread
  uint64_t cv[CDF_VEC];
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) cv[c] = off + gmx_tile_scale(loc + q[c], k_b, K);
  if (base + CDF_VEC <= n) {
    ulonglong2 a, b;
    a.x = cv[0]; a.y = cv[1]; b.x = cv[2]; b.y = cv[3];
    reinterpret_cast<ulonglong2*>(cdf + base)[0] = a;
    reinterpret_cast<ulonglong2*>(cdf + base)[1] = b;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c)
      if (base + c < n) cdf[base + c] = cv[c];
  }
  if ((int64_t)tile == n_tiles - 1 && threadIdx.x == 0) *total_out = tile_prefix + tile_agg;
  // ---- leave the workspace zeroed: the last tile to finish cleans up ----
  if (threadIdx.x == 0) s_last = (atomicAdd(&ws->done, 1u) == (uint32_t)(n_tiles - 1)) ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    for (int64_t t = threadIdx.x; t < n_tiles; t += CDF_THREADS) ws->desc[t] = 0ull;
    if (threadIdx.x == 0) { ws->ticket = 0u; ws->done = 0u; }
  }
}

extern "C" int gmx_reduce_max(const float* partials_d, int64_t n, float* max_d, gmx_stream stream) {
  if (n <= 0) return gmx_fail("gmx_reduce_max: n must be positive%s");
  if (!partials_d || !max_d) return gmx_fail("gmx_reduce_max: null argument%s");
  hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(GMX_BLOCK), 0, (hipStream_t)stream, partials_d, n, max_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_weight_cdf(const float* lw_d, int64_t n, int shift,
                              const float* max_partials_d, int64_t n_partials, float* max_d,
                              uint64_t* cdf_d, uint64_t* total_d, void* workspace_d,
                              gmx_stream stream) {
  if (n <= 0) return gmx_fail("gmx_weight_cdf: n must be positive%s");
  if (!lw_d || !max_d || !cdf_d || !total_d || !workspace_d)
    return gmx_fail("gmx_weight_cdf: null argument%s");
  if (shift < 1 || shift > 62) return gmx_fail("gmx_weight_cdf: shift out of range%s");
  // a sum of n terms each <= 2^shift must stay below 2^62
  int need = 0;
  while (((int64_t)1 << need) < n) ++need;
  if (shift + need > 62) return gmx_fail("gmx_weight_cdf: shift too large for n (overflow)%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)cdf_d & 15))
    return gmx_fail("gmx_weight_cdf: lw_d and cdf_d must be 16-byte aligned%s");
  hipStream_t st = (hipStream_t)stream;
  int max_mode = 0;
  if (max_partials_d) {
    if (n_partials <= 0) return gmx_fail("gmx_weight_cdf: n_partials must be positive%s");
    if (n_partials <= 16384) {
      max_mode = 1;               // every tile reduces the partials itself: no extra launch
    } else {
      hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(GMX_BLOCK), 0, st, max_partials_d, n_partials, max_d);
    }
  }
  int64_t tiles = (n + CDF_TILE - 1) / CDF_TILE;
  float scale = gmx_pow2i(shift);
  hipLaunchKernelGGL(k_weight_cdf, dim3((unsigned)tiles), dim3(CDF_THREADS), 0, st, lw_d, n, scale,
                     max_mode, max_partials_d, n_partials, max_d, cdf_d, total_d, (cdf_ws*)workspace_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// ancestors
//
// ancestor(j) = first i with cdf_i * D > P_j (128-bit integers), where for
//   systematic / stratified  D = n_out * 2^23,  P_j = (j * 2^23 + u_j) * total
//   multinomial              D = 2^23,          P_j = total * (2^23 - u_j) - 1
//
// systematic / stratified: P_j is increasing in j, so source i owns the slot
// range [f(cdf_{i-1}), f(cdf_i)) with f(c) = #{ j : P_j < c * D }.
// k_offspring is source-centric: it reads the CDF once, coalesced, evaluates f
// from an f64 estimate corrected with the exact 128-bit predicate, and writes
// each ancestor index once — no dependent-load search chain.
// multinomial positions are not ordered: k_ancestors binary-searches per slot.
// ---------------------------------------------------------------------------
// (u128 / slot_threshold / slots_below_exact: csrc/gmx_resample.h)

// (slots_below: csrc/gmx_shard_fill.h)

__global__ void __launch_bounds__(GMX_BLOCK)
k_offspring(int kind, uint32_t k0, uint32_t k1, const uint64_t* __restrict__ cdf, int64_t n_in,
            uint64_t cdf_offset, const uint64_t* __restrict__ total_d, int64_t n_out_total,
            int64_t slot_offset, int64_t n_slots, int32_t* __restrict__ anc, int64_t u0_fixed = -1) {
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  const bool in_range = i < n_in;
  const uint64_t total = *total_d;
  gmx_key key; key.k0 = k0; key.k1 = k1;
  // u0_fixed >= 0 (systematic): the offset is given instead of drawn — 0 makes the output the GUIDE TABLE of the CDF,
  // guide[k] = first i with cdf_i * n_out > k * total (gmx_multinomial)
  const uint64_t u0 = u0_fixed >= 0 ? (uint64_t)u0_fixed : (uint64_t)(gmx_bits32(key, 0) >> 9);
  if (u0_fixed >= 0 && blockIdx.x == 0 && threadIdx.x < 4) anc[n_slots + threadIdx.x] = (int32_t)n_in;   // the table's end
  const uint64_t D = (uint64_t)n_out_total << 23;
  if (total == 0) {                       // no mass at all: everything maps to the last particle
    for (int64_t s = i; s < n_slots; s += (int64_t)gridDim.x * GMX_BLOCK) anc[s] = (int32_t)(n_in - 1);
    return;
  }
  const double n_over_total = (double)n_out_total / (double)total;
  const double eps = (double)n_out_total * 0x1p-44 + 0x1p-40;
  const uint64_t c_hi = in_range ? cdf[i] + cdf_offset : total;
  int64_t e = slots_below(kind, key, u0, c_hi, D, total, n_over_total, eps, n_out_total);
  // the lower bound of source i is the upper bound of source i-1: take it from
  // the neighbouring lane; lane 0 of each wave evaluates it itself
  const int lane = threadIdx.x & 63;
  uint32_t e_lo = (uint32_t)e, e_hi32 = (uint32_t)((uint64_t)e >> 32);
  e_lo = __shfl_up(e_lo, 1, GMX_WAVE); e_hi32 = __shfl_up(e_hi32, 1, GMX_WAVE);
  int64_t s = (int64_t)(((uint64_t)e_hi32 << 32) | e_lo);
  if (lane == 0) {
    const uint64_t c_lo = (i == 0 || !in_range) ? cdf_offset : cdf[i - 1] + cdf_offset;
    s = (i == 0) ? slots_below(kind, key, u0, cdf_offset, D, total, n_over_total, eps, n_out_total)
                 : slots_below(kind, key, u0, c_lo, D, total, n_over_total, eps, n_out_total);
  }
  // clip to the slot range this call owns
  if (s < slot_offset) s = slot_offset;
  if (e > slot_offset + n_slots) e = slot_offset + n_slots;
  if (!in_range) e = s;
  // a source with a few offspring writes them itself; a LONG run (a heavy particle: skewed weights) is written by the
  // whole wave, 64 consecutive slots per pass — otherwise one lane would loop over all of it while 63 wait (a weight
  // vector with all its mass on one particle: n_out iterations in one thread)
  const int64_t cnt = e - s;
  constexpr int64_t OWN = 8;
  if (cnt <= OWN)
    for (int64_t j = s; j < e; ++j) anc[j - slot_offset] = (int32_t)i;
  uint64_t heavy = __ballot(cnt > OWN);
  while (heavy) {                                  // wave-uniform
    const int l = __ffsll((unsigned long long)heavy) - 1;
    heavy &= heavy - 1ull;
    const int64_t s_l = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)s >> 32), l) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s, l));
    const int64_t e_l = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)e >> 32), l) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, l));
    const int32_t i_l = (int32_t)(i - lane + l);   // lanes hold consecutive sources
    for (int64_t j = s_l + lane; j < e_l; j += GMX_WAVE) anc[j - slot_offset] = i_l;
  }
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_ancestors(int kind, uint32_t k0, uint32_t k1, const uint64_t* __restrict__ cdf, int64_t n_in,
            uint64_t cdf_offset, const uint64_t* __restrict__ total_d, int64_t n_out_total,
            int64_t slot_offset, int64_t n_slots, int32_t* __restrict__ anc) {
  int64_t s = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (s >= n_slots) return;
  const uint64_t total = *total_d;
  const int64_t j = slot_offset + s;
  gmx_key key; key.k0 = k0; key.k1 = k1;
  uint64_t D;      // multiplier of the CDF side
  u128 P;          // threshold: ancestor = first i with (cdf_i + off) * D > P
  if (kind == GMX_RESAMPLE_MULTINOMIAL) {
    uint64_t u = gmx_bits32(key, (uint64_t)j) >> 9;
    D = 1ull << 23;
    P = mul64(total, (1ull << 23) - u);
    // cdf*D >= P  <=>  cdf*D > P - 1   (P >= 1 whenever total >= 1)
    if (P.lo == 0) { if (P.hi) { P.hi -= 1; P.lo = ~0ull; } } else P.lo -= 1;
  } else {
    uint64_t u0 = gmx_bits32(key, 0) >> 9;
    D = (uint64_t)n_out_total << 23;
    P = slot_threshold(kind, key, u0, j, total);
  }
  // binary search for the first i in [0, n_in) with (cdf[i] + off) * D > P
  int64_t lo = 0, hi = n_in;            // answer in [lo, hi]; hi == n_in means none
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    u128 c = mul64(cdf[mid] + cdf_offset, D);
    if (gt128(c, P)) hi = mid; else lo = mid + 1;
  }
  if (lo >= n_in) lo = n_in - 1;        // only reachable when total == 0 or off-shard
  anc[s] = (int32_t)lo;
}

// Multinomial over a large CDF: the per-slot search in two levels.  Every block first samples the CDF at stride
// S = 2^log2S (<= 1024 samples, L2-resident: all blocks read the same lines) into LDS; a slot then finds its
// window with LDS reads and finishes with log2(S) dependent global reads instead of log2(n_in).
// The threshold is an integer: first i with (cdf_i + off) * 2^23 >= P  <=>  cdf_i + off >= ceil(P / 2^23).
#define MN_SLOTS_PER_THREAD 8
#define MN_COARSE 1024
__global__ void __launch_bounds__(GMX_BLOCK)
k_ancestors_mn(uint32_t k0, uint32_t k1, const uint64_t* __restrict__ cdf, int64_t n_in, uint64_t cdf_offset,
               const uint64_t* __restrict__ total_d, int64_t slot_offset, int64_t n_slots, int32_t* __restrict__ anc,
               int log2S) {
  __shared__ uint64_t s_coarse[MN_COARSE];
  const int64_t S = (int64_t)1 << log2S;
  const int E = (int)((n_in + S - 1) >> log2S);            // <= MN_COARSE
  for (int e = (int)threadIdx.x; e < E; e += GMX_BLOCK) {
    int64_t last = ((int64_t)(e + 1) << log2S) - 1;
    if (last > n_in - 1) last = n_in - 1;
    s_coarse[e] = cdf[last] + cdf_offset;
  }
  __syncthreads();
  const uint64_t total = *total_d;
  gmx_key key; key.k0 = k0; key.k1 = k1;
#pragma unroll 2
  for (int r = 0; r < MN_SLOTS_PER_THREAD; ++r) {
    const int64_t s = ((int64_t)blockIdx.x * MN_SLOTS_PER_THREAD + r) * GMX_BLOCK + threadIdx.x;
    if (s >= n_slots) continue;
    const uint64_t u = gmx_bits32(key, (uint64_t)(slot_offset + s)) >> 9;
    u128 P = mul64(total, (1ull << 23) - u);                    // P >= 1 whenever total >= 1
    // T = ceil(P / 2^23)
    uint64_t lo = P.lo + ((1ull << 23) - 1ull);
    uint64_t hi = P.hi + (lo < P.lo ? 1ull : 0ull);
    const uint64_t T = (hi << 41) | (lo >> 23);
    // coarse: first window whose last element reaches T
    int a = 0, b = E;                                           // answer in [a, b]; b == E means none
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (s_coarse[mid] >= T) b = mid; else a = mid + 1;
    }
    int64_t res;
    if (a >= E || total == 0ull) {
      res = n_in - 1;                                           // no mass at all, or a slot beyond this shard
    } else {
      int64_t l = (int64_t)a << log2S, h = l + S - 1;           // cdf[h] + off >= T is known
      if (h > n_in - 1) h = n_in - 1;
      while (l < h) {
        const int64_t mid = l + ((h - l) >> 1);
        if (cdf[mid] + cdf_offset >= T) h = mid; else l = mid + 1;
      }
      res = l;
    }
    anc[s] = (int32_t)res;
  }
}

// Multinomial through a GUIDE TABLE (the whole, unsharded problem): guide[k] = first i with cdf_i * G > k * total,
// k = 0 .. G - 1 (G = n_in; guide[G .. G + 3] = n_in) — which is k_offspring's output for a systematic resampling with
// offset 0.  A slot with threshold T (first i with cdf_i >= T) then knows its answer lies in
// [guide[k'], guide[k' + 3]] for k' = floor((T - 1) G / total) - 1 (an f64 estimate, one below so that rounding cannot
// overshoot): on average three candidates instead of n_in — two table reads and ~2 CDF reads instead of ~20 dependent
// ones.  Every comparison that decides is the integer one, so the ancestors are k_ancestors' exactly.
__global__ void __launch_bounds__(GMX_BLOCK)
k_ancestors_guided(uint32_t k0, uint32_t k1, const uint64_t* __restrict__ cdf, int64_t n_in,
                   const uint64_t* __restrict__ total_d, const int32_t* __restrict__ guide, int64_t n_slots,
                   int32_t* __restrict__ anc) {
  const int64_t s = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (s >= n_slots) return;
  const uint64_t total = *total_d;
  if (total == 0ull) { anc[s] = (int32_t)(n_in - 1); return; }      // no mass at all
  gmx_key key; key.k0 = k0; key.k1 = k1;
  const uint64_t u = gmx_bits32(key, (uint64_t)s) >> 9;
  const u128 P = mul64(total, (1ull << 23) - u);                    // P >= 1
  const uint64_t lo64 = P.lo + ((1ull << 23) - 1ull);
  const uint64_t hi64 = P.hi + (lo64 < P.lo ? 1ull : 0ull);
  const uint64_t T = (hi64 << 41) | (lo64 >> 23);                   // ceil(P / 2^23), in [1, total]
  const uint64_t t1 = T - 1ull;
  const double v = __builtin_fma((double)(uint32_t)(t1 >> 32), 4294967296.0, (double)(uint32_t)t1) *
                   ((double)n_in / (double)total);                  // (T - 1) G / total, relative error 2^-50
  int64_t k = (int64_t)v - 1;                                       // <= floor((T - 1) G / total), >= that - 2
  k = k < 0 ? 0 : (k > n_in - 1 ? n_in - 1 : k);
  // guide[k], guide[k + 3] in one 16-byte read; then the first three candidates in one round of CDF reads
  typedef int32_t i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
  const i32x4_a4 g4 = *reinterpret_cast<const i32x4_a4*>(guide + k);
  int64_t l = g4.x, h = g4.w;                                       // cdf_{l-1} < T <= cdf_h  (h == n_in: past the end)
  if (h > n_in - 1) h = n_in - 1;                                   // cdf_{n_in-1} = total >= T
  const int64_t i1 = l + 1 < h ? l + 1 : h, i2 = l + 2 < h ? l + 2 : h;
  const uint64_t c0 = cdf[l], c1 = cdf[i1], c2 = cdf[i2];
  int64_t res;
  if (c0 >= T) res = l;
  else if (c1 >= T) res = i1;
  else if (c2 >= T) res = i2;
  else {                                                            // a crowded cell of the table: search the rest
    l = i2 + 1;
    while (l < h) {
      const int64_t mid = l + ((h - l) >> 1);
      if (cdf[mid] >= T) h = mid; else l = mid + 1;
    }
    res = l;
  }
  anc[s] = (int32_t)res;
}

extern "C" size_t gmx_multinomial_workspace(int64_t n_in) { return (size_t)(n_in + 4) * 4; }

extern "C" int gmx_multinomial(const uint32_t key[2], const uint64_t* cdf_d, int64_t n_in, const uint64_t* total_d,
                               int64_t n_out, int32_t* ancestors_d, void* workspace_d, gmx_stream stream) {
  if (n_out <= 0) return 0;
  if (!key || !cdf_d || !total_d || !ancestors_d || !workspace_d) return gmx_fail("gmx_multinomial: null argument%s");
  if (n_in <= 0 || n_in > 0x7ffffff0LL) return gmx_fail("gmx_multinomial: n_in out of range%s");
  if (n_out >= (1LL << 40)) return gmx_fail("gmx_multinomial: n_out out of range%s");
  if ((uintptr_t)workspace_d & 3) return gmx_fail("gmx_multinomial: workspace_d must be 4-byte aligned%s");
  int32_t* guide = (int32_t*)workspace_d;
  hipStream_t st = (hipStream_t)stream;
  auto grid_for = [](int64_t m) { return dim3((unsigned)((m + GMX_BLOCK - 1) / GMX_BLOCK)); };
  // guide table = the systematic offspring assignment of G = n_in slots with offset 0
  hipLaunchKernelGGL(k_offspring, grid_for(n_in), dim3(GMX_BLOCK), 0, st, (int)GMX_RESAMPLE_SYSTEMATIC, key[0], key[1],
                     cdf_d, n_in, (uint64_t)0, total_d, n_in, (int64_t)0, n_in, guide, (int64_t)0);
  hipLaunchKernelGGL(k_ancestors_guided, grid_for(n_out), dim3(GMX_BLOCK), 0, st, key[0], key[1], cdf_d, n_in, total_d,
                     guide, n_out, ancestors_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_ancestors(int kind, const uint32_t key[2], const uint64_t* cdf_d, int64_t n_in,
                             uint64_t cdf_offset, const uint64_t* total_d, int64_t n_out_total,
                             int64_t slot_offset, int64_t n_slots, int32_t* ancestors_d,
                             gmx_stream stream) {
  if (n_slots <= 0) return 0;
  if (!key || !cdf_d || !total_d || !ancestors_d) return gmx_fail("gmx_ancestors: null argument%s");
  if (kind < 0 || kind > 2) return gmx_fail("gmx_ancestors: unknown kind%s");
  if (n_in <= 0 || n_in > 0x7fffffffLL) return gmx_fail("gmx_ancestors: n_in out of range%s");
  if (n_out_total <= 0 || n_out_total >= (1LL << 40))
    return gmx_fail("gmx_ancestors: n_out_total out of range%s");
  if (slot_offset < 0 || slot_offset + n_slots > n_out_total)
    return gmx_fail("gmx_ancestors: slot range outside [0, n_out_total)%s");
  if (kind == GMX_RESAMPLE_MULTINOMIAL && n_in >= 8192) {
    int log2S = 0;
    while (((n_in + ((int64_t)1 << log2S) - 1) >> log2S) > MN_COARSE) ++log2S;
    const int64_t per_block = (int64_t)GMX_BLOCK * MN_SLOTS_PER_THREAD;
    hipLaunchKernelGGL(k_ancestors_mn, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(GMX_BLOCK), 0,
                       (hipStream_t)stream, key[0], key[1], cdf_d, n_in, cdf_offset, total_d, slot_offset, n_slots,
                       ancestors_d, log2S);
  } else if (kind == GMX_RESAMPLE_MULTINOMIAL || cdf_offset != 0 || slot_offset != 0 || n_slots != n_out_total) {
    // unordered positions, or a shard of a larger problem: search per slot
    hipLaunchKernelGGL(k_ancestors, grid_for(n_slots), dim3(GMX_BLOCK), 0, (hipStream_t)stream, kind,
                       key[0], key[1], cdf_d, n_in, cdf_offset, total_d, n_out_total, slot_offset,
                       n_slots, ancestors_d);
  } else {
    hipLaunchKernelGGL(k_offspring, grid_for(n_in), dim3(GMX_BLOCK), 0, (hipStream_t)stream, kind,
                       key[0], key[1], cdf_d, n_in, cdf_offset, total_d, n_out_total, slot_offset,
                       n_slots, ancestors_d);
  }
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// fused resampling (systematic / stratified, one GPU): log-weights -> ancestors
// with NO inter-block waiting and no CDF in memory.  The two-level CDF needs
// only two numbers per 1024-particle tile before anything global can be said:
//   tile stats       m_b = max lw, A_b = sum floor(exp(lw - k_b ln 2) * 2^shift), k_b = ceil(m_b / ln 2)
//                    written by the site program itself (a specialised kernel's
//                    OP_REDMAX epilogue, gmx_run_args.tile_agg_d) or by k_tile_stats
//   k_offspring_tile every block reads the <= RS_MAX_TILES tile stats (<= 24 KB from
//                    L2): M = max m_b, G_b = A_b * 2^(k_b - K) (integer shifts), its tile's
//                    prefix and the total; rebuilds its tile's local CDF from the
//                    log-weights in registers; assigns offspring exactly as
//                    k_offspring does.
// cdf_i = prefix[tile(i)] + (L_i >> (K - k_b)) is the integer k_weight_cdf produces,
// so ancestors are identical.
// ---------------------------------------------------------------------------

struct rs_ws {                 // layout of the gmx_resample workspace
  uint64_t agg[RS_MAX_TILES];  // A_b
  float tmax[RS_MAX_TILES];    // m_b
};


__global__ void __launch_bounds__(RS_THREADS)
k_tile_stats(const float* __restrict__ lw, int64_t n, float scale, float* __restrict__ tmax, uint64_t* __restrict__ agg) {
  __shared__ float lds4[4];
  __shared__ uint64_t s_sum[RS_THREADS / GMX_WAVE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)threadIdx.x * CDF_VEC;
  float x[CDF_VEC];
  if (base + CDF_VEC <= n) {
    float4 v = *reinterpret_cast<const float4*>(lw + base);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) x[c] = (base + c < n) ? lw[base + c] : -gmx_inf();
  }
  float m = -gmx_inf();
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) m = gmx_rmax(m, x[c]);
  m = block_max(m, lds4);
  const float ref = gmx_tile_ref(gmx_tile_exp(m));
  uint64_t run = 0;
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) run += (base + c < n) ? weight_fixed(x[c], ref, scale) : 0ull;
  run = wave_sum_u64(run);
  if (lane == 0) s_sum[wave] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    tmax[blockIdx.x] = m;
    agg[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
  }
}

// (slots_below_est: csrc/gmx_resample.h)

// (k_offspring_tile's body: csrc/gmx_offspring.h — shared with the specialised site programs that resample first)
template <int kind, int PER>
__global__ void __launch_bounds__(RS_BLOCK)
k_offspring_tile(uint32_t k0, uint32_t k1, uint32_t u0_host, const float* __restrict__ lw,
                 const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int64_t n, int n_tiles, float scale,
                 float* __restrict__ max_out, uint64_t* __restrict__ total_out, int32_t* __restrict__ anc,
                 const uint32_t* __restrict__ uslot) {
  GMX_SETPRIO
  gmx_offspring_tile_body<kind, PER, false>(k0, k1, u0_host, lw, tmax, agg, n, n_tiles, scale, max_out, total_out, anc, uslot, 0u);
}

static int resample_shape(const char* who, int64_t n, int shift, bool any_n = false) {
  if (n <= 0) return gmx_fail("%s: n must be positive", who);
  if (!any_n && (n + RS_TILE - 1) / RS_TILE > RS_MAX_TILES)
    return gmx_fail("%s: n too large for the fused path (gmx_tile_stats + gmx_tile_prefix + gmx_resample_tiles_p, or gmx_weight_cdf + gmx_ancestors)", who);
  if (n > 0x7fffffffLL) return gmx_fail("%s: n out of range", who);
  if (shift < 1 || shift > 62) return gmx_fail("%s: shift out of range", who);
  int need = 0;
  while (((int64_t)1 << need) < n) ++need;
  if (shift + need > 62) return gmx_fail("%s: shift too large for n (overflow)", who);
  return 0;
}

static int launch_offspring_tile(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                                 const float* tile_max_d, const uint64_t* tile_agg_d, float* max_d, uint64_t* total_d,
                                 int32_t* ancestors_d, gmx_stream stream, bool pref = false, const uint32_t* u_d = nullptr) {
  uint32_t b0, b1;
  gmx_threefry2x32(key[0], key[1], 0u, 0u, &b0, &b1);           // bits32(key, 0) on the host
  const uint32_t u0 = (b0 ^ b1) >> 9;
  const int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
  const dim3 grid((unsigned)((tiles + RS_TPB - 1) / RS_TPB)), block(RS_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  const float scale = gmx_pow2i(shift);
#define GMX_LAUNCH_OT2(KIND, PER_)                                                                                   \
  hipLaunchKernelGGL((k_offspring_tile<KIND, PER_>), grid, block, 0, st, key[0], key[1], u0, lw_d, tile_max_d,        \
                     tile_agg_d, n, (int)tiles, scale, max_d, total_d, ancestors_d, u_d)
#define GMX_LAUNCH_OT(KIND)                                                                                         \
  do {                                                                                                              \
    if (pref) GMX_LAUNCH_OT2(KIND, 0);          /* `tile_agg_d` is the prefix block gmx_tile_prefix wrote */         \
    else if (tiles <= 1 * RS_BLOCK) GMX_LAUNCH_OT2(KIND, 1);                                                        \
    else if (tiles <= 2 * RS_BLOCK) GMX_LAUNCH_OT2(KIND, 2);                                                        \
    else if (tiles <= 4 * RS_BLOCK) GMX_LAUNCH_OT2(KIND, 4);                                                        \
    else GMX_LAUNCH_OT2(KIND, 8);                                                                                   \
  } while (0)
  if (kind == GMX_RESAMPLE_SYSTEMATIC) GMX_LAUNCH_OT(GMX_RESAMPLE_SYSTEMATIC);
  else if (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED) {        // from the log-weights, the 977-fold statistics pass
    GMX_LAUNCH_OT(GMX_RESAMPLE_MULTINOMIAL_SORTED);
  }
  else GMX_LAUNCH_OT(GMX_RESAMPLE_STRATIFIED);
#undef GMX_LAUNCH_OT
#undef GMX_LAUNCH_OT2
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_tile_stats(const float* lw_d, int64_t n, int shift, float* tile_max_d, uint64_t* tile_agg_d,
                              gmx_stream stream) {
  if (resample_shape("gmx_tile_stats", n, shift, true)) return 1;
  if (!lw_d || !tile_max_d || !tile_agg_d) return gmx_fail("gmx_tile_stats: null argument%s");
  if ((uintptr_t)lw_d & 15) return gmx_fail("gmx_tile_stats: lw_d must be 16-byte aligned%s");
  const int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
  hipLaunchKernelGGL(k_tile_stats, dim3((unsigned)tiles), dim3(RS_THREADS), 0, (hipStream_t)stream, lw_d, n,
                     gmx_pow2i(shift), tile_max_d, tile_agg_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_resample_tiles(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                                  const float* tile_max_d, const uint64_t* tile_agg_d, float* max_d,
                                  uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream) {
  if (resample_shape("gmx_resample_tiles", n, shift)) return 1;
  if (!key || !lw_d || !tile_max_d || !tile_agg_d || !max_d || !total_d || !ancestors_d)
    return gmx_fail("gmx_resample_tiles: null argument%s");
  if (kind != GMX_RESAMPLE_SYSTEMATIC && kind != GMX_RESAMPLE_STRATIFIED)
    return gmx_fail("gmx_resample_tiles: kind must be systematic or stratified (use gmx_weight_cdf + gmx_ancestors)%s");
  if (n > 0x7fffffffLL) return gmx_fail("gmx_resample_tiles: n out of range%s");
  if ((uintptr_t)lw_d & 15) return gmx_fail("gmx_resample_tiles: lw_d must be 16-byte aligned%s");
  return launch_offspring_tile(kind, key, lw_d, n, shift, tile_max_d, tile_agg_d, max_d, total_d, ancestors_d, stream);
}

// ---- two-stage multinomial resampling (include/genmi.h: gmx_multinomial_tiled) ----
// Multinomial resampling as the oracle first defined it (GMX_RESAMPLE_MULTINOMIAL: slot j takes its own uniform and
// the first particle whose CDF exceeds it) is bound by random cache lines: every slot ends in a random tile's CDF and
// the gather behind it reads a random particle.  Two stages remove the randomness from MEMORY:
//   k_mn_hist  every workgroup builds the end-of-tile CDF (<= 2048 entries) in LDS from the tile statistics; a slot's
//              uniform picks its tile there (11 LDS probes), counted in an LDS histogram, flushed with one integer atomic
//              per (workgroup, tile) — integer sums: order-independent, bit-reproducible
//   k_mn_tile  workgroup b owns tile b: the tile's CDF rebuilt in LDS from its log-weights, the counts table reduced
//              to (offset, count) of the tile, then count_b slots each probe the LOCAL CDF (10 LDS probes) and write
//              consecutive output positions
// The output is ordered by tile (iid inside a tile), so the following gather streams.  Same offspring law.
#define MN_HIST_BLOCKS 256
__device__ __forceinline__ uint64_t mn_scale23(uint32_t u, uint64_t m) {      // (u * m) >> 23 for u < 2^23, m < 2^62
  const uint64_t lo = (m & 0xffffffffull) * (uint64_t)u, hi = (m >> 32) * (uint64_t)u;
  return (hi << 9) + (lo >> 23);
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_mn_hist(uint32_t k0, uint32_t k1, const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int64_t n,
          int n_tiles, const uint32_t* __restrict__ u_d, float* __restrict__ max_out, uint64_t* __restrict__ total_out,
          uint32_t* __restrict__ counts) {
  __shared__ uint64_t s_cend[RS_MAX_TILES];
  __shared__ uint32_t s_hist[RS_MAX_TILES];
  __shared__ uint64_t s_w[4];
  __shared__ float s_max[4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PERT = RS_MAX_TILES / GMX_BLOCK;            // 8 consecutive table rows per thread
  float tm[PERT];
  uint64_t ta[PERT];
  float m = -gmx_inf();
#pragma unroll
  for (int r = 0; r < PERT; ++r) {
    const int t = tid * PERT + r;
    const int tc = t < n_tiles ? t : n_tiles - 1;
    tm[r] = tmax[tc]; ta[r] = agg[tc];
    s_hist[t] = 0u;
  }
#pragma unroll
  for (int r = 0; r < PERT; ++r) m = gmx_rmax(m, (tid * PERT + r < n_tiles) ? tm[r] : -gmx_inf());
  m = wave_max(m);
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  const float M = gmx_rmax(gmx_rmax(s_max[0], s_max[1]), gmx_rmax(s_max[2], s_max[3]));
  const int32_t K = gmx_tile_exp(M);
  uint64_t g[PERT], run = 0;
#pragma unroll
  for (int r = 0; r < PERT; ++r) {
    run += (tid * PERT + r < n_tiles) ? gmx_tile_scale(ta[r], gmx_tile_exp(tm[r]), K) : 0ull;
    g[r] = run;
  }
  const uint64_t inc = wave_scan_u64(run);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint64_t off = inc - run;
#pragma unroll
  for (int w = 0; w < 4; ++w) off += (w < wave) ? s_w[w] : 0ull;
#pragma unroll
  for (int r = 0; r < PERT; ++r) s_cend[tid * PERT + r] = off + g[r];       // rows past n_tiles repeat the total
  __syncthreads();
  const uint64_t total = s_cend[n_tiles - 1];
  if (blockIdx.x == 0 && tid == 0) { *total_out = total; *max_out = M; }
  if (total == 0) return;                                   // no mass: k_mn_tile maps every slot to the last particle
  gmx_key key; key.k0 = k0; key.k1 = k1;
  const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t j_lo = (int64_t)blockIdx.x * per_block, j_hi = j_lo + per_block < n ? j_lo + per_block : n;
  // four slots per thread and trip: four independent searches in flight hide the LDS round trips.  A search starts
  // from the tile a flat weight vector would give (u * n_tiles) and gallops: tile masses rarely differ by more than
  // a small factor, so it ends after a few probes instead of log2(n_tiles) = 11.
  const float tiles_f = (float)n_tiles * (1.0f / 8388608.0f);
  // (the uniforms of the NEXT trip are loaded while this one walks: a trip would otherwise start with a global round trip)
  uint32_t un[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t j = j_lo + tid + (int64_t)q * GMX_BLOCK;
    const int64_t jc = j < j_hi ? j : j_hi - 1;
    un[q] = u_d ? u_d[jc] : 0u;
  }
  for (int64_t j0 = j_lo + tid; j0 < j_hi; j0 += 4 * GMX_BLOCK) {
    uint64_t P[4];
    int b[4];
    bool ok[4];
    uint32_t uc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uc[q] = un[q];
      const int64_t jn = j0 + (int64_t)(q + 4) * GMX_BLOCK;
      const int64_t jnc = jn < j_hi ? jn : j_hi - 1;
      un[q] = u_d ? u_d[jnc] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t j = j0 + (int64_t)q * GMX_BLOCK;
      ok[q] = j < j_hi;
      const int64_t jc = ok[q] ? j : j_hi - 1;
      const uint32_t u = u_d ? uc[q] : (gmx_bits32(key, (uint64_t)jc) >> 9);
      P[q] = mn_scale23(u, total);
      int g = (int)((float)u * tiles_f);
      b[q] = g < n_tiles ? g : n_tiles - 1;
    }
    // the answer is the first tile with cend > P: walk down while the tile before still exceeds P, up while this one
    // does not (both loops end: cend is non-decreasing and cend[n_tiles - 1] = total > P)
    bool moving = true;
    int it = 0;
#pragma unroll 1
    while (moving && it < 6) {
      moving = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t here = s_cend[b[q]];
        const uint64_t prev = b[q] > 0 ? s_cend[b[q] - 1] : 0ull;
        const bool up = !(here > P[q]), down = b[q] > 0 && prev > P[q];
        b[q] += up ? 1 : (down ? -1 : 0);
        moving |= up | down;
      }
      ++it;
    }
    if (moving) {                         // strongly uneven tile masses: the plain binary search
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int lo = 0, hi = n_tiles;
#pragma unroll 1
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_cend[mid] > P[q]) hi = mid; else lo = mid + 1;
        }
        b[q] = lo;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ok[q]) atomicAdd(&s_hist[b[q]], 1u);
  }
  __syncthreads();
  for (int t = tid; t < n_tiles; t += GMX_BLOCK) {
    const uint32_t c = s_hist[t];
    if (c) atomicAdd(&counts[t], c);
  }
}

template <int PER>
__global__ void __launch_bounds__(GMX_BLOCK)
k_mn_tile(uint32_t k0, uint32_t k1, const float* __restrict__ lw, const float* __restrict__ tmax,
          const uint64_t* __restrict__ agg, int64_t n, int n_tiles, float scale, const uint32_t* __restrict__ counts,
          uint32_t* __restrict__ zero_other, int32_t* __restrict__ anc) {
  GMX_SETPRIO
  __shared__ uint64_t s_cdf[RS_TILE];
  __shared__ __attribute__((aligned(16))) uint32_t s_guide[RS_TILE];
  __shared__ uint64_t s_scan[4], s_all[4];
  __shared__ uint32_t s_cb[4], s_gc[4];
  __shared__ float s_max[4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int my_tile = (int)blockIdx.x;
  const int64_t i0 = (int64_t)my_tile * RS_TILE + (int64_t)tid * CDF_VEC;
  float x[CDF_VEC];
  if ((int64_t)(my_tile + 1) * RS_TILE <= n) {
    const float4 v = *reinterpret_cast<const float4*>(lw + i0);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) {
      const int64_t ic = i0 + c < n ? i0 + c : n - 1;
      const float xv = lw[ic];
      x[c] = (i0 + c < n) ? xv : -gmx_inf();
    }
  }
  // the table: global max (-> K), the total (zero: no mass), and the slots of the tiles before mine
  float tm[PER];
  uint64_t ta[PER];
  uint32_t tc_[PER];
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const int t = r * GMX_BLOCK + tid;
    const int tc = t < n_tiles ? t : n_tiles - 1;
    tm[r] = tmax[tc]; ta[r] = agg[tc]; tc_[r] = counts[tc];
  }
  const float tmax_mine = tmax[my_tile];
  const uint64_t agg_mine = agg[my_tile];
  const uint32_t n_b = counts[my_tile];
  float m = -gmx_inf();
  uint32_t before = 0;
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const int t = r * GMX_BLOCK + tid;
    m = gmx_rmax(m, t < n_tiles ? tm[r] : -gmx_inf());
    before += (t < my_tile) ? tc_[r] : 0u;
  }
  m = wave_max(m);
  before = (uint32_t)wave_sum_u64((uint64_t)before);
  const int32_t k_b = gmx_tile_exp(tmax_mine);
  const float ref_b = gmx_tile_ref(k_b);
  uint64_t q[CDF_VEC], run = 0;
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) {
    run += (i0 + c < n) ? weight_fixed(x[c], ref_b, scale) : 0ull;
    q[c] = run;
  }
  const uint64_t inc = wave_scan_u64(run);
  if (lane == 0) { s_max[wave] = m; s_cb[wave] = before; }
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  const float M = gmx_rmax(gmx_rmax(s_max[0], s_max[1]), gmx_rmax(s_max[2], s_max[3]));
  const int32_t K = gmx_tile_exp(M);
  const uint32_t O_b = (s_cb[0] + s_cb[1]) + (s_cb[2] + s_cb[3]);
  uint64_t all = 0;
#pragma unroll
  for (int r = 0; r < PER; ++r)
    all += (r * GMX_BLOCK + tid < n_tiles) ? gmx_tile_scale(ta[r], gmx_tile_exp(tm[r]), K) : 0ull;
  all = wave_sum_u64(all);
  if (lane == 0) s_all[wave] = all;
  uint64_t wave_off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? s_scan[w] : 0ull;
  const uint64_t loc = wave_off + (inc - run);
  uint64_t edge[CDF_VEC + 1];                      // scaled local CDF at the lower edge of my first source and after each
  edge[0] = gmx_tile_scale(loc, k_b, K);
#pragma unroll
  for (int c = 0; c < CDF_VEC; ++c) {
    edge[c + 1] = gmx_tile_scale(loc + q[c], k_b, K);
    s_cdf[tid * CDF_VEC + c] = edge[c + 1];          // past n: repeats the last
  }
  reinterpret_cast<uint4*>(s_guide)[tid] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const uint64_t total = (s_all[0] + s_all[1]) + (s_all[2] + s_all[3]);
  if (total == 0) {
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c)
      if (i0 + c < n) anc[i0 + c] = (int32_t)(n - 1);
    if (zero_other && tid == 0) zero_other[my_tile] = 0u;
    return;
  }
  const uint64_t G_b = gmx_tile_scale(agg_mine, k_b, K);
  gmx_key k2; k2.k0 = k0; k2.k1 = k1;
  const gmx_key kb = gmx_fold_in(k2, (uint32_t)my_tile);
  const int32_t base_i = my_tile * RS_TILE;
  const int cnt_i = (int64_t)(my_tile + 1) * RS_TILE <= n ? RS_TILE : (int)(n - (int64_t)my_tile * RS_TILE);
  // a GUIDE over the local CDF instead of a search per slot: bucket g of 1024 covers the positions [g, g + 1) * G_b / 1024;
  // every particle with mass marks the first bucket that starts inside its interval, a max-scan fills the rest, and a
  // slot starts at guide[v >> 13] (its bucket: Q = v * G_b >> 23) and WALKS to the exact answer — the marks come from a
  // float product, so they may be one bucket off; the walk settles that with the exact integers.
  {
    const float inv = 1024.0f / (float)G_b;
#pragma unroll
    for (int c = 0; c < CDF_VEC; ++c) {
      if (edge[c + 1] > edge[c]) {
        const int gb = (int)__builtin_ceilf((float)edge[c] * inv);
        if (gb < RS_TILE) atomicMax(&s_guide[gb], (uint32_t)(tid * CDF_VEC + c));
      }
    }
    __syncthreads();
    uint4 a = reinterpret_cast<const uint4*>(s_guide)[tid];
    a.y = a.y > a.x ? a.y : a.x; a.z = a.z > a.y ? a.z : a.y; a.w = a.w > a.z ? a.w : a.z;
    const uint32_t incl = gmx_wave_umax_scan(a.w);
    uint32_t carry = wave_shr1_u32(incl, 0u);
    if (lane == 63) s_gc[wave] = incl;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; ++w) { const uint32_t v_ = s_gc[w]; carry = (w < wave && v_ > carry) ? v_ : carry; }
    a.x = a.x > carry ? a.x : carry; a.y = a.y > carry ? a.y : carry; a.z = a.z > carry ? a.z : carry; a.w = a.w > carry ? a.w : carry;
    reinterpret_cast<uint4*>(s_guide)[tid] = a;
    __syncthreads();
  }
  for (uint32_t r0 = (uint32_t)tid; r0 < n_b; r0 += 4 * GMX_BLOCK) {     // four independent walks per trip
    uint64_t Q[4];
    int at[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t r = r0 + (uint32_t)q * GMX_BLOCK;
      const uint32_t v = gmx_bits32(kb, (uint64_t)r) >> 9;
      Q[q] = mn_scale23(v, G_b);
      const int g0 = (int)s_guide[v >> 13];
      at[q] = g0 < cnt_i ? g0 : cnt_i - 1;
    }
    // the answer is the first particle with local cdf > Q (Q < G_b = cdf[last]: it exists)
    bool moving = true;
    int it = 0;
#pragma unroll 1
    while (moving && it < 24) {
      moving = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t here = s_cdf[at[q]];
        const uint64_t prev = at[q] > 0 ? s_cdf[at[q] - 1] : 0ull;
        const bool up = !(here > Q[q]) && at[q] + 1 < cnt_i, down = at[q] > 0 && prev > Q[q];
        at[q] += up ? 1 : (down ? -1 : 0);
        moving |= up | down;
      }
      ++it;
    }
    if (moving) {                          // many particles inside one bucket: the plain binary search
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int lo = 0, hi = cnt_i;
#pragma unroll 1
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_cdf[mid] > Q[q]) hi = mid; else lo = mid + 1;
        }
        at[q] = lo < cnt_i ? lo : cnt_i - 1;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t r = r0 + (uint32_t)q * GMX_BLOCK;
      const int64_t pos = (int64_t)O_b + r;
      if (r < n_b && pos < n) anc[pos] = base_i + at[q];
    }
  }
  if (zero_other && tid == 0) zero_other[my_tile] = 0u;     // leave the OTHER count buffer clean for the caller's next call
}

// two count buffers of (tiles + 16) words each
extern "C" size_t gmx_multinomial_tiled_workspace(int64_t n) { return (size_t)((n + RS_TILE - 1) / RS_TILE + 16) * 8; }

extern "C" int gmx_multinomial_tiled(const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                                     const float* tile_max_d, const uint64_t* tile_agg_d, const uint32_t* u_d,
                                     float* max_d, uint64_t* total_d, int32_t* ancestors_d, void* workspace_d,
                                     int phase, gmx_stream stream) {
  if (resample_shape("gmx_multinomial_tiled", n, shift)) return 1;
  if (!key || !lw_d || !tile_max_d || !tile_agg_d || !max_d || !total_d || !ancestors_d || !workspace_d)
    return gmx_fail("gmx_multinomial_tiled: null argument%s");
  if (n > 0x7fffffffLL) return gmx_fail("gmx_multinomial_tiled: n out of range%s");
  if (phase < -1 || phase > 1) return gmx_fail("gmx_multinomial_tiled: phase is -1, 0 or 1%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)workspace_d & 15))
    return gmx_fail("gmx_multinomial_tiled: lw_d and workspace_d must be 16-byte aligned%s");
  hipStream_t st = (hipStream_t)stream;
  const int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
  const size_t words = (size_t)tiles + 16;
  // The call counts into one of two buffers and leaves the OTHER one zero (every workgroup of k_mn_tile clears its own
  // entry there: nobody reads that buffer during the call).  phase 0 / 1: buffer `phase` — which the previous call,
  // made with the other phase, left zero: a caller that alternates needs no memset at all.  phase -1: buffer 0 after
  // zeroing it here (a memset node): the first call of a sequence, or a one-off.
  uint32_t* counts = (uint32_t*)workspace_d + (phase == 1 ? words : 0);
  uint32_t* other = (uint32_t*)workspace_d + (phase == 1 ? 0 : words);
  if (phase < 0) GMX_HIP(hipMemsetAsync(counts, 0, words * 4, st));
  // (k1, k2) = split(key, 2): two Threefry blocks on the host
  uint32_t k1[2], k2[2];
  gmx_threefry2x32(key[0], key[1], 0u, 0u, &k1[0], &k1[1]);
  gmx_threefry2x32(key[0], key[1], 0u, 1u, &k2[0], &k2[1]);
  const int64_t want = (n + 8 * GMX_BLOCK - 1) / (8 * GMX_BLOCK);              // >= 8 slots per thread
  const int max_hb = MN_HIST_BLOCKS;
  const unsigned hb = (unsigned)(want < 1 ? 1 : (want < max_hb ? want : max_hb));
  hipLaunchKernelGGL(k_mn_hist, dim3(hb), dim3(GMX_BLOCK), 0, st, k1[0], k1[1], tile_max_d, tile_agg_d, n, (int)tiles, u_d,
                     max_d, total_d, counts);
  const float scale = gmx_pow2i(shift);
#define GMX_LAUNCH_MT(PER_)                                                                                          \
  hipLaunchKernelGGL((k_mn_tile<PER_>), dim3((unsigned)tiles), dim3(GMX_BLOCK), 0, st, k2[0], k2[1], lw_d, tile_max_d,  \
                     tile_agg_d, n, (int)tiles, scale, (const uint32_t*)counts, other, ancestors_d)
  if (tiles <= 1 * GMX_BLOCK) GMX_LAUNCH_MT(1);
  else if (tiles <= 2 * GMX_BLOCK) GMX_LAUNCH_MT(2);
  else if (tiles <= 4 * GMX_BLOCK) GMX_LAUNCH_MT(4);
  else GMX_LAUNCH_MT(8);
#undef GMX_LAUNCH_MT
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---- the stratified resampler's per-slot uniforms, ahead of time ----
// out[r][j] = bits32(keys[r], j) >> 9: what slots_below_est<stratified> would draw for slot j of the resampling keyed
// keys[r].  They depend on keys and slot numbers only, so a sweep draws them on its background stream, a group of steps
// ahead (the same treatment as the steps' normal draws: priority 0, `lds_pad` bytes of unused LDS per workgroup cap the
// residency), and gmx_resample_tiles_u reads them.  2-D launch: one grid row per key.
__global__ void __launch_bounds__(GMX_BLOCK)
k_slot_uniforms(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ out) {
  gmx_key key;
  key.k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y]);
  key.k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y + 1]);
  const int64_t j0 = ((int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x) * 4;
  uint32_t* row = out + (int64_t)blockIdx.y * n;
  uint32_t u[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) u[c] = gmx_bits32(key, (uint64_t)(j0 + c)) >> 9;
  if (j0 + 4 <= n && (((uintptr_t)(row + j0)) & 15) == 0) {
    *reinterpret_cast<uint4*>(row + j0) = make_uint4(u[0], u[1], u[2], u[3]);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) if (j0 + c < n) row[j0 + c] = u[c];
  }
}

extern "C" int gmx_slot_uniforms(const uint32_t* keys_d, int rows, int64_t n, uint32_t* out_d, int lds_pad,
                                 gmx_stream stream) {
  if (!keys_d || !out_d) return gmx_fail("gmx_slot_uniforms: null argument%s");
  if (rows < 1 || rows > 65535 || n <= 0 || n > 0x7fffffffLL) return gmx_fail("gmx_slot_uniforms: rows / n out of range%s");
  if (lds_pad < 0 || lds_pad > 64 * 1024) return gmx_fail("gmx_slot_uniforms: lds_pad out of range (<= 64 KB)%s");
  hipLaunchKernelGGL(k_slot_uniforms, dim3((unsigned)((n + 4 * GMX_BLOCK - 1) / (4 * GMX_BLOCK)), (unsigned)rows),
                     dim3(GMX_BLOCK), (size_t)lds_pad, (hipStream_t)stream, keys_d, n, out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_resample_tiles_u(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                                    const float* tile_max_d, const uint64_t* tile_agg_d, const uint32_t* u_d,
                                    float* max_d, uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream) {
  if (resample_shape("gmx_resample_tiles_u", n, shift)) return 1;
  if (!key || !lw_d || !tile_max_d || !tile_agg_d || !u_d || !max_d || !total_d || !ancestors_d)
    return gmx_fail("gmx_resample_tiles_u: null argument%s");
  if (kind != GMX_RESAMPLE_STRATIFIED) return gmx_fail("gmx_resample_tiles_u: stratified only (systematic draws one uniform)%s");
  if (n > 0x7fffffffLL) return gmx_fail("gmx_resample_tiles_u: n out of range%s");
  if ((uintptr_t)lw_d & 15) return gmx_fail("gmx_resample_tiles_u: lw_d must be 16-byte aligned%s");
  return launch_offspring_tile(kind, key, lw_d, n, shift, tile_max_d, tile_agg_d, max_d, total_d, ancestors_d, stream,
                               false, u_d);
}

// ---- multinomial resampling with sorted uniforms (GMX_RESAMPLE_MULTINOMIAL_SORTED; csrc/gmx_sorted.h) ----
// The order-statistics table of a resampling depends on its key and n only: two launches, meant for the background
// stream of a sweep (2-D: one grid row per key, `lds_pad` caps the residency like the normals' programs).
//   k_sorted_exp    workgroup = 1024 slots: E_j, the tile-local inclusive sums (u32: < 2^31), the tile's sum
//   k_sorted_offsets one workgroup per key: the tile sums -> tile offsets, S_total, sh
//   k_sorted_guide  a workgroup rewrites its slots as the low words of the GLOBAL sums and fills the guide: slot j owns
//                   the buckets g in (bucket(S_{j-1}), bucket(S_j)] (one on average), written through LDS
__global__ void __launch_bounds__(GMX_BLOCK)
k_sorted_exp(const uint32_t* __restrict__ keys, uint32_t hk0, uint32_t hk1, int64_t n, size_t words, uint32_t* __restrict__ out) {
  __shared__ uint64_t s_w[GMX_BLOCK / GMX_WAVE];
  gmx_key key; key.k0 = hk0; key.k1 = hk1;
  if (keys) {
    key.k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y]);
    key.k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y + 1]);
  }
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  uint32_t* row = out + (size_t)blockIdx.y * words;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j0 = (int64_t)blockIdx.x * GMX_SORTED_TILE + (int64_t)threadIdx.x * 4;
  uint32_t q[4], ev[4];
  gmx_sorted_exp_pair(key, (uint64_t)(j0 >> 1), &ev[0], &ev[1]);          // j0 is a multiple of 4: two whole blocks
  gmx_sorted_exp_pair(key, (uint64_t)(j0 >> 1) + 1ull, &ev[2], &ev[3]);
  uint64_t run = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    run += (j0 + c < n) ? ev[c] : 0u;
    q[c] = (uint32_t)run;
  }
  const uint64_t inc = wave_scan_u64(run);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint64_t off = 0, all = 0;
#pragma unroll
  for (int w = 0; w < GMX_BLOCK / GMX_WAVE; ++w) { const uint64_t v = s_w[w]; all += v; off += (w < wave) ? v : 0ull; }
  const uint32_t base = (uint32_t)(off + (inc - run));
  *reinterpret_cast<uint4*>(row + j0) = make_uint4(base + q[0], base + q[1], base + q[2], base + q[3]);
  if (threadIdx.x == 0) reinterpret_cast<uint64_t*>(row + L.off_tsum)[blockIdx.x] = all;
}

// one workgroup per resampling: the tile sums -> the tiles' offsets, S_total and sh — in chunks of 2048 tiles (thread t
// owns 8 consecutive tiles of a chunk: a u64 wave scan and four wave totals) with a running carry, so any n < 2^31
__global__ void __launch_bounds__(GMX_BLOCK)
k_sorted_offsets(const uint32_t* __restrict__ keys, uint32_t hk0, uint32_t hk1, int64_t n, size_t words, uint32_t* __restrict__ out) {
  __shared__ uint64_t s_w[GMX_BLOCK / GMX_WAVE];
  gmx_key key; key.k0 = hk0; key.k1 = hk1;
  if (keys) {
    key.k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y]);
    key.k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y + 1]);
  }
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  uint32_t* row = out + (size_t)blockIdx.y * words;
  const uint64_t* tsum = reinterpret_cast<const uint64_t*>(row + L.off_tsum);
  uint64_t* toff = reinterpret_cast<uint64_t*>(row + L.off_toff);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tiles = (int64_t)L.tiles;
  constexpr int PERC = RS_MAX_TILES / GMX_BLOCK;                         // 8 tiles per thread and chunk
  uint64_t carry = 0;
  for (int64_t base = 0; base < tiles; base += RS_MAX_TILES) {          // block-uniform trip count (1 up to n = 2^21)
    uint64_t tv[PERC];
    uint64_t run = 0;
#pragma unroll
    for (int r = 0; r < PERC; ++r) {
      const int64_t t = base + (int64_t)threadIdx.x * PERC + r;
      const uint64_t v = tsum[t < tiles ? t : tiles - 1];
      tv[r] = (t < tiles) ? v : 0ull;
      run += tv[r];
    }
    const uint64_t inc = wave_scan_u64(run);
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint64_t off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < GMX_BLOCK / GMX_WAVE; ++w) { const uint64_t v = s_w[w]; all += v; off += (w < wave) ? v : 0ull; }
    uint64_t at = carry + off + (inc - run);
#pragma unroll
    for (int r = 0; r < PERC; ++r) {
      const int64_t t = base + (int64_t)threadIdx.x * PERC + r;
      if (t < tiles) toff[t] = at;
      at += tv[r];
    }
    carry += all;
  }
  if (threadIdx.x == 0) {
    const uint64_t stot = carry + gmx_sorted_exp(key, (uint64_t)n);
    toff[tiles] = stot;
    row[L.off_sh] = gmx_sorted_shift(stot, L.ng);
  }
}

#define SORTED_FILL 2048               /* guide entries filled per pass (8 per thread): a tile owns ~1024 */
__global__ void __launch_bounds__(GMX_BLOCK)
k_sorted_guide(const uint32_t* __restrict__ keys, uint32_t hk0, uint32_t hk1, int64_t n, size_t words, uint32_t* __restrict__ out) {
  __shared__ uint32_t s_last[GMX_BLOCK];
  __shared__ __attribute__((aligned(16))) uint32_t s_mark[SORTED_FILL];
  __shared__ uint32_t s_carry[GMX_BLOCK / GMX_WAVE];
  gmx_key key; key.k0 = hk0; key.k1 = hk1;
  if (keys) {
    key.k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y]);
    key.k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2 * blockIdx.y + 1]);
  }
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  uint32_t* row = out + (size_t)blockIdx.y * words;
  uint32_t* guide = row + L.off_guide;
  const uint64_t* tsum = reinterpret_cast<const uint64_t*>(row + L.off_tsum);
  const uint64_t* toff = reinterpret_cast<const uint64_t*>(row + L.off_toff);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = (int)blockIdx.x, tiles = (int)L.tiles;
  const int64_t j0 = (int64_t)tile * GMX_SORTED_TILE + (int64_t)threadIdx.x * 4;
  const uint4 loc = *reinterpret_cast<const uint4*>(row + j0);
  const uint64_t own = tsum[tile];
  const uint64_t below = toff[tile], stot = toff[tiles];        // k_sorted_offsets wrote them
  const uint32_t sh = row[L.off_sh];
  s_last[threadIdx.x] = loc.w;
  __syncthreads();
  // the tile's slots as low words of the global sums; slot j owns the guide entries (bucket(S_{j-1}), bucket(S_j)]
  const uint32_t lv[4] = {loc.x, loc.y, loc.z, loc.w};
  const int32_t gA = tile == 0 ? 0 : (int32_t)(below >> sh) + 1;             // the tile's first guide entry ...
  const int32_t gB = (int32_t)((below + own) >> sh);                         // ... and its last (buckets < NG < 2^22)
  uint64_t prev = below + (threadIdx.x ? s_last[threadIdx.x - 1] : 0u);      // S_{j0 - 1}
  uint32_t lowv[4];
  int32_t st[4], en[4];                                                      // entries [st, en] get the value j
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t j = j0 + c;
    const uint64_t S = below + lv[c];
    lowv[c] = (j < n) ? (uint32_t)S : 0u;
    st[c] = (j == 0) ? 0 : (int32_t)(prev >> sh) + 1;
    en[c] = (j < n) ? (int32_t)(S >> sh) : st[c] - 1;
    if (j == n - 1) {                      // past the last slot: guide[g] = n up to (S_total >> sh) + 1 (a few entries)
      const int32_t top = (int32_t)(stot >> sh) + 1;
      for (int32_t g = en[c] + 1; g <= top; ++g) guide[g] = (uint32_t)n;
    }
    prev = S;
  }
  *reinterpret_cast<uint4*>(row + j0) = make_uint4(lowv[0], lowv[1], lowv[2], lowv[3]);
  // through LDS (k_offspring_tile's fill): a slot marks the first of its entries with j + 1, a max-scan fills the rest
  // (j increases with the entry), the workgroup stores 8 consecutive entries per thread
  for (int32_t base = gA; base <= gB; base += SORTED_FILL) {                 // block-uniform trip count (1, rarely 2)
    reinterpret_cast<uint4*>(s_mark)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(s_mark)[threadIdx.x + GMX_BLOCK] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int32_t lo = st[c] > base ? st[c] : base;
      if (en[c] >= lo && lo - base < SORTED_FILL) s_mark[lo - base] = (uint32_t)(j0 + c) + 1u;
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
    for (int w = 0; w < GMX_BLOCK / GMX_WAVE - 1; ++w) { const uint32_t v = s_carry[w]; carry = (w < wave && v > carry) ? v : carry; }
    a.x = a.x > carry ? a.x : carry; a.y = a.y > carry ? a.y : carry; a.z = a.z > carry ? a.z : carry; a.w = a.w > carry ? a.w : carry;
    b.x = b.x > carry ? b.x : carry; b.y = b.y > carry ? b.y : carry; b.z = b.z > carry ? b.z : carry; b.w = b.w > carry ? b.w : carry;
    const int32_t g = base + 8 * (int32_t)threadIdx.x;
    if (g + 7 <= gB) {
      rs_u32x4_a4 va, vb;
      va.x = a.x - 1u; va.y = a.y - 1u; va.z = a.z - 1u; va.w = a.w - 1u;
      vb.x = b.x - 1u; vb.y = b.y - 1u; vb.z = b.z - 1u; vb.w = b.w - 1u;
      *reinterpret_cast<rs_u32x4_a4*>(guide + g) = va;
      *reinterpret_cast<rs_u32x4_a4*>(guide + g + 4) = vb;
    } else {
      if (g + 0 <= gB) guide[g + 0] = a.x - 1u;
      if (g + 1 <= gB) guide[g + 1] = a.y - 1u;
      if (g + 2 <= gB) guide[g + 2] = a.z - 1u;
      if (g + 3 <= gB) guide[g + 3] = a.w - 1u;
      if (g + 4 <= gB) guide[g + 4] = b.x - 1u;
      if (g + 5 <= gB) guide[g + 5] = b.y - 1u;
      if (g + 6 <= gB) guide[g + 6] = b.z - 1u;
      if (g + 7 <= gB) guide[g + 7] = b.w - 1u;
    }
    __syncthreads();
  }
}

extern "C" size_t gmx_sorted_uniforms_words(int64_t n) { return n > 0 ? gmx_sorted_layout_of(n).words : 0; }

static int launch_sorted_uniforms(const uint32_t* keys_d, const uint32_t* hkey, int rows, int64_t n, uint32_t* out_d,
                                  int lds_pad, gmx_stream stream) {
  const gmx_sorted_layout L = gmx_sorted_layout_of(n);
  const dim3 grid((unsigned)L.tiles, (unsigned)rows), block(GMX_BLOCK);
  const uint32_t h0 = hkey ? hkey[0] : 0u, h1 = hkey ? hkey[1] : 0u;
  hipLaunchKernelGGL(k_sorted_exp, grid, block, (size_t)lds_pad, (hipStream_t)stream, keys_d, h0, h1, n, L.words, out_d);
  hipLaunchKernelGGL(k_sorted_offsets, dim3(1, (unsigned)rows), block, 0, (hipStream_t)stream, keys_d, h0, h1, n, L.words, out_d);
  hipLaunchKernelGGL(k_sorted_guide, grid, block, (size_t)lds_pad, (hipStream_t)stream, keys_d, h0, h1, n, L.words, out_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_sorted_uniforms(const uint32_t* keys_d, int rows, int64_t n, uint32_t* out_d, int lds_pad,
                                   gmx_stream stream) {
  if (!keys_d || !out_d) return gmx_fail("gmx_sorted_uniforms: null argument%s");
  if (rows < 1 || rows > 65535 || n <= 0 || n >= 0x7fffffffLL - 4096)
    return gmx_fail("gmx_sorted_uniforms: rows / n out of range%s");
  if (lds_pad < 0 || lds_pad > 64 * 1024) return gmx_fail("gmx_sorted_uniforms: lds_pad out of range (<= 64 KB)%s");
  if ((uintptr_t)out_d & 15) return gmx_fail("gmx_sorted_uniforms: out_d must be 16-byte aligned%s");
  return launch_sorted_uniforms(keys_d, nullptr, rows, n, out_d, lds_pad, stream);
}

extern "C" int gmx_resample_sorted(const uint32_t key[2], const float* lw_d, int64_t n, int shift, const float* tile_max_d,
                                   const uint64_t* tile_agg_d, uint32_t* table_d, int table_ready, float* max_d,
                                   uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream) {
  if (resample_shape("gmx_resample_sorted", n, shift)) return 1;
  if (!key || !lw_d || !tile_max_d || !tile_agg_d || !table_d || !max_d || !total_d || !ancestors_d)
    return gmx_fail("gmx_resample_sorted: null argument%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)table_d & 15))
    return gmx_fail("gmx_resample_sorted: lw_d and table_d must be 16-byte aligned%s");
  if (!table_ready && launch_sorted_uniforms(nullptr, key, 1, n, table_d, 0, stream)) return 1;
  return launch_offspring_tile(GMX_RESAMPLE_MULTINOMIAL_SORTED, key, lw_d, n, shift, tile_max_d, tile_agg_d, max_d,
                               total_d, ancestors_d, stream, false, table_d);
}

// ... and past 2048 tiles (n > 2^21: BASELINE config 4's k = 1e7): the resampler reads the tile PREFIXES gmx_tile_prefix
// wrote instead of reducing the statistics table in every workgroup
extern "C" int gmx_resample_sorted_p(const uint32_t key[2], const float* lw_d, int64_t n, int shift, const float* tile_max_d,
                                     const uint64_t* tile_pref_d, uint32_t* table_d, int table_ready, float* max_d,
                                     uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream) {
  if (resample_shape("gmx_resample_sorted_p", n, shift, true)) return 1;
  if (!key || !lw_d || !tile_max_d || !tile_pref_d || !table_d || !max_d || !total_d || !ancestors_d)
    return gmx_fail("gmx_resample_sorted_p: null argument%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)table_d & 15))
    return gmx_fail("gmx_resample_sorted_p: lw_d and table_d must be 16-byte aligned%s");
  if (!table_ready && launch_sorted_uniforms(nullptr, key, 1, n, table_d, 0, stream)) return 1;
  return launch_offspring_tile(GMX_RESAMPLE_MULTINOMIAL_SORTED, key, lw_d, n, shift, tile_max_d, tile_pref_d, max_d,
                               total_d, ancestors_d, stream, true, table_d);
}

extern "C" size_t gmx_tile_prefix_words(int64_t n) { return gmx_tile_prefix_words_(n); }

__global__ void __launch_bounds__(GMX_BLOCK)
k_tile_prefix(const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int n_tiles, uint64_t* __restrict__ pref) {
  __shared__ float lds4[4];
  __shared__ uint64_t lds8[4];
  gmx_tile_prefix_block(tmax, agg, n_tiles, pref, lds4, lds8);
}

// more than RS_MAX_TILES tiles (n > 2^21: BASELINE config 4's 1e7 particles are 9766): ONE workgroup walks the table in
// chunks of 2048 tiles (8 per thread) with a running carry; the same integers as gmx_tile_prefix_block
__global__ void __launch_bounds__(GMX_BLOCK)
k_tile_prefix_big(const float* __restrict__ tmax, const uint64_t* __restrict__ agg, int n_tiles, uint64_t* __restrict__ pref) {
  __shared__ float lds4[4];
  __shared__ uint64_t lds8[4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float M = -gmx_inf();
  for (int t = tid; t < n_tiles; t += GMX_BLOCK) M = gmx_rmax(M, tmax[t]);
  M = block_max(M, lds4);
  const int32_t K = gmx_tile_exp(M);
  uint64_t carry = 0;
  for (int base = 0; base < n_tiles; base += 8 * GMX_BLOCK) {          // block-uniform trip count
    uint64_t P[8];
    uint64_t run = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int t = base + tid * 8 + r;
      const int tc = t < n_tiles ? t : n_tiles - 1;
      const uint64_t G = gmx_tile_scale(agg[tc], gmx_tile_exp(tmax[tc]), K);
      P[r] = run;
      run += (t < n_tiles) ? G : 0ull;
    }
    const uint64_t inc = wave_scan_u64(run);
    __syncthreads();
    if (lane == 63) lds8[wave] = inc;
    __syncthreads();
    uint64_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint64_t v = lds8[w]; tot += v; wave_off += (w < wave) ? v : 0ull; }
    const uint64_t b0 = carry + wave_off + (inc - run);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int t = base + tid * 8 + r;
      if (t < n_tiles) pref[t] = b0 + P[r];
    }
    carry += tot;
  }
  if (tid == 0) {
    pref[n_tiles] = carry;
    pref[n_tiles + 1] = (uint64_t)gmx_f2u(M) | ((uint64_t)(uint32_t)K << 32);
  }
}

extern "C" int gmx_tile_prefix(const float* tile_max_d, const uint64_t* tile_agg_d, int64_t n, uint64_t* tile_pref_d,
                               gmx_stream stream) {
  if (n <= 0 || n > 0x7fffffffLL) return gmx_fail("gmx_tile_prefix: n out of range%s");
  if (!tile_max_d || !tile_agg_d || !tile_pref_d) return gmx_fail("gmx_tile_prefix: null argument%s");
  if ((n + RS_TILE - 1) / RS_TILE > RS_MAX_TILES) {
    hipLaunchKernelGGL(k_tile_prefix_big, dim3(1), dim3(GMX_BLOCK), 0, (hipStream_t)stream, tile_max_d, tile_agg_d,
                       (int)((n + RS_TILE - 1) / RS_TILE), tile_pref_d);
    GMX_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(GMX_BLOCK), 0, (hipStream_t)stream, tile_max_d, tile_agg_d,
                     (int)((n + RS_TILE - 1) / RS_TILE), tile_pref_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_resample_tiles_p(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                                    const float* tile_max_d, const uint64_t* tile_pref_d, float* max_d,
                                    uint64_t* total_d, int32_t* ancestors_d, gmx_stream stream) {
  if (resample_shape("gmx_resample_tiles_p", n, shift, true)) return 1;       // any n < 2^31: a workgroup reads ONE prefix
  if (!key || !lw_d || !tile_max_d || !tile_pref_d || !max_d || !total_d || !ancestors_d)
    return gmx_fail("gmx_resample_tiles_p: null argument%s");
  if (kind != GMX_RESAMPLE_SYSTEMATIC && kind != GMX_RESAMPLE_STRATIFIED)
    return gmx_fail("gmx_resample_tiles_p: kind must be systematic or stratified%s");
  if (n > 0x7fffffffLL) return gmx_fail("gmx_resample_tiles_p: n out of range%s");
  if ((uintptr_t)lw_d & 15) return gmx_fail("gmx_resample_tiles_p: lw_d must be 16-byte aligned%s");
  return launch_offspring_tile(kind, key, lw_d, n, shift, tile_max_d, tile_pref_d, max_d, total_d, ancestors_d,
                               stream, true);
}

// log-weights -> ancestors: gmx_tile_stats + gmx_resample_tiles with the tile stats in the workspace.
// max_partials_d / n_partials are accepted for compatibility and unused: the tile maxima are recomputed.
// THE resampling entry point (include/genmi.h): log-weights -> ancestors for every kind and every n < 2^31, dispatching
// to the staged forms a caller that already holds tile statistics uses directly.
struct resample_ws_layout {
  size_t agg, tmax, pref, table, mnt, cdf, cdfws, mnws, total;
};
static size_t ws_up(size_t x) { return (x + 255) & ~(size_t)255; }
static resample_ws_layout resample_ws_of(int64_t n) {
  resample_ws_layout L;
  const size_t tiles = (size_t)((n + RS_TILE - 1) / RS_TILE);
  size_t o = 0;
  L.agg = o; o = ws_up(o + tiles * 8);
  L.tmax = o; o = ws_up(o + tiles * 4);
  L.pref = o; o = ws_up(o + gmx_tile_prefix_words(n) * 8);
  L.table = o; o = ws_up(o + gmx_sorted_uniforms_words(n) * 4);
  L.mnt = o; o = ws_up(o + (n <= (int64_t)RS_MAX_TILES * RS_TILE ? gmx_multinomial_tiled_workspace(n) : 0));
  L.cdf = o; o = ws_up(o + (size_t)n * 8);
  L.cdfws = o; o = ws_up(o + gmx_weight_cdf_workspace(n));
  L.mnws = o; o = ws_up(o + gmx_multinomial_workspace(n));
  L.total = o;
  return L;
}
extern "C" size_t gmx_resample_workspace(int64_t n) { return n > 0 ? resample_ws_of(n).total : 0; }

extern "C" int gmx_resample(int kind, const uint32_t key[2], const float* lw_d, int64_t n, int shift,
                            const float* max_partials_d, int64_t n_partials, float* max_d,
                            uint64_t* total_d, int32_t* ancestors_d, void* workspace_d, gmx_stream stream) {
  (void)max_partials_d; (void)n_partials;
  if (!workspace_d || !key || !lw_d || !max_d || !total_d || !ancestors_d) return gmx_fail("gmx_resample: null argument%s");
  if ((uintptr_t)workspace_d & 15) return gmx_fail("gmx_resample: workspace_d must be 16-byte aligned%s");
  if (n <= 0 || n > 0x7fffffffLL) return gmx_fail("gmx_resample: n out of range%s");
  const resample_ws_layout L = resample_ws_of(n);
  uint8_t* w = (uint8_t*)workspace_d;
  uint64_t* agg = (uint64_t*)(w + L.agg);
  float* tmax = (float*)(w + L.tmax);
  const bool small = (n + RS_TILE - 1) / RS_TILE <= RS_MAX_TILES;
  if (kind == GMX_RESAMPLE_MULTINOMIAL) {          // iid slot order: the CDF array + the guide-table search
    uint64_t* cdf = (uint64_t*)(w + L.cdf);
    if (gmx_tile_stats(lw_d, n, shift, tmax, agg, stream)) return 1;        // (the tile maxima: gmx_weight_cdf reduces them to the max)
    if (gmx_weight_cdf(lw_d, n, shift, tmax, (n + RS_TILE - 1) / RS_TILE, max_d, cdf, total_d, w + L.cdfws, stream)) return 1;
    return gmx_multinomial(key, cdf, n, total_d, n, ancestors_d, w + L.mnws, stream);
  }
  if (kind != GMX_RESAMPLE_SYSTEMATIC && kind != GMX_RESAMPLE_STRATIFIED && kind != GMX_RESAMPLE_MULTINOMIAL_TILED &&
      kind != GMX_RESAMPLE_MULTINOMIAL_SORTED)
    return gmx_fail("gmx_resample: unknown kind%s");
  if (gmx_tile_stats(lw_d, n, shift, tmax, agg, stream)) return 1;
  if (kind == GMX_RESAMPLE_MULTINOMIAL_TILED) {
    if (!small) return gmx_fail("gmx_resample: GMX_RESAMPLE_MULTINOMIAL_TILED takes n <= 2^21 (use GMX_RESAMPLE_MULTINOMIAL_SORTED)%s");
    return gmx_multinomial_tiled(key, lw_d, n, shift, tmax, agg, nullptr, max_d, total_d, ancestors_d, w + L.mnt, -1, stream);
  }
  if (small) {
    if (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED)
      return gmx_resample_sorted(key, lw_d, n, shift, tmax, agg, (uint32_t*)(w + L.table), 0, max_d, total_d, ancestors_d, stream);
    return gmx_resample_tiles(kind, key, lw_d, n, shift, tmax, agg, max_d, total_d, ancestors_d, stream);
  }
  uint64_t* pref = (uint64_t*)(w + L.pref);           // past 2048 tiles: one prefix pass, then the prefix-reading forms
  if (gmx_tile_prefix(tmax, agg, n, pref, stream)) return 1;
  if (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED)
    return gmx_resample_sorted_p(key, lw_d, n, shift, tmax, pref, (uint32_t*)(w + L.table), 0, max_d, total_d, ancestors_d, stream);
  return gmx_resample_tiles_p(kind, key, lw_d, n, shift, tmax, pref, max_d, total_d, ancestors_d, stream);
}

// ---------------------------------------------------------------------------
// global resampling across ranks (one process per GPU): plan + route
//
// Every rank holds its local inclusive CDF (relative to the GLOBAL max) and,
// after an all-gather, every rank's integer total.  k_shard_plan turns the
// totals into the exact slot boundaries bounds[s] = f(offset_s) — the same
// integer predicate k_offspring evaluates — so every rank knows which slots
// each rank sources, with no host round trip.  k_shard_route is k_offspring
// with a routing step: a slot this rank owns gets the local ancestor index, a
// slot another rank owns gets the ancestor's STATE written into the
// fixed-capacity send block of that rank; slots of this rank whose ancestor is
// remote get an index into the receive area the all-to-all fills.
// ---------------------------------------------------------------------------
// GMX_RESAMPLE_MULTINOMIAL_SORTED across ranks: every rank holds the SAME order-statistics table of the N = n * world
// global slots (gmx_sorted_uniforms with the step's key: integers, identical on every partitioning), and "slots below
// the CDF value c" is a guided look-up in it (the bucket of c * S_total / total, then the exact 128-bit predicate).
__device__ __forceinline__ int64_t shard_sorted_below(const uint32_t* __restrict__ table, int64_t N, uint64_t c,
                                                      uint64_t total) {
  const gmx_sorted_layout L = gmx_sorted_layout_of(N);
  const uint64_t stot = reinterpret_cast<const uint64_t*>(table + L.off_toff)[L.tiles];
  const sorted_ctx X = sorted_ctx_of(table, N, total, stot, table[L.off_sh]);
  const double t = __builtin_fma((double)(uint32_t)(c >> 32), 4294967296.0, (double)(uint32_t)c) * X.ratio;
  uint64_t g = (uint64_t)t >> X.sh;
  g = g < X.gmax ? g : X.gmax;
  return sorted_below_exact(X, c, total, (int64_t)X.guide[g], N);
}
__device__ __forceinline__ int64_t shard_slots_below(int kind, const uint32_t* __restrict__ sorted_tab, gmx_key key,
                                                     uint64_t u0, uint64_t c, uint64_t D, uint64_t total,
                                                     double n_over_total, double eps, int64_t N) {
  if (kind == GMX_RESAMPLE_MULTINOMIAL_SORTED) return shard_sorted_below(sorted_tab, N, c, total);
  return slots_below(kind, key, u0, c, D, total, n_over_total, eps, N);
}

__global__ void k_shard_plan(int kind, uint32_t k0, uint32_t k1, const uint64_t* __restrict__ totals, int rank,
                             int world, int64_t n, int64_t* __restrict__ plan, uint64_t* __restrict__ total_out,
                             const uint32_t* __restrict__ sorted_tab) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  gmx_key key; key.k0 = k0; key.k1 = k1;
  const uint64_t u0 = gmx_bits32(key, 0) >> 9;
  const int64_t N = n * world;
  uint64_t total = 0;
  for (int s = 0; s < world; ++s) total += totals[s];
  const uint64_t D = (uint64_t)N << 23;
  const double n_over_total = total ? (double)N / (double)total : 0.0;
  const double eps = (double)N * 0x1p-44 + 0x1p-40;
  uint64_t off = 0;
  for (int s = 0; s < world; ++s) {
    if (s == rank) plan[GMX_PLAN_OFFSET] = (int64_t)off;
    plan[GMX_PLAN_BOUNDS + s] = total ? shard_slots_below(kind, sorted_tab, key, u0, off, D, total, n_over_total, eps, N) : 0;
    off += totals[s];
  }
  plan[GMX_PLAN_BOUNDS + world] = N;      // no mass at all: the last rank sources every slot
  plan[GMX_PLAN_TOTAL] = (int64_t)total;
  if (total_out) *total_out = total;
}

// Routing of one source's slot run [s, e) (global slot numbers): a slot owned by this rank gets the source's local
// index, a slot owned by rank d gets the source's state in this rank's send block for d.  `step` = 1: the thread
// walks its own run; `step` = 64 with j0 = s + lane: the whole wave walks a LONG run (a heavy particle under skewed
// weights would otherwise keep one lane looping while 63 wait).
struct shard_route_ctx {
  int64_t n, S, base, cap;
  int rank;
  int32_t* next_idx;
  uint32_t* send;
};
__device__ __forceinline__ void shard_route_run(const shard_route_ctx& R, int64_t j0, int64_t e, int64_t step, int32_t i,
                                                uint32_t v, bool& overflow) {
  if (j0 >= e) return;
  int64_t d = j0 / R.n;                     // owner of slot j; advances as j crosses a block of n
  int64_t d_end = (d + 1) * R.n;
  int64_t first = R.S > d * R.n ? R.S : d * R.n;      // first slot this rank sends to d
  for (int64_t j = j0; j < e; j += step) {
    while (j >= d_end) { ++d; d_end += R.n; first = d * R.n; }
    if (d == R.rank) R.next_idx[j - R.base] = i;
    else {
      const int64_t k = j - first;
      if (k < R.cap) R.send[d * R.cap + k] = v; else overflow = true;
    }
  }
}
#define SHARD_OWN_RUN 8
// every lane of the wave calls this (has = this lane's source is real and owns slots)
__device__ __forceinline__ void shard_route_source(const shard_route_ctx& R, bool has, int64_t s, int64_t e, int32_t i,
                                                   uint32_t v, int lane, bool& overflow) {
  const int64_t cnt = has ? e - s : 0;
  if (cnt > 0 && cnt <= SHARD_OWN_RUN) shard_route_run(R, s, e, 1, i, v, overflow);
  uint64_t heavy = __ballot(cnt > SHARD_OWN_RUN);
  while (heavy) {                           // wave-uniform
    const int l = __ffsll((unsigned long long)heavy) - 1;
    heavy &= heavy - 1ull;
    const int64_t s_l = shard_readlane64(s, l), e_l = shard_readlane64(e, l);
    const int32_t i_l = __builtin_amdgcn_readlane(i, l);
    const uint32_t v_l = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
    shard_route_run(R, s_l + lane, e_l, GMX_WAVE, i_l, v_l, overflow);
  }
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_shard_route(int kind, uint32_t k0, uint32_t k1, int64_t* __restrict__ plan, const uint64_t* __restrict__ cdf,
              int rank, int world, int64_t n, int64_t cap, const uint32_t* __restrict__ state,
              uint32_t* __restrict__ send, int32_t* __restrict__ next_idx, const uint32_t* __restrict__ sorted_tab) {
  const int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  const bool in_range = i < n;
  const uint64_t total = (uint64_t)plan[GMX_PLAN_TOTAL];
  const uint64_t cdf_offset = (uint64_t)plan[GMX_PLAN_OFFSET];
  const int64_t* bounds = plan + GMX_PLAN_BOUNDS;
  const int64_t N = n * world, base = (int64_t)rank * n;
  bool overflow = false;
  // (b) my slot base+i: if its ancestor is on rank s != rank it arrives at recv[s*cap + k]
  if (in_range) {
    const int64_t jj = base + i;
    int s = 0;
    while (s + 1 < world && bounds[s + 1] <= jj) ++s;
    if (s != rank) {
      const int64_t first = bounds[s] > base ? bounds[s] : base;
      const int64_t k = jj - first;
      if (k < cap) next_idx[i] = (int32_t)(n + (int64_t)s * cap + k);
      else { next_idx[i] = 0; overflow = true; }
    }
  }
  // (a) the slots source i owns: [f(cdf_{i-1}), f(cdf_i))  (as k_offspring)
  gmx_key key; key.k0 = k0; key.k1 = k1;
  const uint64_t u0 = gmx_bits32(key, 0) >> 9;
  const uint64_t D = (uint64_t)N << 23;
  int64_t s_lo, e;
  if (total == 0) {                         // only the globally last particle has offspring
    const bool last = in_range && rank == world - 1 && i == n - 1;
    s_lo = 0; e = last ? N : 0;
  } else {
    const double n_over_total = (double)N / (double)total;
    const double eps = (double)N * 0x1p-44 + 0x1p-40;
    const uint64_t c_hi = in_range ? cdf[i] + cdf_offset : total;
    e = shard_slots_below(kind, sorted_tab, key, u0, c_hi, D, total, n_over_total, eps, N);
    const int lane = threadIdx.x & 63;
    uint32_t e_l = (uint32_t)e, e_h = (uint32_t)((uint64_t)e >> 32);
    e_l = __shfl_up(e_l, 1, GMX_WAVE); e_h = __shfl_up(e_h, 1, GMX_WAVE);
    s_lo = (int64_t)(((uint64_t)e_h << 32) | e_l);
    if (lane == 0) {
      const uint64_t c_lo = (i == 0 || !in_range) ? cdf_offset : cdf[i - 1] + cdf_offset;
      s_lo = shard_slots_below(kind, sorted_tab, key, u0, c_lo, D, total, n_over_total, eps, N);
    }
  }
  {
    const bool has = in_range && s_lo < e;
    const uint32_t v = has ? state[i] : 0u;
    shard_route_ctx R; R.n = n; R.S = bounds[rank]; R.base = base; R.cap = cap; R.rank = rank; R.next_idx = next_idx; R.send = send;
    shard_route_source(R, has, s_lo, e, (int32_t)i, v, (int)(threadIdx.x & 63), overflow);
  }
  if (overflow) plan[GMX_PLAN_OVERFLOW] = 1;
}

// plan + route in ONE launch (what the sharded sweep issues): every block derives the slot boundaries
// from the all-gathered totals itself (world <= 64 evaluations of f) and handles 4 consecutive
// sources per thread with two 16-byte CDF loads, like k_offspring_local.
// where a source's local CDF value comes from: the array gmx_weight_cdf wrote (TILES = false), or — like
// k_offspring_tile — rebuilt in registers from the log-weights and this rank's tile statistics (TILES = true:
// a block is one 1024-particle tile; tmax / agg are this rank's, *max_g the GLOBAL max log-weight)
struct shard_tiles { const float* lw; const float* tmax; const uint64_t* agg; const float* max_g; float scale; int n_tiles; };

template <bool TILES>
__global__ void __launch_bounds__(GMX_BLOCK)
k_shard_step(int kind, uint32_t k0, uint32_t k1, const uint64_t* __restrict__ totals, int64_t* __restrict__ plan,
             uint64_t* __restrict__ total_out, const uint64_t* __restrict__ cdf, const shard_tiles TS, int rank,
             int world, int64_t n, int64_t cap, const uint32_t* __restrict__ state, uint32_t* __restrict__ send,
             int32_t* __restrict__ next_idx) {
  __shared__ int64_t s_bounds[SHARD_MAX_WORLD + 1];
  __shared__ uint64_t s_tot[2];            // global total, this rank's CDF offset
  __shared__ uint64_t s_below[4], s_scan[4];
  gmx_key key; key.k0 = k0; key.k1 = k1;
  const uint64_t u0 = gmx_bits32(key, 0) >> 9;
  const int64_t N = n * world, base = (int64_t)rank * n;
  const uint64_t D = (uint64_t)N << 23;
  if (threadIdx.x == 0) {
    uint64_t total = 0;
    for (int s = 0; s < world; ++s) total += totals[s];
    const double not_ = total ? (double)N / (double)total : 0.0;
    const double eps_ = (double)N * 0x1p-44 + 0x1p-40;
    uint64_t off = 0, mine = 0;
    for (int s = 0; s < world; ++s) {
      if (s == rank) mine = off;
      s_bounds[s] = total ? slots_below(kind, key, u0, off, D, total, not_, eps_, N) : 0;
      off += totals[s];
    }
    s_bounds[world] = N;
    s_tot[0] = total; s_tot[1] = mine;
    if (blockIdx.x == 0) {               // published for inspection / tests; the overflow word is left alone
      plan[GMX_PLAN_TOTAL] = (int64_t)total; plan[GMX_PLAN_OFFSET] = (int64_t)mine;
      for (int s = 0; s <= world; ++s) plan[GMX_PLAN_BOUNDS + s] = s_bounds[s];
      if (total_out) *total_out = total;
    }
  }
  __syncthreads();
  const uint64_t total = s_tot[0], cdf_offset = s_tot[1];
  const int64_t i0 = ((int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x) * 4;
  bool overflow = false;
  // (b) my slots base+i0..+3: an ancestor on rank s != rank arrives at recv[s*cap + k]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t i = i0 + c;
    if (i < n) {
      const int64_t jj = base + i;
      int s = 0;
      while (s + 1 < world && s_bounds[s + 1] <= jj) ++s;
      if (s != rank) {
        const int64_t first = s_bounds[s] > base ? s_bounds[s] : base;
        const int64_t k = jj - first;
        if (k < cap) next_idx[i] = (int32_t)(n + (int64_t)s * cap + k);
        else { next_idx[i] = 0; overflow = true; }
      }
    }
  }
  // (a) the slots each of my 4 sources owns
  int64_t e[4], s_lo;
  const int lane = threadIdx.x & 63;
  if (total == 0) {                        // only the globally last particle has offspring
#pragma unroll
    for (int c = 0; c < 4; ++c) e[c] = (rank == world - 1 && i0 + c == n - 1) ? N : 0;
    s_lo = 0;
  } else {
    uint64_t loc[4];
    uint64_t loc_prev;
    if (TILES) {
      // this rank's CDF at my 4 sources = (mass of this rank's earlier tiles) + (tile-local sums >> (K - k_b)),
      // with the tile-local sums rebuilt from the log-weights (k_offspring_tile's scheme)
      const int wave = threadIdx.x >> 6, my_tile = (int)blockIdx.x;
      const int32_t K = gmx_tile_exp(*TS.max_g);
      const int32_t k_b = gmx_tile_exp(TS.tmax[my_tile]);
      const float ref_b = gmx_tile_ref(k_b);
      float x[4];
      if (i0 + 4 <= n) {
        float4 v = *reinterpret_cast<const float4*>(TS.lw + i0);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = (i0 + c < n) ? TS.lw[i0 + c] : -gmx_inf();
      }
      uint64_t below = 0;
      for (int t = (int)threadIdx.x; t < my_tile; t += GMX_BLOCK)
        below += gmx_tile_scale(TS.agg[t], gmx_tile_exp(TS.tmax[t]), K);
      uint64_t q[4], run = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        run += (i0 + c < n) ? weight_fixed(x[c], ref_b, TS.scale) : 0ull;
        q[c] = run;
      }
      const uint64_t inc = wave_scan_u64(run);
      below = wave_sum_u64(below);
      if (lane == 0) s_below[wave] = below;
      if (lane == 63) s_scan[wave] = inc;
      __syncthreads();                       // uniform: `total` is block-uniform
      const uint64_t prefix = (s_below[0] + s_below[1]) + (s_below[2] + s_below[3]);
      uint64_t wave_off = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? s_scan[w] : 0ull;
      const uint64_t excl = wave_off + (inc - run);
#pragma unroll
      for (int c = 0; c < 4; ++c) loc[c] = prefix + gmx_tile_scale(excl + q[c], k_b, K);
      loc_prev = prefix + gmx_tile_scale(excl, k_b, K);
    } else {
      if (i0 + 4 <= n) {
        ulonglong2 a = reinterpret_cast<const ulonglong2*>(cdf + i0)[0];
        ulonglong2 b = reinterpret_cast<const ulonglong2*>(cdf + i0)[1];
        loc[0] = a.x; loc[1] = a.y; loc[2] = b.x; loc[3] = b.y;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) loc[c] = (i0 + c < n) ? cdf[i0 + c] : 0ull;
      }
      loc_prev = (lane == 0 && i0 > 0 && i0 < n) ? cdf[i0 - 1] : 0ull;
    }
    const double n_over_total = (double)N / (double)total;
    const double eps = (double)N * 0x1p-44 + 0x1p-40;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint64_t c_hi = (i0 + c < n) ? loc[c] + cdf_offset : total;
      e[c] = slots_below(kind, key, u0, c_hi, D, total, n_over_total, eps, N);
    }
    uint32_t e_l = (uint32_t)e[3], e_h = (uint32_t)((uint64_t)e[3] >> 32);
    e_l = __shfl_up(e_l, 1, GMX_WAVE); e_h = __shfl_up(e_h, 1, GMX_WAVE);
    s_lo = (int64_t)(((uint64_t)e_h << 32) | e_l);
    if (lane == 0) s_lo = slots_below(kind, key, u0, loc_prev + cdf_offset, D, total, n_over_total, eps, N);
  }
  shard_route_ctx R; R.n = n; R.S = s_bounds[rank]; R.base = base; R.cap = cap; R.rank = rank; R.next_idx = next_idx; R.send = send;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t i = i0 + c;
    const bool has = i < n && s_lo < e[c];
    const uint32_t v = has ? state[i] : 0u;
    shard_route_source(R, has, s_lo, e[c], (int32_t)i, v, lane, overflow);
    if (i < n) s_lo = e[c];
  }
  if (overflow) plan[GMX_PLAN_OVERFLOW] = 1;
}

extern "C" size_t gmx_shard_plan_words(int world) { return (size_t)(GMX_PLAN_BOUNDS + world + 1); }

static int shard_check(const char* who, int kind, const void* key, int rank, int world, int64_t n) {
  if (!key) return gmx_fail("%s: null key", who);
  if (kind != GMX_RESAMPLE_SYSTEMATIC && kind != GMX_RESAMPLE_STRATIFIED)
    return gmx_fail("%s: systematic / stratified only (ordered slot thresholds)", who);
  if (world < 1 || world > 1024 || rank < 0 || rank >= world) return gmx_fail("%s: rank / world out of range", who);
  if (n <= 0 || n > 0x7fffffffLL || n * world >= (1LL << 40)) return gmx_fail("%s: n_per_rank out of range", who);
  return 0;
}

extern "C" int gmx_shard_plan(int kind, const uint32_t key[2], const uint64_t* totals_d, int rank, int world,
                              int64_t n_per_rank, int64_t* plan_d, uint64_t* total_out_d, gmx_stream stream) {
  if (shard_check("gmx_shard_plan", kind, key, rank, world, n_per_rank)) return 1;
  if (!totals_d || !plan_d) return gmx_fail("gmx_shard_plan: null argument%s");
  hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, kind, key[0], key[1], totals_d, rank,
                     world, n_per_rank, plan_d, total_out_d, (const uint32_t*)nullptr);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_shard_route(int kind, const uint32_t key[2], int64_t* plan_d, const uint64_t* cdf_d, int rank,
                               int world, int64_t n_per_rank, int64_t capacity, const void* state_d, void* send_d,
                               int32_t* next_idx_d, gmx_stream stream) {
  if (shard_check("gmx_shard_route", kind, key, rank, world, n_per_rank)) return 1;
  if (!plan_d || !cdf_d || !state_d || !send_d || !next_idx_d) return gmx_fail("gmx_shard_route: null argument%s");
  if (capacity < 1 || capacity > n_per_rank) return gmx_fail("gmx_shard_route: capacity must be in [1, n_per_rank]%s");
  if (n_per_rank + (int64_t)world * capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_route: extended state index exceeds int32%s");
  hipLaunchKernelGGL(k_shard_route, grid_for(n_per_rank), dim3(GMX_BLOCK), 0, (hipStream_t)stream, kind, key[0],
                     key[1], plan_d, cdf_d, rank, world, n_per_rank, capacity, (const uint32_t*)state_d,
                     (uint32_t*)send_d, next_idx_d, (const uint32_t*)nullptr);
  GMX_HIP(hipGetLastError());
  return 0;
}

// gmx_shard_step for GMX_RESAMPLE_MULTINOMIAL_SORTED (include/genmi.h): plan + route against the order-statistics table
// of the n_per_rank * world global slots (table_d: gmx_sorted_uniforms of the step's key, the same on every rank).
extern "C" int gmx_shard_step_sorted(const uint32_t* table_d, const uint64_t* totals_d, int64_t* plan_d,
                                     uint64_t* total_out_d, const uint64_t* cdf_d, int rank, int world,
                                     int64_t n_per_rank, int64_t capacity, const void* state_d, void* send_d,
                                     int32_t* next_idx_d, gmx_stream stream) {
  if (!table_d || !totals_d || !plan_d || !cdf_d || !state_d || !send_d || !next_idx_d)
    return gmx_fail("gmx_shard_step_sorted: null argument%s");
  if (world < 1 || world > 1024 || rank < 0 || rank >= world) return gmx_fail("gmx_shard_step_sorted: rank / world out of range%s");
  if (n_per_rank <= 0 || n_per_rank * world > 0x7fffffffLL) return gmx_fail("gmx_shard_step_sorted: n_per_rank * world out of range (< 2^31)%s");
  if (capacity < 1 || capacity > n_per_rank) return gmx_fail("gmx_shard_step_sorted: capacity must be in [1, n_per_rank]%s");
  if (n_per_rank + (int64_t)world * capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_step_sorted: extended state index exceeds int32%s");
  if ((uintptr_t)table_d & 15) return gmx_fail("gmx_shard_step_sorted: table_d must be 16-byte aligned%s");
  const int kind = GMX_RESAMPLE_MULTINOMIAL_SORTED;
  hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, kind, 0u, 0u, totals_d, rank, world,
                     n_per_rank, plan_d, total_out_d, table_d);
  GMX_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_shard_route, grid_for(n_per_rank), dim3(GMX_BLOCK), 0, (hipStream_t)stream, kind, 0u, 0u, plan_d,
                     cdf_d, rank, world, n_per_rank, capacity, (const uint32_t*)state_d, (uint32_t*)send_d, next_idx_d,
                     table_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_shard_step(int kind, const uint32_t key[2], const uint64_t* totals_d, int64_t* plan_d,
                              uint64_t* total_out_d, const uint64_t* cdf_d, int rank, int world, int64_t n_per_rank,
                              int64_t capacity, const void* state_d, void* send_d, int32_t* next_idx_d,
                              gmx_stream stream) {
  if (shard_check("gmx_shard_step", kind, key, rank, world, n_per_rank)) return 1;
  if (!totals_d || !plan_d || !cdf_d || !state_d || !send_d || !next_idx_d)
    return gmx_fail("gmx_shard_step: null argument%s");
  if (capacity < 1 || capacity > n_per_rank) return gmx_fail("gmx_shard_step: capacity must be in [1, n_per_rank]%s");
  if (n_per_rank + (int64_t)world * capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_step: extended state index exceeds int32%s");
  if (world > SHARD_MAX_WORLD) {          // large worlds: the two-launch form
    if (gmx_shard_plan(kind, key, totals_d, rank, world, n_per_rank, plan_d, total_out_d, stream)) return 1;
    return gmx_shard_route(kind, key, plan_d, cdf_d, rank, world, n_per_rank, capacity, state_d, send_d, next_idx_d,
                           stream);
  }
  if ((uintptr_t)cdf_d & 15) return gmx_fail("gmx_shard_step: cdf_d must be 16-byte aligned%s");
  shard_tiles none = {nullptr, nullptr, nullptr, nullptr, 0.0f, 0};
  hipLaunchKernelGGL((k_shard_step<false>), grid_for((n_per_rank + 3) / 4), dim3(GMX_BLOCK), 0, (hipStream_t)stream,
                     kind, key[0], key[1], totals_d, plan_d, total_out_d, cdf_d, none, rank, world, n_per_rank,
                     capacity, (const uint32_t*)state_d, (uint32_t*)send_d, next_idx_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---- the same step from tile statistics: no local CDF array, no all-reduce of the max ----
// Every rank all-gathers its tile statistics (gmx_shard_stats_bytes per rank: agg[tiles_pad] u64 then
// tmax[tiles_pad] f32, tiles_pad = tiles rounded up to even) — ONE collective carries what the max all-reduce
// and the totals all-gather carried.  k_shard_totals (one block) turns the gathered table into the global max
// and every rank's integer total; k_shard_step<true> rebuilds this rank's CDF per tile in registers.
extern "C" size_t gmx_shard_stats_bytes(int64_t n_per_rank) {
  int64_t tiles = (n_per_rank + RS_TILE - 1) / RS_TILE;
  tiles += tiles & 1;
  return (size_t)tiles * 12;
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_shard_totals(const uint8_t* __restrict__ stats_all, int world, int n_tiles, size_t stride,
               uint64_t* __restrict__ totals, float* __restrict__ max_out) {
  __shared__ float lds4[4];
  __shared__ uint64_t s_sum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_pad = n_tiles + (n_tiles & 1);
  float m = -gmx_inf();
  for (int r = 0; r < world; ++r) {
    const float* tmax = reinterpret_cast<const float*>(stats_all + (size_t)r * stride + (size_t)tiles_pad * 8);
    for (int t = (int)threadIdx.x; t < n_tiles; t += GMX_BLOCK) m = gmx_rmax(m, tmax[t]);
  }
  const float M = block_max(m, lds4);
  const int32_t K = gmx_tile_exp(M);
  if (threadIdx.x == 0) *max_out = M;
  for (int r = 0; r < world; ++r) {
    const uint64_t* agg = reinterpret_cast<const uint64_t*>(stats_all + (size_t)r * stride);
    const float* tmax = reinterpret_cast<const float*>(stats_all + (size_t)r * stride + (size_t)tiles_pad * 8);
    uint64_t sum = 0;
    for (int t = (int)threadIdx.x; t < n_tiles; t += GMX_BLOCK) sum += gmx_tile_scale(agg[t], gmx_tile_exp(tmax[t]), K);
    sum = wave_sum_u64(sum);
    __syncthreads();
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) totals[r] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
  }
}

extern "C" int gmx_shard_totals(const void* stats_all_d, int world, int64_t n_per_rank, uint64_t* totals_d,
                                float* max_d, gmx_stream stream) {
  if (!stats_all_d || !totals_d || !max_d) return gmx_fail("gmx_shard_totals: null argument%s");
  if (world < 1 || world > 1024) return gmx_fail("gmx_shard_totals: world out of range%s");
  const int64_t tiles = (n_per_rank + RS_TILE - 1) / RS_TILE;
  // (one workgroup strides over the whole gathered table: any number of tiles — config 4's 1e7 particles on ONE rank are 9766)
  if (n_per_rank <= 0 || n_per_rank > 0x7fffffffLL) return gmx_fail("gmx_shard_totals: n_per_rank out of range (< 2^31)%s");
  if ((uintptr_t)stats_all_d & 7) return gmx_fail("gmx_shard_totals: stats_all_d must be 8-byte aligned%s");
  hipLaunchKernelGGL(k_shard_totals, dim3(1), dim3(GMX_BLOCK), 0, (hipStream_t)stream, (const uint8_t*)stats_all_d,
                     world, (int)tiles, gmx_shard_stats_bytes(n_per_rank), totals_d, max_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_shard_step_tiles(int kind, const uint32_t key[2], const uint64_t* totals_d, int64_t* plan_d,
                                    uint64_t* total_out_d, const float* lw_d, const void* stats_own_d,
                                    const float* max_d, int shift, int rank, int world, int64_t n_per_rank,
                                    int64_t capacity, const void* state_d, void* send_d, int32_t* next_idx_d,
                                    gmx_stream stream) {
  if (shard_check("gmx_shard_step_tiles", kind, key, rank, world, n_per_rank)) return 1;
  if (!totals_d || !plan_d || !lw_d || !stats_own_d || !max_d || !state_d || !send_d || !next_idx_d)
    return gmx_fail("gmx_shard_step_tiles: null argument%s");
  if (capacity < 1 || capacity > n_per_rank)
    return gmx_fail("gmx_shard_step_tiles: capacity must be in [1, n_per_rank]%s");
  if (n_per_rank + (int64_t)world * capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_step_tiles: extended state index exceeds int32%s");
  if (world > SHARD_MAX_WORLD) return gmx_fail("gmx_shard_step_tiles: world <= 64 (use gmx_weight_cdf + gmx_shard_step)%s");
  if (shift < 1 || shift > 62) return gmx_fail("gmx_shard_step_tiles: shift out of range%s");
  const int64_t tiles = (n_per_rank + RS_TILE - 1) / RS_TILE;
  // (k_shard_step<true> sums the mass of its rank's EARLIER tiles by striding over them — no table in registers or LDS —
  //  so the per-rank size is not bounded by RS_MAX_TILES; the fused / peer forms below hold the table and are)
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)stats_own_d & 7))
    return gmx_fail("gmx_shard_step_tiles: lw_d must be 16-byte and stats_own_d 8-byte aligned%s");
  const int64_t tiles_pad = tiles + (tiles & 1);
  shard_tiles ts;
  ts.lw = lw_d;
  ts.agg = (const uint64_t*)stats_own_d;
  ts.tmax = (const float*)((const uint8_t*)stats_own_d + (size_t)tiles_pad * 8);
  ts.max_g = max_d; ts.scale = gmx_pow2i(shift); ts.n_tiles = (int)tiles;
  hipLaunchKernelGGL((k_shard_step<true>), dim3((unsigned)tiles), dim3(GMX_BLOCK), 0, (hipStream_t)stream, kind, key[0],
                     key[1], totals_d, plan_d, total_out_d, (const uint64_t*)nullptr, ts, rank, world, n_per_rank,
                     capacity, (const uint32_t*)state_d, (uint32_t*)send_d, next_idx_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// The fused step on k_offspring_tile's scheme (what gmx_shard_step_fused launches).
// A workgroup is one 1024-particle tile of this rank's shard.  It derives M, K, every rank's total and the slot bounds
// from the gathered statistics table (two barriers whatever the world size: partial sums per (rank, wave), then
// thread s evaluates rank s's bound), rebuilds its tile's CDF in registers, gets its sources' slot runs from the
// straight-line f64 estimate (+ the cold exact predicate), and ROUTES THROUGH LDS: every source with a slot marks its
// first one, a max-scan fills the tile's contiguous GLOBAL slot range [T0, T1), and each thread handles 8 consecutive
// slots — a slot this rank owns gets the source's local index (two 16-byte stores on the common path), a slot rank d
// owns gets the source's state in send block d.  Same plan, send buffer and indices as k_shard_step, at a cost that
// does not depend on the weights (k_shard_step walks each source's run per thread, with a 64-bit division per source).
// SMALL: the whole gathered table has <= 1024 rows and <= 8 ranks (every single-node strong-scaling split of 1e6
// particles: 8 x 123, 4 x 245, 2 x 489, 1 x 977): a thread holds its <= 4 rows in registers, so ALL loads of the launch
// are issued before anything waits and the table is read once — the generic form walks it twice, the second time
// behind a barrier.
// PEER (include/genmi.h "Fused peer exchange"): no collective launch on either side of this kernel.  The other ranks'
// statistics are granules in this rank's landing block, polled until they carry the step's tag (the rows of this rank
// itself come from the local table `stats_all`, which then holds ONE block: stride = 0 is never used for r != rank);
// a slot another rank owns gets the ancestor's state as a granule put into THAT rank's landing block; and at the end
// every slot of this rank whose ancestor is remote waits for its granule and stores the value in the local tail.
// (shard_peer, the body of the fill kernel: csrc/gmx_shard_fill.h — shared with the specialised site programs that route first)
template <int kind, bool SMALL, bool PEER>
__global__ void __launch_bounds__(GMX_BLOCK)
k_shard_step_fill(uint32_t k0, uint32_t k1, uint32_t u0_host, const float* __restrict__ lw,
                  const uint8_t* __restrict__ stats_all, size_t stride, int n_tiles, float scale, int rank, int world,
                  int32_t n, int32_t cap, int64_t* __restrict__ plan, uint64_t* __restrict__ total_out,
                  float* __restrict__ max_out, const uint32_t* __restrict__ state, uint32_t* __restrict__ send,
                  int32_t* __restrict__ next_idx, const shard_peer_args Pa) {
  GMX_SETPRIO
  shard_peer P;
  P.land = Pa.land; P.tag_base = Pa.tag_base; P.status = Pa.status; P.step = Pa.step; P.leaves = Pa.leaves;
  P.state = Pa.state; P.tail = Pa.tail;
  gmx_shard_fill_body<kind, SMALL, PEER, false>(k0, k1, u0_host, lw, stats_all, stride, n_tiles, scale, rank, world, n, cap, plan,
                                                total_out, max_out, state, send, next_idx, P, 0u);
}

// gmx_shard_totals + gmx_shard_step_tiles as ONE launch: straight from the all-gathered statistics table.
extern "C" int gmx_shard_step_fused(int kind, const uint32_t key[2], const void* stats_all_d, int64_t* plan_d,
                                    uint64_t* total_out_d, const float* lw_d, float* max_out_d, int shift, int rank,
                                    int world, int64_t n_per_rank, int64_t capacity, const void* state_d, void* send_d,
                                    int32_t* next_idx_d, gmx_stream stream) {
  if (shard_check("gmx_shard_step_fused", kind, key, rank, world, n_per_rank)) return 1;
  if (!stats_all_d || !plan_d || !lw_d || !max_out_d || !state_d || !send_d || !next_idx_d)
    return gmx_fail("gmx_shard_step_fused: null argument%s");
  if (capacity < 1 || capacity > n_per_rank)
    return gmx_fail("gmx_shard_step_fused: capacity must be in [1, n_per_rank]%s");
  if (n_per_rank + (int64_t)world * capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_step_fused: extended state index exceeds int32%s");
  if (world > SHARD_MAX_WORLD) return gmx_fail("gmx_shard_step_fused: world <= 64%s");
  if (shift < 1 || shift > 62) return gmx_fail("gmx_shard_step_fused: shift out of range%s");
  const int64_t tiles = (n_per_rank + RS_TILE - 1) / RS_TILE;
  if (tiles > RS_MAX_TILES) return gmx_fail("gmx_shard_step_fused: n_per_rank too large (<= 2^21)%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)stats_all_d & 7))
    return gmx_fail("gmx_shard_step_fused: lw_d must be 16-byte and stats_all_d 8-byte aligned%s");
  const size_t stride = gmx_shard_stats_bytes(n_per_rank);
  const float scale = gmx_pow2i(shift);
  // n_per_rank <= 2^21 and world <= 64: slots are 32-bit integers
  uint32_t b0, b1;
  gmx_threefry2x32(key[0], key[1], 0u, 0u, &b0, &b1);           // bits32(key, 0) on the host
  const uint32_t u0 = (b0 ^ b1) >> 9;
  const bool small = world <= 8 && (int64_t)world * tiles <= 4 * GMX_BLOCK;
#define GMX_LAUNCH_SF2(KIND, SM)                                                                                      \
  hipLaunchKernelGGL((k_shard_step_fill<KIND, SM, false>), dim3((unsigned)tiles), dim3(GMX_BLOCK), 0, (hipStream_t)stream, key[0],  \
                     key[1], u0, lw_d, (const uint8_t*)stats_all_d, stride, (int)tiles, scale, rank, world,           \
                     (int32_t)n_per_rank, (int32_t)capacity, plan_d, total_out_d, max_out_d, (const uint32_t*)state_d,   \
                     (uint32_t*)send_d, next_idx_d, shard_peer_args())
#define GMX_LAUNCH_SF(KIND) do { if (small) GMX_LAUNCH_SF2(KIND, true); else GMX_LAUNCH_SF2(KIND, false); } while (0)
  if (kind == GMX_RESAMPLE_SYSTEMATIC) GMX_LAUNCH_SF(GMX_RESAMPLE_SYSTEMATIC); else GMX_LAUNCH_SF(GMX_RESAMPLE_STRATIFIED);
#undef GMX_LAUNCH_SF
#undef GMX_LAUNCH_SF2
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// fused peer exchange (include/genmi.h "Fused peer exchange"): the entry points around k_shard_step_fill<.., PEER>
// ---------------------------------------------------------------------------
extern "C" size_t gmx_peer_landing_bytes(int world, int64_t n_per_rank, int64_t capacity, int leaves) {
  if (world < 1 || n_per_rank < 1 || capacity < 1 || leaves < 1) return 0;
  const int tiles = (int)((n_per_rank + RS_TILE - 1) / RS_TILE);
  return (gmx_peer_stats_words(world, tiles) + gmx_peer_state_words(world, capacity, leaves)) * sizeof(uint64_t);
}

__global__ void k_peer_bump(uint32_t* tag_base, uint32_t T) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *tag_base += T;
}
extern "C" int gmx_peer_bump(uint32_t* tag_base_d, int32_t T, gmx_stream stream) {
  if (!tag_base_d || T < 1) return gmx_fail("gmx_peer_bump: bad argument%s");
  hipLaunchKernelGGL(k_peer_bump, dim3(1), dim3(64), 0, (hipStream_t)stream, tag_base_d, (uint32_t)T);
  GMX_HIP(hipGetLastError());
  return 0;
}

__global__ void k_sweep_verdict(const int64_t* overflow, const uint64_t* w0, const uint64_t* w1, const uint64_t* w2,
                                const uint64_t* w3, int64_t* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool bad = (w0 && *w0) || (w1 && *w1) || (w2 && *w2) || (w3 && *w3);
    *out = bad ? (int64_t)2 : *overflow;
  }
}
extern "C" int gmx_sweep_verdict(const int64_t* overflow_d, const uint64_t* const* status_h, int32_t n_status,
                                 int64_t* verdict_d, gmx_stream stream) {
  if (!overflow_d || !verdict_d || n_status < 0 || n_status > 4 || (n_status && !status_h))
    return gmx_fail("gmx_sweep_verdict: bad argument (at most 4 status words)%s");
  const uint64_t* w[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < n_status; ++k) w[k] = status_h[k];
  hipLaunchKernelGGL(k_sweep_verdict, dim3(1), dim3(64), 0, (hipStream_t)stream, overflow_d, w[0], w[1], w[2], w[3], verdict_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

static int peer_check(const char* who, const gmx_peer& P, int64_t n_per_rank) {
  if (!P.land_d || !P.tag_base_d || !P.status_d) return gmx_fail("%s: peer has a null pointer", who);
  if (P.world < 1 || P.world > SHARD_MAX_WORLD || P.rank < 0 || P.rank >= P.world) return gmx_fail("%s: peer rank / world out of range (world <= 64)", who);
  if (P.step < 0) return gmx_fail("%s: peer.step is negative", who);
  const int64_t tiles = (n_per_rank + RS_TILE - 1) / RS_TILE;
  if (n_per_rank <= 0 || tiles > RS_MAX_TILES || P.tiles != (int32_t)tiles) return gmx_fail("%s: peer.tiles must be ceil(n_per_rank / 1024) <= 2048", who);
  if (P.capacity < 1 || P.capacity > n_per_rank) return gmx_fail("%s: peer.capacity must be in [1, n_per_rank]", who);
  if (P.leaves < 1 || P.leaves > GMX_PEER_MAX_LEAVES) return gmx_fail("%s: peer.leaves must be in [1, GMX_PEER_MAX_LEAVES = 32]", who);
  return 0;
}

// the statistics block of a launch that has no epilogue put (gmx_tile_stats, an interpreted site program): thread b
// puts tile b's row to every other rank
__global__ void __launch_bounds__(GMX_BLOCK)
k_peer_put_stats(const uint8_t* __restrict__ own, int tiles, int tiles_pad, uint64_t* const* __restrict__ land,
                 const uint32_t* __restrict__ tag_base, int step, int rank, int world) {
  const int b = (int)(blockIdx.x * GMX_BLOCK + threadIdx.x);
  if (b >= tiles) return;
  const uint32_t tag = *tag_base + (uint32_t)step;
  const uint64_t a = reinterpret_cast<const uint64_t*>(own)[b];
  const float m = reinterpret_cast<const float*>(own + (size_t)tiles_pad * 8)[b];
  for (int d = 0; d < world; ++d)
    if (d != rank) gmx_peer_put_tile(land[d], tag, world, tiles, rank, b, a, m);
}
extern "C" int gmx_peer_put_stats(const void* stats_own_d, gmx_peer P, int64_t n_per_rank, gmx_stream stream) {
  if (!stats_own_d) return gmx_fail("gmx_peer_put_stats: null argument%s");
  if (peer_check("gmx_peer_put_stats", P, n_per_rank)) return 1;
  if (P.world == 1) return 0;
  const int tiles = P.tiles, tiles_pad = tiles + (tiles & 1);
  hipLaunchKernelGGL(k_peer_put_stats, dim3((unsigned)((tiles + GMX_BLOCK - 1) / GMX_BLOCK)), dim3(GMX_BLOCK), 0,
                     (hipStream_t)stream, (const uint8_t*)stats_own_d, tiles, tiles_pad, (uint64_t* const*)P.land_d,
                     P.tag_base_d, P.step, P.rank, P.world);
  GMX_HIP(hipGetLastError());
  return 0;
}

extern "C" int gmx_shard_step_peer(int kind, const uint32_t key[2], const void* stats_own_d, gmx_peer Pe, int64_t* plan_d,
                                   uint64_t* total_out_d, const float* lw_d, float* max_out_d, int shift,
                                   int64_t n_per_rank, const void* const* state_rows_h, void* const* tail_rows_h,
                                   int32_t* next_idx_d, gmx_stream stream) {
  if (shard_check("gmx_shard_step_peer", kind, key, Pe.rank, Pe.world, n_per_rank)) return 1;
  if (peer_check("gmx_shard_step_peer", Pe, n_per_rank)) return 1;
  if (!stats_own_d || !plan_d || !lw_d || !max_out_d || !state_rows_h || !tail_rows_h || !next_idx_d)
    return gmx_fail("gmx_shard_step_peer: null argument%s");
  if (n_per_rank + (int64_t)Pe.world * Pe.capacity > 0x7fffffffLL)
    return gmx_fail("gmx_shard_step_peer: extended state index exceeds int32%s");
  if (shift < 1 || shift > 62) return gmx_fail("gmx_shard_step_peer: shift out of range%s");
  if (((uintptr_t)lw_d & 15) || ((uintptr_t)stats_own_d & 7))
    return gmx_fail("gmx_shard_step_peer: lw_d must be 16-byte and stats_own_d 8-byte aligned%s");
  shard_peer_args P;
  memset(&P, 0, sizeof(P));
  P.land = (uint64_t* const*)Pe.land_d;
  P.tag_base = Pe.tag_base_d; P.status = Pe.status_d; P.step = Pe.step; P.leaves = Pe.leaves;
  for (int l = 0; l < Pe.leaves; ++l) {
    if (!state_rows_h[l] || !tail_rows_h[l]) return gmx_fail("gmx_shard_step_peer: a leaf pointer is null%s");
    P.state[l] = (const uint32_t*)state_rows_h[l];
    P.tail[l] = (uint32_t*)tail_rows_h[l];
  }
  const int64_t tiles = Pe.tiles;
  const float scale = gmx_pow2i(shift);
  uint32_t b0, b1;
  gmx_threefry2x32(key[0], key[1], 0u, 0u, &b0, &b1);           // bits32(key, 0) on the host
  const uint32_t u0 = (b0 ^ b1) >> 9;
  const bool small = Pe.world <= 8 && (int64_t)Pe.world * tiles <= 4 * GMX_BLOCK;
#define GMX_LAUNCH_SP2(KIND, SM)                                                                                      \
  hipLaunchKernelGGL((k_shard_step_fill<KIND, SM, true>), dim3((unsigned)tiles), dim3(GMX_BLOCK), 0, (hipStream_t)stream, key[0],  \
                     key[1], u0, lw_d, (const uint8_t*)stats_own_d, (size_t)0, (int)tiles, scale, Pe.rank, Pe.world,  \
                     (int32_t)n_per_rank, (int32_t)Pe.capacity, plan_d, total_out_d, max_out_d, (const uint32_t*)nullptr, \
                     (uint32_t*)nullptr, next_idx_d, P)
#define GMX_LAUNCH_SP(KIND) do { if (small) GMX_LAUNCH_SP2(KIND, true); else GMX_LAUNCH_SP2(KIND, false); } while (0)
  if (kind == GMX_RESAMPLE_SYSTEMATIC) GMX_LAUNCH_SP(GMX_RESAMPLE_SYSTEMATIC); else GMX_LAUNCH_SP(GMX_RESAMPLE_STRATIFIED);
#undef GMX_LAUNCH_SP
#undef GMX_LAUNCH_SP2
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// peer-mapped exchange (include/genmi.h "Peer-mapped exchange"): one launch, one rendezvous per collective
// ---------------------------------------------------------------------------
#define GMX_P2P_CHUNK (16u * 1024u)         /* bytes one workgroup moves: a block of `bytes` is split over up to 64 of them */
// No fences: a system-scope release / acquire on gfx950 is a write-back / invalidate of the XCD's whole L2 (measured: the
// exchange took the sharded step to 65 us beside the noise programs).  Instead every word that crosses is stored
// WRITE-THROUGH (a system-scope relaxed atomic store: `global_store ... sc0 sc1`, nothing left dirty in L2), each storing
// wave drains its stores (`s_waitcnt vmcnt(0)`: the write-through stores have been acknowledged), the workgroup's
// barrier collects its waves, one lane takes a ticket, and the LAST ticket holder's flag store (relaxed, system scope)
// therefore follows every store of the block.  The reader polls the flag with system-scope loads and then reads the
// landing data with system-scope loads (past the caches): nothing to invalidate.
__device__ __forceinline__ void p2p_put(uint8_t* t, const uint8_t* s, size_t lo, size_t hi) {
  if ((((uintptr_t)s | (uintptr_t)t | lo | hi) & 7) == 0) {
    const uint64_t* s8 = reinterpret_cast<const uint64_t*>(s);
    uint64_t* t8 = reinterpret_cast<uint64_t*>(t);
    for (size_t i = lo / 8 + threadIdx.x; i < hi / 8; i += GMX_BLOCK)
      __hip_atomic_store(t8 + i, s8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += GMX_BLOCK)
      __hip_atomic_store(t + i, s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void p2p_take(uint8_t* t, const uint8_t* s, size_t lo, size_t hi) {
  if ((((uintptr_t)s | (uintptr_t)t | lo | hi) & 7) == 0) {
    const uint64_t* s8 = reinterpret_cast<const uint64_t*>(s);
    uint64_t* t8 = reinterpret_cast<uint64_t*>(t);
    for (size_t i = lo / 8 + threadIdx.x; i < hi / 8; i += GMX_BLOCK)
      t8[i] = __hip_atomic_load(s8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += GMX_BLOCK)
      t[i] = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_p2p_exchange(const uint8_t* __restrict__ src, size_t src_stride, void* const* __restrict__ land_peers,
               const uint8_t* __restrict__ land_local, uint8_t* __restrict__ out, uint64_t* const* __restrict__ flag_peers,
               uint64_t* __restrict__ flags_local, uint64_t* __restrict__ state, int rank, int world, size_t bytes) {
  __shared__ uint32_t s_ok;
  const int d = (int)blockIdx.x;                        // the peer this workgroup serves
  const uint32_t nblk = gridDim.y, j = blockIdx.y;      // ... and its share [lo, hi) of the block of `bytes`
  size_t per = (bytes + nblk - 1) / nblk;
  per = (per + 15) & ~(size_t)15;
  const size_t lo = (size_t)j * per < bytes ? (size_t)j * per : bytes, hi = lo + per < bytes ? lo + per : bytes;
  uint32_t* end_ticket = reinterpret_cast<uint32_t*>(state + 2);
  uint32_t* put_ticket = reinterpret_cast<uint32_t*>(state + 3 + d);
  const uint64_t epoch = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  const size_t half = (size_t)(epoch & 1ull) * (size_t)world * bytes;      // landing buffers alternate by epoch parity
  // ---- put: my block for peer d -> peer d's landing buffer, slot `rank` (write-through, drained) ----
  p2p_put((uint8_t*)land_peers[d] + half + (size_t)rank * bytes, src + (size_t)d * src_stride, lo, hi);
  __syncthreads();                                      // every wave of the workgroup has drained its stores
  if (threadIdx.x == 0) {
    // the LAST of the nblk workgroups serving peer d announces the whole block: its ticket follows every other
    // workgroup's (their stores were acknowledged before they took theirs)
    if (__hip_atomic_fetch_add(put_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1u) {
      __hip_atomic_store(put_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag_peers[d] + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait: peer d's block for me (relaxed system-scope polls: past the caches) ----
    uint32_t ok = 1u;
    if (__hip_atomic_load(flags_local + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      const uint64_t t0 = wall_clock64();
      while (__hip_atomic_load(flags_local + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 >= GMX_PEER_TIMEOUT_TICKS) {      // a peer that never arrives must not hang the GPU
          __hip_atomic_store(state + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0u;
          break;
        }
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  // ---- copy out: slot d of my landing buffer (fine-grained memory, system-scope loads) -> the caller's destination ----
  if (s_ok) p2p_take(out + (size_t)d * bytes, land_local + half + (size_t)d * bytes, lo, hi);
  __syncthreads();
  if (threadIdx.x == 0) {
    // the last workgroup of the launch advances the epoch for the next one
    if (__hip_atomic_fetch_add(end_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)world * nblk - 1u) {
      __hip_atomic_store(end_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" int gmx_p2p_alloc(size_t bytes, void** ptr_out, void* handle_out) {
  if (!ptr_out || !handle_out || bytes == 0) return gmx_fail("gmx_p2p_alloc: bad argument%s");
  static_assert(sizeof(hipIpcMemHandle_t) <= GMX_P2P_HANDLE_BYTES, "IPC handle size");
  void* p = nullptr;
  // fine-grained: stores from a peer over xGMI must be seen by this GPU's kernels without a cache flush
  // (no fallback to ordinary memory: a peer's stores would then not be visible to a running kernel — fail loudly)
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { (void)hipGetLastError(); return gmx_fail("gmx_p2p_alloc: fine-grained device memory is unavailable (%s)", hipGetErrorString(e)); }
  GMX_HIP(hipMemset(p, 0, bytes));
  hipIpcMemHandle_t h;
  memset(handle_out, 0, GMX_P2P_HANDLE_BYTES);
  e = hipIpcGetMemHandle(&h, p);
  if (e == hipSuccess) memcpy(handle_out, &h, sizeof(h));
  else (void)hipGetLastError();                         // a single process never opens it (world size 1)
  *ptr_out = p;
  return 0;
}
extern "C" int gmx_p2p_open(const void* handle, void** ptr_out) {
  if (!handle || !ptr_out) return gmx_fail("gmx_p2p_open: null argument%s");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  GMX_HIP(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}
extern "C" int gmx_p2p_close(void* ptr) { if (ptr) GMX_HIP(hipIpcCloseMemHandle(ptr)); return 0; }
extern "C" int gmx_p2p_free(void* ptr) { if (ptr) GMX_HIP(hipFree(ptr)); return 0; }

extern "C" int gmx_p2p_exchange(const void* src_d, size_t src_stride, void* const* land_peers_d, const void* land_local_d,
                                void* out_d, uint64_t* const* flag_peers_d, uint64_t* flags_local_d, uint64_t* state_d,
                                int rank, int world, size_t bytes, gmx_stream stream) {
  if (!src_d || !land_peers_d || !land_local_d || !out_d || !flag_peers_d || !flags_local_d || !state_d)
    return gmx_fail("gmx_p2p_exchange: null argument%s");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return gmx_fail("gmx_p2p_exchange: rank / world out of range%s");
  if (bytes == 0) return 0;
  // state_d: [0] epoch, [1] error, [2] end ticket, [3 + d] the put ticket of peer d (low words)
  unsigned nblk = (unsigned)((bytes + GMX_P2P_CHUNK - 1) / GMX_P2P_CHUNK);
  nblk = nblk < 1u ? 1u : (nblk > 64u ? 64u : nblk);
  // every workgroup waits on a peer that in turn waits for ALL of this launch's puts: the whole grid must be resident
  // (256 CUs x 8 workgroups); keep it to at most 1024 workgroups
  while (nblk > 1u && (unsigned)world * nblk > 1024u) nblk >>= 1;
  hipLaunchKernelGGL(k_p2p_exchange, dim3((unsigned)world, nblk), dim3(GMX_BLOCK), 0, (hipStream_t)stream,
                     (const uint8_t*)src_d, src_stride, land_peers_d, (const uint8_t*)land_local_d, (uint8_t*)out_d,
                     flag_peers_d, flags_local_d, state_d, rank, world, bytes);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// gather / select over a table of leaves
// ---------------------------------------------------------------------------
#define GMX_MAX_LEAVES 32
struct leaf_table {
  const void* a[GMX_MAX_LEAVES];
  const void* b[GMX_MAX_LEAVES];
  void* out[GMX_MAX_LEAVES];
  int32_t bytes[GMX_MAX_LEAVES];
  int32_t n;
};

__device__ __forceinline__ void copy_elem(void* dst, int64_t di, const void* src, int64_t si, int bytes) {
  if (bytes == 4) ((uint32_t*)dst)[di] = ((const uint32_t*)src)[si];
  else if (bytes == 1) ((uint8_t*)dst)[di] = ((const uint8_t*)src)[si];
  else if (bytes == 8) ((uint64_t*)dst)[di] = ((const uint64_t*)src)[si];
  else if (bytes == 2) ((uint16_t*)dst)[di] = ((const uint16_t*)src)[si];
  else {
    const uint8_t* s = (const uint8_t*)src + si * bytes;
    uint8_t* d = (uint8_t*)dst + di * bytes;
    for (int k = 0; k < bytes; ++k) d[k] = s[k];
  }
}

__global__ void __launch_bounds__(GMX_BLOCK)
k_gather(const leaf_table T, const int32_t* __restrict__ anc, int64_t n_out) {
  int64_t j = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (j >= n_out) return;
  int64_t a = anc[j];
  for (int l = 0; l < T.n; ++l) copy_elem(T.out[l], j, T.a[l], a, T.bytes[l]);
}
// every leaf 4 bytes wide, ancestors and destinations 16-byte aligned: four consecutive outputs per thread — one 16-byte
// load of ancestors, and per leaf four 4-byte loads (near each other: resampling's ancestors ascend) and ONE 16-byte store
__global__ void __launch_bounds__(GMX_BLOCK)
k_gather4(const leaf_table T, const int32_t* __restrict__ anc, int64_t n_out) {
  const int64_t j = ((int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x) * 4;
  if (j >= n_out) return;
  if (j + 4 <= n_out) {
    const int4 a = *reinterpret_cast<const int4*>(anc + j);
    for (int l = 0; l < T.n; ++l) {
      const uint32_t* s = (const uint32_t*)T.a[l];
      const uint4 v = make_uint4(s[a.x], s[a.y], s[a.z], s[a.w]);
      *reinterpret_cast<uint4*>((uint32_t*)T.out[l] + j) = v;
    }
    return;
  }
  for (int64_t q = j; q < n_out; ++q) {
    const int64_t a = anc[q];
    for (int l = 0; l < T.n; ++l) ((uint32_t*)T.out[l])[q] = ((const uint32_t*)T.a[l])[a];
  }
}
__global__ void __launch_bounds__(GMX_BLOCK)
k_select(const leaf_table T, const uint8_t* __restrict__ mask, int64_t n) {
  int64_t j = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (j >= n) return;
  bool m = mask[j] != 0;
  for (int l = 0; l < T.n; ++l) copy_elem(T.out[l], j, m ? T.a[l] : T.b[l], j, T.bytes[l]);
}

extern "C" int gmx_gather(const void* const* src_d, void* const* dst_d, const int32_t* elem_bytes,
                          int32_t n_leaves, const int32_t* ancestors_d, int64_t n_out,
                          gmx_stream stream) {
  if (n_out <= 0 || n_leaves <= 0) return 0;
  if (!src_d || !dst_d || !elem_bytes || !ancestors_d) return gmx_fail("gmx_gather: null argument%s");
  for (int32_t base = 0; base < n_leaves; base += GMX_MAX_LEAVES) {
    leaf_table T;
    memset(&T, 0, sizeof(T));
    T.n = (n_leaves - base < GMX_MAX_LEAVES) ? n_leaves - base : GMX_MAX_LEAVES;
    for (int l = 0; l < T.n; ++l) {
      if (!src_d[base + l] || !dst_d[base + l]) return gmx_fail("gmx_gather: null leaf%s");
      if (elem_bytes[base + l] <= 0) return gmx_fail("gmx_gather: bad element size%s");
      T.a[l] = src_d[base + l]; T.out[l] = dst_d[base + l]; T.bytes[l] = elem_bytes[base + l];
    }
    bool wide = (((uintptr_t)ancestors_d) & 15) == 0;
    for (int l = 0; l < T.n; ++l) wide = wide && T.bytes[l] == 4 && (((uintptr_t)T.out[l]) & 15) == 0 && (((uintptr_t)T.a[l]) & 3) == 0;
    if (wide)
      hipLaunchKernelGGL(k_gather4, grid_for((n_out + 3) / 4), dim3(GMX_BLOCK), 0, (hipStream_t)stream, T, ancestors_d, n_out);
    else
      hipLaunchKernelGGL(k_gather, grid_for(n_out), dim3(GMX_BLOCK), 0, (hipStream_t)stream, T, ancestors_d, n_out);
  }
  GMX_HIP(hipGetLastError());
  return 0;
}
extern "C" int gmx_select(const uint8_t* mask_d, const void* const* a_d, const void* const* b_d,
                          void* const* out_d, const int32_t* elem_bytes, int32_t n_leaves, int64_t n,
                          gmx_stream stream) {
  if (n <= 0 || n_leaves <= 0) return 0;
  if (!mask_d || !a_d || !b_d || !out_d || !elem_bytes) return gmx_fail("gmx_select: null argument%s");
  for (int32_t base = 0; base < n_leaves; base += GMX_MAX_LEAVES) {
    leaf_table T;
    memset(&T, 0, sizeof(T));
    T.n = (n_leaves - base < GMX_MAX_LEAVES) ? n_leaves - base : GMX_MAX_LEAVES;
    for (int l = 0; l < T.n; ++l) {
      if (!a_d[base + l] || !b_d[base + l] || !out_d[base + l]) return gmx_fail("gmx_select: null leaf%s");
      if (elem_bytes[base + l] <= 0) return gmx_fail("gmx_select: bad element size%s");
      T.a[l] = a_d[base + l]; T.b[l] = b_d[base + l]; T.out[l] = out_d[base + l];
      T.bytes[l] = elem_bytes[base + l];
    }
    hipLaunchKernelGGL(k_select, grid_for(n), dim3(GMX_BLOCK), 0, (hipStream_t)stream, T, mask_d, n);
  }
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// categorical per row (Gumbel-max; ParticleCollection.sample_particle)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GMX_BLOCK)
k_categorical_rows(const uint32_t* __restrict__ keys, const float* __restrict__ logits, int64_t rows,
                   int64_t cols, int32_t* __restrict__ out) {
  int64_t row = (int64_t)blockIdx.x * (GMX_BLOCK / GMX_WAVE) + (threadIdx.x >> 6);
  if (row >= rows) return;
  int lane = threadIdx.x & 63;
  uint2 kk = reinterpret_cast<const uint2*>(keys)[row];
  gmx_key k; k.k0 = kk.x; k.k1 = kk.y;
  const float* x = logits + row * cols;
  float best = -gmx_inf();
  int64_t bi = 0x7fffffffffffffffLL;
  for (int64_t j = lane; j < cols; j += GMX_WAVE) {
    float v = x[j] + gmx_gumbel_from_bits(gmx_bits32(k, (uint64_t)j));
    if (bi == 0x7fffffffffffffffLL || v > best) { best = v; bi = j; }
  }
  // wave argmax, lowest index wins ties
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    float ov = __shfl_xor(best, m, GMX_WAVE);
    uint32_t lo = (uint32_t)bi, hi = (uint32_t)((uint64_t)bi >> 32);
    lo = __shfl_xor(lo, m, GMX_WAVE); hi = __shfl_xor(hi, m, GMX_WAVE);
    int64_t oi = (int64_t)(((uint64_t)hi << 32) | lo);
    bool other_valid = oi != 0x7fffffffffffffffLL;
    bool mine_valid = bi != 0x7fffffffffffffffLL;
    if (other_valid && (!mine_valid || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = (int32_t)bi;
}
extern "C" int gmx_categorical_rows(const uint32_t* keys_d, const float* logits_d, int64_t rows,
                                    int64_t cols, int32_t* out_idx_d, gmx_stream stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 0x7fffffffLL) return gmx_fail("gmx_categorical_rows: cols out of range%s");
  if (!keys_d || !logits_d || !out_idx_d) return gmx_fail("gmx_categorical_rows: null argument%s");
  int64_t blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(k_categorical_rows, dim3((unsigned)blocks), dim3(GMX_BLOCK), 0,
                     (hipStream_t)stream, keys_d, logits_d, rows, cols, out_idx_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// MH accept
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GMX_BLOCK)
k_mh_accept(const uint32_t* __restrict__ keys, const float* __restrict__ log_alpha, int64_t n,
            uint8_t* __restrict__ accept) {
  int64_t i = (int64_t)blockIdx.x * GMX_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint2 kk = reinterpret_cast<const uint2*>(keys)[i];
  gmx_key k; k.k0 = kk.x; k.k1 = kk.y;
  float u = gmx_uniform_sample(k, 0, 0.0f, 1.0f);
  accept[i] = gmx_logf(u) < log_alpha[i] ? 1 : 0;
}
extern "C" int gmx_mh_accept(const uint32_t* keys_d, const float* log_alpha_d, int64_t n,
                             uint8_t* accept_d, gmx_stream stream) {
  if (n <= 0) return 0;
  if (!keys_d || !log_alpha_d || !accept_d) return gmx_fail("gmx_mh_accept: null argument%s");
  hipLaunchKernelGGL(k_mh_accept, grid_for(n), dim3(GMX_BLOCK), 0, (hipStream_t)stream, keys_d,
                     log_alpha_d, n, accept_d);
  GMX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// graph capture + timers
// ---------------------------------------------------------------------------
struct gmx_graph { hipGraph_t graph; hipGraphExec_t exec; };

extern "C" int gmx_capture_begin(gmx_stream stream) {
  GMX_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}
extern "C" int gmx_capture_end(gmx_stream stream, gmx_graph** out) {
  if (!out) return gmx_fail("gmx_capture_end: null argument%s");
  hipGraph_t g = nullptr;
  GMX_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    return gmx_fail("hipGraphInstantiate: %s", hipGetErrorString(err));
  }
  gmx_graph* h = new (std::nothrow) gmx_graph;
  if (!h) return gmx_fail("gmx_capture_end: out of host memory%s");
  h->graph = g; h->exec = e;
  *out = h;
  return 0;
}
extern "C" int gmx_graph_launch(gmx_graph* g, gmx_stream stream) {
  if (!g) return gmx_fail("gmx_graph_launch: null graph%s");
  GMX_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return 0;
}
extern "C" int gmx_graph_destroy(gmx_graph* g) {
  if (!g) return 0;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

struct gmx_timer { hipEvent_t a, b; };
extern "C" int gmx_timer_create(gmx_timer** out) {
  if (!out) return gmx_fail("gmx_timer_create: null argument%s");
  gmx_timer* t = new (std::nothrow) gmx_timer;
  if (!t) return gmx_fail("gmx_timer_create: out of host memory%s");
  GMX_HIP(hipEventCreate(&t->a));
  GMX_HIP(hipEventCreate(&t->b));
  *out = t;
  return 0;
}
extern "C" int gmx_timer_start(gmx_timer* t, gmx_stream stream) {
  if (!t) return gmx_fail("gmx_timer_start: null timer%s");
  GMX_HIP(hipEventRecord(t->a, (hipStream_t)stream));
  return 0;
}
extern "C" int gmx_timer_stop(gmx_timer* t, gmx_stream stream) {
  if (!t) return gmx_fail("gmx_timer_stop: null timer%s");
  GMX_HIP(hipEventRecord(t->b, (hipStream_t)stream));
  return 0;
}
extern "C" int gmx_timer_elapsed_ms(gmx_timer* t, float* ms_out) {
  if (!t || !ms_out) return gmx_fail("gmx_timer_elapsed_ms: null argument%s");
  GMX_HIP(hipEventSynchronize(t->b));
  GMX_HIP(hipEventElapsedTime(ms_out, t->a, t->b));
  return 0;
}
extern "C" int gmx_timer_destroy(gmx_timer* t) {
  if (!t) return 0;
  (void)hipEventDestroy(t->a);
  (void)hipEventDestroy(t->b);
  delete t;
  return 0;
}
