// gmx_block.h — wave / block reductions shared by the AOT kernels and the
// hiprtc-specialised site programs (device code only).
//
// Order-independent reductions (max, integer sums, integer scans) go through DPP row shifts and
// row broadcasts (gfx9: row_shr:1/2/4/8, row_bcast:15, row_bcast:31) — six dependent VALU steps and
// one v_readlane, no LDS round trip.  The ds_bpermute butterflies they replace cost one LDS-crossbar
// latency (and an s_waitcnt) per step, on the critical path of every kernel that ends in a block
// reduction.  Float SUMS keep the fixed butterfly order: their bits depend on it.
#pragma once
// Wave priority.  The kernels of a dependent chain (site programs, resampler) raise theirs, so that a BACKGROUND
// kernel running beside them on a second stream (a noise program of BootstrapSweep's noise-ahead form:
// gmx_program_set_background, priority 0) only takes the issue slots the chain leaves.  With nothing beside the
// chain every wave has the same priority and nothing changes.  Measured on MI355X (config 2, two streams):
// 16.5 -> 15.9 us/step (profiles/r02f_experiment_noise_ahead_prio_rows.txt).
#define GMX_CHAIN_PRIO 2
#define GMX_SETPRIO __builtin_amdgcn_s_setprio(GMX_CHAIN_PRIO);
#include "gmx_math.h"

#define GMX_BLOCK 256
#define GMX_WAVE 64

// v moved by one DPP pattern; lanes the pattern does not feed (or rows outside ROWMASK) read `identity`
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t gmx_dpp(uint32_t identity, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROWMASK, 0xf, false);
}
#define GMX_DPP_SCAN_STEPS(STEP) \
  STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)

// The scans below are ONE vector instruction per step: the operation itself reads its first operand through the
// DPP pattern (v_max_f32_dpp / v_add_co_u32_dpp), and a lane the pattern does not feed (bound_ctrl off) or whose
// row is masked is simply not written — the identity element for free.  (Through __builtin_amdgcn_update_dpp the
// compiler emits mov-identity + v_mov_dpp + the operation, and for a float max a canonicalising v_max before it.)
// `s_nop 1`: a DPP read of a VGPR needs two wait states after the VALU write of it, and the hazard recogniser does not
// look inside inline assembly; GMX_DPP_ASM_ENTER (`s_nop 4`, once per scan) covers the longest DPP hazard against
// whatever the compiler scheduled just before the block (a VALU write of EXEC: five wait states).
#define GMX_DPP_ASM_ENTER "s_nop 4\n\t"
#define GMX_DPP_ASM_STEPS(OP) \
  OP("row_shr:1 row_mask:0xf bank_mask:0xf") OP("row_shr:2 row_mask:0xf bank_mask:0xf") \
  OP("row_shr:4 row_mask:0xf bank_mask:0xf") OP("row_shr:8 row_mask:0xf bank_mask:0xf") \
  OP("row_bcast:15 row_mask:0xa bank_mask:0xf") OP("row_bcast:31 row_mask:0xc bank_mask:0xf")

// inclusive max-scan; lane 63 holds the wave maximum.  v_max_f32 (IEEE mode) returns the other operand when one is
// NaN and +0 for max(-0, +0): gmx_rmax below is the same function on the host.
__device__ __forceinline__ float wave_max_scan(float v) {
#define GMX_OP(CTRL) "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL "\n\t"
  asm(GMX_DPP_ASM_ENTER GMX_DPP_ASM_STEPS(GMX_OP) : "+v"(v));
#undef GMX_OP
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  return gmx_u2f((uint32_t)__builtin_amdgcn_readlane((int)gmx_f2u(wave_max_scan(v)), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, GMX_WAVE);
  return v;
}
// block of 256 threads = 4 waves; result broadcast to all threads
__device__ __forceinline__ float block_max(float v, float* lds4) {
  v = wave_max(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  float r = gmx_rmax(gmx_rmax(lds4[0], lds4[1]), gmx_rmax(lds4[2], lds4[3]));
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* lds4) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  float r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  return r;
}

// OP_REDMAX / OP_REDLSE epilogues.  `row` = index of the 256-particle group,
// `rows` = number of groups; partials are planar: max[rows] then sumexp[rows].
__device__ __forceinline__ void gmx_red_max(float* red_out, float* lds4, uint32_t row, float x, bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  if (threadIdx.x == 0 && red_out) red_out[row] = m;
}
__device__ __forceinline__ void gmx_red_lse(float* red_out, float* lds4, uint32_t row, uint32_t rows, float x,
                                            bool active) {
  float m = block_max(active ? x : -gmx_inf(), lds4);
  float e = active ? gmx_expf(x - m) : 0.0f;
  if (!(m > -gmx_inf())) e = 0.0f;  // empty / all -inf block
  float s = block_sum(e, lds4);
  if (threadIdx.x == 0 && red_out) {
    red_out[row] = m;
    red_out[(size_t)rows + row] = s;
  }
}

// ---- 64-bit wave primitives and the fixed-point weight of the two-level CDF (include/genmi.h) ----
// inclusive scan of u64 over the wave (integer: any order gives the same bits)
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t v) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#define GMX_OP(CTRL) "s_nop 1\n\tv_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL "\n\t"
  asm(GMX_DPP_ASM_ENTER GMX_DPP_ASM_STEPS(GMX_OP) : "+v"(lo), "+v"(hi) : : "vcc");
#undef GMX_OP
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_last_u64(uint64_t v) {      // lane 63's value, wave-uniform
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return wave_last_u64(wave_scan_u64(v)); }
// lane l receives lane l-1's value (wave_shr:1); lane 0 receives `first`
__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v, uint32_t first) { return gmx_dpp<0x138, 0xf>(first, v); }
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_up(lo, d, GMX_WAVE);
  hi = __shfl_up(hi, d, GMX_WAVE);
  return ((uint64_t)hi << 32) | lo;
}

// q = floor(exp(lw - ref) * 2^shift) as u64; NaN / negative / not below 2^63 (lw = +inf) -> 0.  `scale` = 2^shift
// (launch-uniform: its exponent is read back on the scalar unit); the integer form is gmx_math.h's gmx_exp_fixed.
__device__ __forceinline__ uint64_t weight_fixed(float lw, float ref, float scale) {
  return gmx_exp_fixed(lw - ref, (int)(gmx_f2u(scale) >> 23) - 127);
}

// ---- tile statistics -> tile prefixes, ONCE per resampling (include/genmi.h: gmx_tile_prefix) ----
// One workgroup of 256 threads turns the <= 2048 (m_b, A_b) into M, K = ceil(M / ln 2), the exclusive prefixes
// P_b = sum_{b' < b} A_b' >> (K - k_b'), and the total.  Thread t owns the `per` consecutive tiles [t per, (t+1) per)
// (per = ceil(tiles / 256): 1 .. 8): a running sum in registers, one u64 wave scan, three wave offsets.
// Every thread of the workgroup reaches both barriers.  lds: >= 4 floats and 4 u64.
#define GMX_TP_MAX_TILES 2048
#define GMX_TP_PER_MAX (GMX_TP_MAX_TILES / GMX_BLOCK)
// layout of the block (u64 words): [0, tiles) prefixes | [tiles] total | [tiles + 1] M, K | pad to a multiple of 16
GMX_HD size_t gmx_tile_prefix_words_(int64_t n) { return (size_t)(((n + 1023) / 1024 + 2 + 15) / 16) * 16; }
__device__ __forceinline__ void gmx_tile_prefix_block(const float* tmax, const uint64_t* agg, int n_tiles,
                                                      uint64_t* pref, float* lds4, uint64_t* lds8) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n_tiles + GMX_BLOCK - 1) / GMX_BLOCK;                 // 1 .. 8, uniform
  uint64_t ta[GMX_TP_PER_MAX];
  float tm[GMX_TP_PER_MAX];
#pragma unroll
  for (int r = 0; r < GMX_TP_PER_MAX; ++r) {          // loads first (clamped rows)
    ta[r] = 0ull; tm[r] = -gmx_inf();
    if (r < per) {
      const int t = tid * per + r;
      const int tc = t < n_tiles ? t : n_tiles - 1;
      ta[r] = agg[tc];
      tm[r] = tmax[tc];
    }
  }
  float M = -gmx_inf();
#pragma unroll
  for (int r = 0; r < GMX_TP_PER_MAX; ++r) {
    if (r < per) {
      const bool ok = tid * per + r < n_tiles;
      ta[r] = ok ? ta[r] : 0ull;
      tm[r] = ok ? tm[r] : -gmx_inf();
      M = gmx_rmax(M, tm[r]);
    }
  }
  M = block_max(M, lds4);
  const int32_t K = gmx_tile_exp(M);
  uint64_t P[GMX_TP_PER_MAX];
  uint64_t run = 0;
#pragma unroll
  for (int r = 0; r < GMX_TP_PER_MAX; ++r) {
    P[r] = run;
    if (r < per) run += gmx_tile_scale(ta[r], gmx_tile_exp(tm[r]), K);
  }
  const uint64_t inc = wave_scan_u64(run);
  __syncthreads();                                    // (lds8 may have been read by the caller just before)
  if (lane == 63) lds8[wave] = inc;
  __syncthreads();
  uint64_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const uint64_t v = lds8[w]; total += v; wave_off += (w < wave) ? v : 0ull; }
  const uint64_t base = wave_off + (inc - run);
#pragma unroll
  for (int r = 0; r < GMX_TP_PER_MAX; ++r) {
    if (r < per) {
      const int t = tid * per + r;
      if (t < n_tiles) pref[t] = base + P[r];
    }
  }
  if (tid == 0) {
    pref[n_tiles] = total;
    pref[n_tiles + 1] = (uint64_t)gmx_f2u(M) | ((uint64_t)(uint32_t)K << 32);
  }
}
