#!/usr/bin/env python3
"""Generates tools/calib_valu.hip — the vector-instruction ISSUE calibration of gfx950 (not part of
the product).

Every stream is ONE `asm volatile` block of 64 instructions inside a rolled loop on registers the block names itself,
so the compiler can neither fold, pack nor reorder it.  Streams rotate eight chains (`{d}` = chain register; an
instruction reads a result eight instructions old) unless marked dependent (`dep`: one chain).  `{a}`, `{b}` are
loop-invariant VGPRs, `{s}`, `{t}` loop-invariant SGPRs, `{m}` an SGPR pair holding a lane mask.

Residency is set by the grid (CUs x W workgroups of 256 threads → W waves per SIMD) and VERIFIED from HW_ID / XCC_ID;
every wave stamps s_memtime at the start and end of its loop, so the rate per SIMD is
   (instructions of the SIMD's waves) / (last end − first start on that SIMD)          [shader cycles]
and is not fooled by waves that do not overlap.  Wall time (HIP events) gives T lane-ops/s beside it.

    python tools/calib_valu_gen.py            # writes tools/calib_valu.hip
    hipcc --offload-arch=gfx950 -O2 tools/calib_valu.hip -o tools/calib_valu && tools/calib_valu [iters] [mix.txt]
"""
import os

ROOT = os.path.dirname(os.path.abspath(__file__))

# name, template of ONE instruction, flags
STREAMS = [
    ("v_xor_b32", "v_xor_b32 {d}, {d}, {a}", ""),
    ("v_xor_b32 dep", "v_xor_b32 {d}, {d}, {a}", "dep"),
    ("v_add_u32", "v_add_u32 {d}, {d}, {a}", ""),
    ("v_add_u32 sgpr-src", "v_add_u32 {d}, {s}, {d}", ""),
    ("v_add_u32 literal", "v_add_u32 {d}, 0x9e3779b9, {d}", ""),
    ("v_sub_u32", "v_sub_u32 {d}, {d}, {a}", ""),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {a}, {b}", ""),
    ("v_xad_u32", "v_xad_u32 {d}, {d}, {a}, {b}", ""),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {d}, {d}, 19", ""),
    ("v_alignbit_b32 dep", "v_alignbit_b32 {d}, {d}, {d}, 19", "dep"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {d}, 1, {d}", ""),
    ("v_lshrrev_b32", "v_lshrrev_b32 {d}, 1, {d}", ""),
    ("v_and_b32", "v_and_b32 {d}, {d}, {a}", ""),
    ("v_or_b32", "v_or_b32 {d}, {d}, {a}", ""),
    ("v_and_or_b32", "v_and_or_b32 {d}, {d}, {a}, {b}", ""),
    ("v_bfe_u32", "v_bfe_u32 {d}, {d}, 3, 20", ""),
    ("v_not_b32", "v_not_b32 {d}, {d}", ""),
    ("v_mov_b32", "v_mov_b32 {d}, {a}", ""),
    ("v_mov_b32 sgpr", "v_mov_b32 {d}, {s}", ""),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {a}", ""),
    ("v_mul_hi_u32", "v_mul_hi_u32 {d}, {d}, {a}", ""),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {a}", ""),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {a}, {b}", ""),
    ("v_lshl_add_u64", "v_lshl_add_u64 {D}, {D}, 2, {A}", "wide"),
    ("v_add_f32", "v_add_f32 {d}, {d}, {a}", ""),
    ("v_add_f32 dep", "v_add_f32 {d}, {d}, {a}", "dep"),
    ("v_sub_f32", "v_sub_f32 {d}, {d}, {a}", ""),
    ("v_mul_f32", "v_mul_f32 {d}, {d}, {a}", ""),
    ("v_mul_f32 sgpr-src", "v_mul_f32 {d}, {s}, {d}", ""),
    ("v_fma_f32", "v_fma_f32 {d}, {d}, {a}, {b}", ""),
    ("v_fma_f32 dep", "v_fma_f32 {d}, {d}, {a}, {b}", "dep"),
    ("v_fma_f32 sgpr-src", "v_fma_f32 {d}, {d}, {s}, {b}", ""),
    ("v_fmac_f32", "v_fmac_f32 {d}, {a}, {b}", ""),
    ("v_fmaak_f32", "v_fmaak_f32 {d}, {d}, {a}, 0x3f7fff00", ""),
    ("v_fmamk_f32", "v_fmamk_f32 {d}, {d}, 0x3f7fff00, {a}", ""),
    ("v_max_f32", "v_max_f32 {d}, {d}, {a}", ""),
    ("v_min_f32", "v_min_f32 {d}, {d}, {a}", ""),
    ("v_max3_f32", "v_max3_f32 {d}, {d}, {a}, {b}", ""),
    ("v_ldexp_f32", "v_ldexp_f32 {d}, {d}, {a}", ""),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {d}", ""),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {d}", ""),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {d}", ""),
    ("v_rndne_f32", "v_rndne_f32 {d}, {d}", ""),
    ("v_floor_f32", "v_floor_f32 {d}, {d}", ""),
    ("v_frexp_mant_f32", "v_frexp_mant_f32 {d}, {d}", ""),
    ("v_pk_fma_f32 (2 f32/lane)", "v_pk_fma_f32 {D}, {D}, {A}, {B}", "wide packed"),
    ("v_pk_mul_f32 (2 f32/lane)", "v_pk_mul_f32 {D}, {D}, {A}", "wide packed"),
    ("v_pk_add_f32 (2 f32/lane)", "v_pk_add_f32 {D}, {D}, {A}", "wide packed"),
    ("v_exp_f32", "v_exp_f32 {d}, {d}", ""),
    ("v_log_f32", "v_log_f32 {d}, {d}", ""),
    ("v_rcp_f32", "v_rcp_f32 {d}, {d}", ""),
    ("v_rsq_f32", "v_rsq_f32 {d}, {d}", ""),
    ("v_sqrt_f32", "v_sqrt_f32 {d}, {d}", ""),
    ("v_div_scale_f32", "v_div_scale_f32 {d}, vcc, {d}, {a}, {d}", ""),
    ("v_div_fmas_f32", "v_div_fmas_f32 {d}, {d}, {a}, {b}", ""),
    ("v_div_fixup_f32", "v_div_fixup_f32 {d}, {d}, {a}, {b}", ""),
    ("v_cmp_lt_f32 vcc", "v_cmp_lt_f32 vcc, {d}, {a}", "sink"),
    ("v_cmp_lt_f32 sgpr-pair", "v_cmp_lt_f32 {m}, {d}, {a}", "sink"),
    ("v_cmp_class_f32 vcc", "v_cmp_class_f32 vcc, {d}, {a}", "sink"),
    ("v_cndmask_b32 vcc", "v_cndmask_b32 {d}, {d}, {a}, vcc", ""),
    ("v_cndmask_b32 sgpr-pair", "v_cndmask_b32 {d}, {d}, {a}, {m}", ""),
    ("v_cmp_lt_f32 + v_cndmask_b32 (pair)", "v_cmp_lt_f32 vcc, {d}, {a}\nv_cndmask_b32 {d}, {d}, {b}, vcc", "pair"),
    ("v_cmp_lt_u32 + v_cndmask_b32 (pair)", "v_cmp_lt_u32 vcc, {d}, {a}\nv_cndmask_b32 {d}, {d}, {b}, vcc", "pair"),
    ("v_readlane_b32 -> sgpr", "v_readlane_b32 {t}, {d}, 3", "sink"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 {t}, {d}", "sink"),
    ("v_add_co_u32 + v_addc_co_u32 (pair)", "v_add_co_u32 {d}, vcc, {d}, {a}\nv_addc_co_u32 {d}, vcc, {d}, {b}, vcc", "pair"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf", ""),
    ("threefry round: add, alignbit, xor", "TF", "tf"),
    ("threefry round: add, alignbit, xor dep", "TF", "tf dep"),
    ("mix: 2 fma + 1 alignbit + 1 xor", "v_fma_f32 {d}, {d}, {a}, {b}\nv_alignbit_b32 {d2}, {d2}, {d2}, 19\nv_fma_f32 {d3}, {d3}, {a}, {b}\nv_xor_b32 {d4}, {d4}, {a}", "mix4"),
    ("mix: 3 add_u32 + 1 alignbit", "v_add_u32 {d}, {d}, {a}\nv_add_u32 {d2}, {d2}, {a}\nv_add_u32 {d3}, {d3}, {a}\nv_alignbit_b32 {d4}, {d4}, {d4}, 19", "mix4"),
    # --- which streams share the fast rate?  (cycled over the 64 slots; every instruction on its own rotating chain)
    ("mix 1:1 add_u32, fma", "v_add_u32 {d}, {d}, {a}|v_fma_f32 {d}, {d}, {a}, {b}", "cycle"),
    ("mix 1:1 add_u32, alignbit", "v_add_u32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:1 xor, alignbit", "v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:1 fma, alignbit", "v_fma_f32 {d}, {d}, {a}, {b}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:1 add_f32, alignbit", "v_add_f32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:1 mov, alignbit", "v_mov_b32 {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 3:1 xor, alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 3:1 add_f32, alignbit", "v_add_f32 {d}, {d}, {a}|v_add_f32 {d}, {d}, {a}|v_add_f32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 3:1 fma, alignbit", "v_fma_f32 {d}, {d}, {a}, {b}|v_fma_f32 {d}, {d}, {a}, {b}|v_fma_f32 {d}, {d}, {a}, {b}|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:3 fma, alignbit", "v_fma_f32 {d}, {d}, {a}, {b}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("mix 1:1 fma, max_f32", "v_fma_f32 {d}, {d}, {a}, {b}|v_max_f32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 add_f32, max_f32", "v_add_f32 {d}, {d}, {a}|v_max_f32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 add_u32, max_f32", "v_add_u32 {d}, {d}, {a}|v_max_f32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 fma, exp", "v_fma_f32 {d}, {d}, {a}, {b}|v_exp_f32 {d}, {d}", "cycle"),
    ("mix 3:1 fma, exp", "v_fma_f32 {d}, {d}, {a}, {b}|v_fma_f32 {d}, {d}, {a}, {b}|v_fma_f32 {d}, {d}, {a}, {b}|v_exp_f32 {d}, {d}", "cycle"),
    ("mix 1:1 fma, fma sgpr-src", "v_fma_f32 {d}, {d}, {a}, {b}|v_fma_f32 {d}, {d}, {s}, {b}", "cycle"),
    ("mix 1:1 fma, cndmask sgpr-pair", "v_fma_f32 {d}, {d}, {a}, {b}|v_cndmask_b32 {d}, {d}, {a}, {m}", "cycle"),
    ("mix 1:1 add_u32, xor", "v_add_u32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 add_u32, add_f32", "v_add_u32 {d}, {d}, {a}|v_add_f32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 fma, mul_lo_u32", "v_fma_f32 {d}, {d}, {a}, {b}|v_mul_lo_u32 {d}, {d}, {a}", "cycle"),
    ("mix 1:1 fma, s_nop", "v_fma_f32 {d}, {d}, {a}, {b}|s_nop 0", "cycle"),
    ("mix 1:1 alignbit, s_nop", "v_alignbit_b32 {d}, {d}, {d}, 19|s_nop 0", "cycle"),
    ("mix 1:1 alignbit, s_add_u32", "v_alignbit_b32 {d}, {d}, {d}, 19|s_add_u32 {t}, {t}, {s}", "cycle"),
    ("mix 1:1 fma, s_add_u32", "v_fma_f32 {d}, {d}, {a}, {b}|s_add_u32 {t}, {t}, {s}", "cycle"),
    ("runs of 2: xor then alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs of 4: xor then alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs of 8: xor then alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs of 16: xor then alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs of 32: xor then alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs: 8 xor/add then 4 alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs: 16 xor/add then 8 alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("runs: 32 xor/add then 16 alignbit", "v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_xor_b32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19|v_alignbit_b32 {d}, {d}, {d}, 19", "cycle"),
    ("threefry x4 in phases (4 add, 4 alignbit, 4 xor)", "RAW:v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7|v_alignbit_b32 %1, %1, %1, 19|v_alignbit_b32 %3, %3, %3, 19|v_alignbit_b32 %5, %5, %5, 19|v_alignbit_b32 %7, %7, %7, 19|v_xor_b32 %1, %1, %0|v_xor_b32 %3, %3, %2|v_xor_b32 %5, %5, %4|v_xor_b32 %7, %7, %6|v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7|v_alignbit_b32 %1, %1, %1, 19|v_alignbit_b32 %3, %3, %3, 19|v_alignbit_b32 %5, %5, %5, 19|v_alignbit_b32 %7, %7, %7, 19|v_xor_b32 %1, %1, %0|v_xor_b32 %3, %3, %2|v_xor_b32 %5, %5, %4|v_xor_b32 %7, %7, %6|v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7|v_alignbit_b32 %1, %1, %1, 19|v_alignbit_b32 %3, %3, %3, 19|v_alignbit_b32 %5, %5, %5, 19|v_alignbit_b32 %7, %7, %7, 19|v_xor_b32 %1, %1, %0|v_xor_b32 %3, %3, %2|v_xor_b32 %5, %5, %4|v_xor_b32 %7, %7, %6|v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7|v_alignbit_b32 %1, %1, %1, 19|v_alignbit_b32 %3, %3, %3, 19|v_alignbit_b32 %5, %5, %5, 19|v_alignbit_b32 %7, %7, %7, 19|v_xor_b32 %1, %1, %0|v_xor_b32 %3, %3, %2|v_xor_b32 %5, %5, %4|v_xor_b32 %7, %7, %6|v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7|v_alignbit_b32 %1, %1, %1, 19|v_alignbit_b32 %3, %3, %3, 19|v_alignbit_b32 %5, %5, %5, 19|v_alignbit_b32 %7, %7, %7, 19|v_xor_b32 %1, %1, %0|v_xor_b32 %3, %3, %2|v_xor_b32 %5, %5, %4|v_xor_b32 %7, %7, %6|v_add_u32 %0, %0, %1|v_add_u32 %2, %2, %3|v_add_u32 %4, %4, %5|v_add_u32 %6, %6, %7", "cycle"),
    ("mix: threefry round + 3 fma", "RAW8:v_add_u32 %0, %0, %1|v_alignbit_b32 %1, %1, %1, 19|v_xor_b32 %1, %1, %0|v_fma_f32 %2, %2, %10, %11|v_fma_f32 %3, %3, %10, %11|v_fma_f32 %4, %4, %10, %11|v_add_u32 %5, %5, %6|v_alignbit_b32 %6, %6, %6, 19", "cycle"),
    ("mix 1:1 alignbit, add_f32 sgpr-src", "v_alignbit_b32 {d}, {d}, {d}, 19|v_add_f32 {d}, {s}, {d}", "cycle"),
    ("mix 1:1 xor, cvt_f32_u32", "v_xor_b32 {d}, {d}, {a}|v_cvt_f32_u32 {d}, {d}", "cycle"),
    ("mix 1:1 add_f32, cvt_f32_u32", "v_add_f32 {d}, {d}, {a}|v_cvt_f32_u32 {d}, {d}", "cycle"),
    ("mix 1:1 add_f32, cmp_lt_f32", "v_add_f32 {d}, {d}, {a}|v_cmp_lt_f32 vcc, {d}, {a}", "cycle"),
    ("mix 1:1 add_f32, rcp", "v_add_f32 {d}, {d}, {a}|v_rcp_f32 {d}, {d}", "cycle"),
    ("mix 1:1 alignbit, rcp", "v_alignbit_b32 {d}, {d}, {d}, 19|v_rcp_f32 {d}, {d}", "cycle"),
    ("mix 1:1 xor, lshrrev", "v_xor_b32 {d}, {d}, {a}|v_lshrrev_b32 {d}, 1, {d}", "cycle"),
    ("mix 1:1 fma, v_mul_f32 literal", "v_fma_f32 {d}, {d}, {a}, {b}|v_mul_f32 {d}, 0x3f7fff00, {d}", "cycle"),
    ("mix 2:1 (xor, add_u32), lshlrev", "v_xor_b32 {d}, {d}, {a}|v_add_u32 {d}, {d}, {a}|v_lshlrev_b32 {d}, 1, {d}", "cycle"),
    ("s_nop 0", "s_nop 0", "scalar"),
    ("s_add_u32 (salu)", "s_add_u32 {t}, {t}, {s}", "scalar"),
]


def body(tmpl, flags):
    """64 instructions (or 32 pairs / 16 groups of four) as asm lines with %N operands.
    32-bit streams: %0..%7 = chains, %8 = t (SGPR, written), %9 = m (SGPR pair, written), %10 = a, %11 = b (VGPRs), %12 = s (SGPR);
    wide streams:   %0..%3 = 64-bit chains, %4 = A, %5 = B (64-bit VGPR pairs)."""
    out = []
    if "tf" in flags:
        dep = "dep" in flags
        for r in range(21):
            p = 0 if dep else (r % 4) * 2
            out += [f"v_add_u32 %{p}, %{p}, %{p + 1}", f"v_alignbit_b32 %{p + 1}, %{p + 1}, %{p + 1}, 19", f"v_xor_b32 %{p + 1}, %{p + 1}, %{p}"]
        out.append("v_add_u32 %0, %0, %1")
        return out
    if "wide" in flags:
        for k in range(64):
            c = 0 if "dep" in flags else k % 4
            out.append(tmpl.format(D=f"%{c}", A="%4", B="%5"))
        return out
    if "mix4" in flags:
        for k in range(16):
            c = (k % 2) * 4
            out += tmpl.format(d=f"%{c}", d2=f"%{c + 1}", d3=f"%{c + 2}", d4=f"%{c + 3}", a="%10", b="%11").split("\n")
        return out
    if "cycle" in flags:
        if tmpl.startswith("RAW:"):
            return tmpl[4:].split("|")
        if tmpl.startswith("RAW8:"):
            ts = tmpl[5:].split("|")
            return [ts[k % 8] for k in range(64)]
        ts = tmpl.split("|")
        for k in range(64):
            out.append(ts[k % len(ts)].format(d=f"%{k % 8}", a="%10", b="%11", s="%12", t="%8", m="%9"))
        return out
    n = 32 if "pair" in flags else 64
    for k in range(n):
        c = 0 if "dep" in flags else k % 8
        out += tmpl.format(d=f"%{c}", a="%10", b="%11", s="%12", t="%8", m="%9").split("\n")
    return out


def main():
    parts = [open(os.path.join(ROOT, "calib_valu_head.inc")).read()]
    names = []
    for i, (name, tmpl, flags) in enumerate(STREAMS):
        lines = body(tmpl, flags)
        assert len(lines) == 64, (name, len(lines))
        txt = "".join(f'      "{ln}\\n"\n' for ln in lines)
        if "wide" in flags:
            ops = '"+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb)'
        else:
            ops = ('"+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(st), "+s"(sm) '
                   ': "v"(a), "v"(b), "s"(sa)')
        parts.append(f"""
__global__ __launch_bounds__(256) void k_s{i}(rec* out, uint32_t* sink, int iters, uint32_t a, uint32_t b) {{
  PROLOGUE
#pragma unroll 1
  for (int k = 0; k < iters; ++k) {{
    asm volatile(
{txt}      : {ops} : "vcc", "scc");
  }}
  EPILOGUE
}}
""")
        lanes = 128 if "packed" in flags else 64
        names.append((name, i, lanes))
    parts.append("static const stream_t streams[] = {\n" + "".join(f'  {{"{n}", k_s{i}, {l}}},\n' for n, i, l in names) + "};\n")
    parts.append(open(os.path.join(ROOT, "calib_valu_tail.inc")).read())
    with open(os.path.join(ROOT, "calib_valu.hip"), "w") as fh:
        fh.write("// GENERATED by tools/calib_valu_gen.py — edit that file (and calib_valu_head.inc / _tail.inc), not this one\n" + "".join(parts))


if __name__ == "__main__":
    main()
