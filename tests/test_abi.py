"""The C-ABI library loads without a GPU and exports every symbol
include/genmi.h declares (no compute calls here); the hostsim harness exports
the same set, so host-logic tests exercise the same boundary."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "genmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_surveyed_entry_points():
    syms = declared_symbols()
    for name in ("gmx_version", "gmx_last_error", "gmx_split", "gmx_fold_in", "gmx_program_create",
                 "gmx_program_run", "gmx_program_destroy", "gmx_logsumexp", "gmx_weight_cdf", "gmx_ancestors",
                 "gmx_resample", "gmx_gather", "gmx_mh_accept", "gmx_select", "gmx_categorical_rows"):
        assert name in syms


def test_hip_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "genjax_amd", "lib", "libgenmi_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build_hip()
    lib = ctypes.CDLL(so)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.gmx_version.restype = ctypes.c_int
    from genjax_amd import _lib
    assert lib.gmx_version() == _lib.ABI_VERSION == 8
    # pure-host entry point: Threefry known-answer vector (no GPU involved)
    out = (ctypes.c_uint32 * 2)()
    lib.gmx_threefry2x32_host(ctypes.c_uint32(0x13198A2E), ctypes.c_uint32(0x03707344),
                              ctypes.c_uint32(0x243F6A88), ctypes.c_uint32(0x85A308D3), out)
    assert (out[0], out[1]) == (0xC4923A9C, 0x483DF7A0)


def test_hostsim_exports_the_same_boundary():
    import tests.hostsim as hs
    lib = ctypes.CDLL(hs.build())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_genmi_lib_names_another_build(tmp_path):
    """GENMI_LIB: the product loads the library the switch names (tuning builds) — and says so when it cannot"""
    import shutil
    import subprocess
    import sys
    import __graft_entry__ as g
    other = tmp_path / "libgenmi_other.so"
    shutil.copy(g.HIP_SO, other)
    code = ("from genjax_amd import _lib; import ctypes; assert _lib.LIB_PATH == %r, _lib.LIB_PATH; "
            "lib = ctypes.CDLL(_lib.LIB_PATH); lib.gmx_version.restype = ctypes.c_int; "
            "assert lib.gmx_version() == _lib.ABI_VERSION") % str(other)
    env = dict(os.environ, GENMI_LIB=str(other))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]


def test_environment_switches_are_the_documented_ones():
    """Every GENMI_* switch the code reads is in DESIGN.md section 10's table (15 of them), and nothing else: a new
    switch is a new code path somebody has to keep parity-tested."""
    import re
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("## 10. Switches"):]
    documented = set(re.findall(r"`(GENMI_[A-Z_]+)`", sec))
    used = set()
    for base, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "profiles", "__pycache__", "_build", "experiments")]
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh", ".c")) and f != "test_abi.py":
                txt = open(os.path.join(base, f), errors="ignore").read()
                used |= set(re.findall(r"GENMI_[A-Z_]+", txt))
    # GENMI_H: the header's include guard
    used -= {"GENMI_", "GENMI_H"}
    assert not any("GENMI_JIT_DEFS" in open(os.path.join(ROOT, "genjax_amd", *f)).read()
                   for f in (("_lib.py",), ("engine.py",), ("csrc", "gmx_rng.h")))      # (the diagnostic short-Threefry build is gone)
    assert used <= documented, sorted(used - documented)
    assert len(documented) <= 20


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device (and without the test harness
    installed) every entry point raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from genjax_amd import _lib
    _lib.install(None)
    import genjax_amd as genjax
    with pytest.raises(_lib.GenmiError):
        genjax.normal.simulate(genjax.key(0), (0.0, 1.0))


def test_specialisation_compiles_offline():
    """gmx_program_specialize's translation unit builds with hiprtc for gfx950
    (no GPU needed for the compile itself)."""
    import numpy as np
    import torch
    import tests.hostsim as hs
    hs.install()
    try:
        import genjax_amd as genjax
        from genjax_amd import workloads
        from genjax_amd.core.choice_map import ChoiceMap
        from genjax_amd.engine import Gathered
        from genjax_amd.static import MinimalGenerate
        _, step = workloads.make_lgssm(genjax)
        n = 64
        p = MinimalGenerate(step, (Gathered(torch.zeros(n), torch.zeros(n, dtype=torch.int32)),),
                            ChoiceMap.empty().set("y", torch.tensor(0.3)), (n,))
        blob = np.ascontiguousarray(p.comp.blob, dtype=np.uint32)
    finally:
        hs.uninstall()
    so = os.path.join(ROOT, "genjax_amd", "lib", "libgenmi_hip.so")
    lib = ctypes.CDLL(so)
    lib.gmx_specialize_dryrun.restype = ctypes.c_size_t
    lib.gmx_specialize_dryrun.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.c_char_p, ctypes.c_size_t]
    log = ctypes.create_string_buffer(4096)
    size = lib.gmx_specialize_dryrun(blob.ctypes.data, blob.size, log, 4096, None, 0)
    assert size > 0, log.value.decode()


def test_entry_points_reject_null_arguments_before_any_launch():
    """Host-side validation comes first: with null pointers every entry point returns non-zero and leaves
    a message in gmx_last_error() — nothing reaches the GPU (runs on a box without one)."""
    so = os.path.join(ROOT, "genjax_amd", "lib", "libgenmi_hip.so")
    lib = ctypes.CDLL(so)
    lib.gmx_last_error.restype = ctypes.c_char_p
    N, i64, i32, u32 = ctypes.c_void_p(0), ctypes.c_int64, ctypes.c_int, ctypes.c_uint32
    calls = {
        "gmx_split": (N, i64(10), i64(0), N, N),
        "gmx_split_rows": (N, i64(4), i64(4), N, N),
        "gmx_fold_in": (N, u32(1), i64(10), N, N),
        "gmx_random_bits": (N, i64(4), i64(4), N, N),
        "gmx_logsumexp": (N, i64(3), i64(5), N, N, N, N),
        "gmx_reduce_max": (N, i64(4), N, N),
        "gmx_weight_cdf": (N, i64(10), i32(40), N, i64(0), N, N, N, N, N),
        "gmx_ancestors": (i32(0), N, N, i64(10), ctypes.c_uint64(0), N, i64(10), i64(0), i64(10), N, N),
        "gmx_resample": (i32(0), N, N, i64(10), i32(40), N, i64(0), N, N, N, N, N),
        "gmx_tile_stats": (N, i64(10), i32(40), N, N, N),
        "gmx_resample_tiles": (i32(0), N, N, i64(10), i32(40), N, N, N, N, N, N),
        "gmx_resample_tiles_p": (i32(0), N, N, i64(10), i32(40), N, N, N, N, N, N),
        "gmx_tile_prefix": (N, N, i64(10), N, N),
        "gmx_multinomial_tiled": (N, N, i64(10), i32(40), N, N, N, N, N, N, N, i32(-1), N),
        "gmx_slot_uniforms": (N, i32(1), i64(10), N, i32(0), N),
        "gmx_resample_tiles_u": (i32(1), N, N, i64(10), i32(40), N, N, N, N, N, N, N),
        "gmx_sorted_uniforms": (N, i32(1), i64(10), N, i32(0), N),
        "gmx_resample_sorted": (N, N, i64(10), i32(40), N, N, N, i32(0), N, N, N, N),
        "gmx_shard_totals": (N, i32(2), i64(1024), N, N, N),
        "gmx_shard_step_tiles": (i32(0), N, N, N, N, N, N, N, i32(40), i32(0), i32(2), i64(1024), i64(4), N, N, N, N),
        "gmx_shard_step_fused": (i32(0), N, N, N, N, N, N, i32(40), i32(0), i32(2), i64(1024), i64(4), N, N, N, N),
        "gmx_gather": (N, N, N, ctypes.c_int32(3), N, i64(10), N),
        "gmx_select": (N, N, N, N, N, ctypes.c_int32(1), i64(10), N),
        "gmx_categorical_rows": (N, N, i64(4), i64(4), N, N),
        "gmx_mh_accept": (N, N, i64(4), N, N),
        "gmx_program_create": (N, ctypes.c_size_t(0), N),
        "gmx_program_run": (N, i64(4), N, N),
        "gmx_shard_plan": (i32(0), N, N, i32(0), i32(4), i64(10), N, N),
        "gmx_shard_route": (i32(0), N, N, N, i32(0), i32(4), i64(10), i64(4), N, N, N, N),
        "gmx_shard_step": (i32(0), N, N, N, N, N, i32(0), i32(4), i64(10), i64(4), N, N, N, N),
        "gmx_graph_launch": (N, N),
        "gmx_capture_end": (N, N),
        "gmx_timer_start": (N, N),
    }
    for name, args in calls.items():
        rc = getattr(lib, name)(*args)
        assert rc != 0, name
        assert name.encode() in lib.gmx_last_error(), (name, lib.gmx_last_error())
