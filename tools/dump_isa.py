#!/usr/bin/env python3
"""Offline (no GPU): build the bench's LGSSM step program through the tests' CPU mirror of the C-ABI, specialise it
with hiprtc for gfx950 (gmx_specialize_dryrun) and disassemble the code object.

  python tools/dump_isa.py [out_prefix]     -> <out_prefix>.co, <out_prefix>.s  (default /tmp/gmx_step)
Prints the instruction-class histogram of gmx_jit_kernel.
"""
import collections
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def step_blob(which="step"):
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
        init, step = workloads.make_lgssm(genjax)
        n = 64
        obs = ChoiceMap.empty().set("y", torch.tensor(0.3))
        if which == "init":
            p = MinimalGenerate(init, (), obs, (n,))
        else:
            p = MinimalGenerate(step, (Gathered(torch.zeros(n), torch.zeros(n, dtype=torch.int32)),), obs, (n,))
        return np.ascontiguousarray(p.comp.blob, dtype=np.uint32)
    finally:
        hs.uninstall()


def compile_blob(blob, out_prefix):
    so = os.path.join(ROOT, "genjax_amd", "lib", "libgenmi_hip.so")
    lib = ctypes.CDLL(so)
    lib.gmx_specialize_dryrun2.restype = ctypes.c_size_t
    lib.gmx_specialize_dryrun2.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                           ctypes.c_char_p, ctypes.c_size_t]
    log = ctypes.create_string_buffer(1 << 16)
    code = ctypes.create_string_buffer(8 << 20)
    flags = int(os.environ.get("DUMP_ISA_FLAGS", "0"))       # 1: the resample-first prologue, 2: a background kernel
    size = lib.gmx_specialize_dryrun2(blob.ctypes.data, blob.size, flags, log, len(log), code, len(code))
    if not size:
        raise SystemExit("hiprtc failed: " + log.value.decode())
    co = out_prefix + ".co"
    with open(co, "wb") as fh:
        fh.write(code.raw[:size])
    # hiprtc returns a clang offload bundle or a bare ELF depending on version: unbundle if needed
    elf = co
    if code.raw[:24].startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
        elf = out_prefix + ".elf"
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={co}", f"--output={elf}"])
    s = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", elf]).decode()
    with open(out_prefix + ".s", "w") as fh:
        fh.write(s)
    meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", elf], capture_output=True).stdout.decode()
    with open(out_prefix + ".meta", "w") as fh:
        fh.write(meta)
    return s, meta


def histogram(s):
    h = collections.Counter()
    for line in s.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            h["VALU"] += 1
        elif op.startswith("s_"):
            h["SALU/ctrl"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
            h["VMEM"] += 1
        elif op.startswith("ds_"):
            h["LDS"] += 1
        h[op] += 0
    return h


if __name__ == "__main__":
    prefix = sys.argv[1] if len(sys.argv) > 1 else "/tmp/gmx_step"
    which = sys.argv[2] if len(sys.argv) > 2 else "step"
    s, meta = compile_blob(step_blob(which), prefix)
    print({k: v for k, v in histogram(s).items() if v})
    for key in (".vgpr_count", ".sgpr_count", ".lds_size", ".private_segment_fixed_size"):
        m = re.search(re.escape(key) + r":\s*(\d+)", meta)
        print(key, m.group(1) if m else "?")
    print("wrote", prefix + ".s")
