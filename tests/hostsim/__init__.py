"""Test-only CPU stand-in for libgenmi_hip.so (see hostsim.cpp)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgmx_hostsim.so")


def build():
    src = os.path.join(_HERE, "hostsim.cpp")
    deps = [src, os.path.join(_HERE, "..", "..", "include", "genmi.h")] + [
        os.path.join(_HERE, "..", "..", "genjax_amd", "csrc", f)
        for f in ("gmx_vm.h", "gmx_dist.h", "gmx_rng.h", "gmx_math.h", "gmx_program.h", "gmx_sorted.h", "gmx_peer.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-I", os.path.join(_HERE, "..", "..", "include"), src, "-o", _SO, "-lrt"])
    return _SO


SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def build_sanitized():
    """The same mirror under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: tests/test_sanitizers.py
    loads it in a child process started with the ASan runtime preloaded)."""
    so = os.path.join(_HERE, "_build", "libgmx_hostsim_san.so")
    src = os.path.join(_HERE, "hostsim.cpp")
    deps = [src, os.path.join(_HERE, "..", "..", "include", "genmi.h")] + [
        os.path.join(_HERE, "..", "..", "genjax_amd", "csrc", f)
        for f in ("gmx_vm.h", "gmx_dist.h", "gmx_rng.h", "gmx_math.h", "gmx_program.h", "gmx_sorted.h", "gmx_peer.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"] + SAN_FLAGS +
                              ["-I", os.path.join(_HERE, "..", "..", "include"), src, "-o", so, "-lrt"])
    return so


def install():
    from genjax_amd import _lib
    path = os.environ.get("GENMI_HOSTSIM_SO") or build()        # GENMI_HOSTSIM_SO: the sanitized build
    be = _lib.Backend(ctypes.CDLL(path), torch.device("cpu"), uses_streams=False)
    _lib.install(be)
    return be


def uninstall():
    from genjax_amd import _lib
    _lib.install(None)
