#!/usr/bin/env python3
"""Generates tests/golden/full_size.json: the CPU oracle's outputs for the BASELINE configs at their FULL sizes, as
hashes plus a few hundred sampled entries, so that the GPU suite compares the device results with a committed vector
instead of re-running minutes of numpy in every run (the same comparison; a mid-size case of each config still runs the
oracle live in tests/test_gpu_parity.py).  The oracle is the build's own restatement (oracle/genjax_oracle.py): these
are data — inputs are seeds, outputs are hashes — not reference source.

    python tests/golden/make_full_size.py                    (about four minutes of CPU)
    python tests/golden/make_full_size.py config2_sizes      (only the config-2 sizes 2e6 / 8e6: oracle/orc_sweep.c)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import genjax_oracle as O  # noqa: E402

O.build()
from tests import parity  # noqa: E402

SAMPLES = 300


def digest(a):
    a = np.ascontiguousarray(a)
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "dtype": str(a.dtype), "shape": list(a.shape)}


def sampled(a, idx):
    a = np.ascontiguousarray(a).reshape(-1)
    v = a[idx]
    if v.dtype.kind == "f":
        return [float(np.float32(x)).hex() for x in v]
    return [int(x) for x in v]


def config2_size(n, T=100, seed=314159):
    """BASELINE config 2 at n particles through oracle/orc_sweep.c (the C restatement; one core): every step's integer
    total and maximum (the terms of the log-ML), the last step's ancestors, the log-ML"""
    import ctypes
    from genjax_amd import workloads          # (data only: the observations)
    ys = workloads.lgssm_data(T)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liborc_sweep.so"))
    f32, u64, i32 = np.float32, np.uint64, np.int32
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    shift = O.cdf_shift(n)
    ox, ox2, olw = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, f32)
    ocdf, oanc = np.zeros(n, u64), np.zeros(n, i32)
    omax, otot = np.zeros(T, f32), np.zeros(T, u64)
    rc = lib.orc_lgssm_sweep(ctypes.c_int64(n), ctypes.c_int64(T), P(ys), ctypes.c_uint32(0), ctypes.c_uint32(seed),
                             ctypes.c_float(0.9), ctypes.c_float(0.5), ctypes.c_float(1.0), ctypes.c_float(1.0),
                             ctypes.c_int(shift), P(ox), P(ox2), P(olw), P(ocdf), P(oanc), P(omax), P(otot))
    assert rc == 0
    lml = float(np.sum(np.array([O.cdf_reference(v) for v in omax], np.float64) + np.log(otot.astype(np.float64))
                       - shift * np.log(2.0) - np.log(n)))
    idx = np.random.default_rng(1).integers(0, n, SAMPLES)
    return {"n": n, "T": T, "seed": seed, "index": [int(i) for i in idx], "totals": [int(v) for v in otot],
            "maxs_bits": [int(v) for v in omax.view(np.uint32)], "anc": dict(digest(oanc), sample=sampled(oanc, idx)),
            "log_ml": lml.hex()}


def main():
    path = os.path.join(HERE, "full_size.json")
    if len(sys.argv) > 1 and sys.argv[1] == "config2_sizes":       # only this part, kept beside what the file holds
        out = json.load(open(path))
        out["config2_sizes"] = {str(n): config2_size(n) for n in (2_000_000, 8_000_000)}
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
        print("wrote", path)
        return
    out = {"about": "oracle outputs at full size: sha256 of the raw little-endian arrays + SAMPLES entries at fixed indices"}
    out["config2_sizes"] = {str(n): config2_size(n) for n in (2_000_000, 8_000_000)}
    # ---- config 3: nonlinear SSM, 1e6 particles x 100 steps, one Rejuvenate move per step (parity.oracle_nlssm_mh_sweep) ----
    n, T, seed = 1_000_000, 100, 7
    ref = parity.oracle_nlssm_mh_sweep(n, T, seed)
    idx = np.random.default_rng(0).integers(0, n, SAMPLES)
    out["config3"] = {"n": n, "T": T, "seed": seed, "index": [int(i) for i in idx],
                      "x": dict(digest(ref["x"]), sample=sampled(ref["x"], idx)),
                      "lw": dict(digest(ref["lw"]), sample=sampled(ref["lw"], idx)),
                      "acc": dict(digest(np.asarray(ref["acc"], np.bool_)), sample=sampled(np.asarray(ref["acc"], np.uint8), idx)),
                      "log_ml": float(sum(ref["terms"])).hex(), "accept_rate": float(np.mean(ref["acc"]))}
    with open(os.path.join(HERE, "full_size.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", os.path.join(HERE, "full_size.json"))


if __name__ == "__main__":
    main()
