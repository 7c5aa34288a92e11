#!/usr/bin/env python3
"""Generates tests/golden/full_size.json: the CPU oracle's outputs for the BASELINE configs at their FULL sizes, as
hashes plus a few hundred sampled entries, so that the GPU suite compares the device results with a committed vector
instead of re-running minutes of numpy in every run (the same comparison; a mid-size case of each config still runs the
oracle live in tests/test_gpu_parity.py).  The oracle is the build's own restatement (oracle/genjax_oracle.py): these
are data — inputs are seeds, outputs are hashes — not reference source.

    python tests/golden/make_full_size.py            (about three minutes of CPU)
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


def main():
    out = {"about": "oracle outputs at full size: sha256 of the raw little-endian arrays + SAMPLES entries at fixed indices"}
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
