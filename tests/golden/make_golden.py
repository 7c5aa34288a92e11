#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle.

The reference cannot be imported in the build container (no jax / tfp; Python
3.10 < the 3.11 it requires — SURVEY.md §8c), so these vectors come from the
build's own oracle, whose Threefry core is pinned by the public Random123
known-answer vectors and whose jax.random / TFP layering is restated from the
published algorithms.  If a jax 0.5.2 + tfp 0.23 environment
becomes available, regenerate them from the real libraries and diff.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import genjax_oracle as O  # noqa: E402


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    k = O.key(314159)
    out = {}
    out["threefry_kat"] = [
        {"key": [0, 0], "ctr": [0, 0], "out": [0x6B200159, 0x99BA4EFE]},
        {"key": [0xFFFFFFFF, 0xFFFFFFFF], "ctr": [0xFFFFFFFF, 0xFFFFFFFF], "out": [0x1CB996FC, 0xBB002BE7]},
        {"key": [0x13198A2E, 0x03707344], "ctr": [0x243F6A88, 0x85A308D3], "out": [0xC4923A9C, 0x483DF7A0]},
    ]
    out["key"] = [int(v) for v in k]
    out["split8"] = O.split(k, 8).tolist()
    out["fold_in_1_to_4"] = [O.fold_in(k, d).tolist() for d in (1, 2, 3, 4)]
    out["bits32_first16"] = O.bits32(k[None, :], np.arange(16, dtype=np.uint64)).tolist()
    site = O.fold_in(k, 1)
    # 64-sample streams: element j of a vector site = counter j of the site key
    n = 64
    keys = O.split(k, n)
    f = lambda a: [float(np.float32(v)).hex() for v in np.asarray(a).reshape(-1)]
    out["uniform64"] = f(O.random_uniform(site, (n,)))
    out["normal64"] = f(O.random_normal(site, (n,)))
    out["gumbel64"] = f(O.random_gumbel(site, (n,)))
    out["normal_site_per_particle64"] = f(O.normal.sample(O.fold_in(keys, 1), np.float32(1.5), np.float32(2.0)))
    out["flip_per_particle64"] = [int(v) for v in O.flip.sample(O.fold_in(keys, 2), np.float32(0.3))]
    logits = np.log(np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float32))
    out["categorical_per_particle64"] = [int(v) for v in O.categorical.sample(O.fold_in(keys, 3), logits)]
    out["beta22_per_particle64"] = f(O.beta.sample(O.fold_in(keys, 4), np.float32(2.0), np.float32(2.0)))
    # log-densities
    xs = np.array([-2.0, -0.5, 0.0, 0.3, 1.0, 4.0], dtype=np.float32)
    out["normal_logpdf"] = f(O.normal.logpdf(xs, np.float32(0.5), np.float32(1.5)))
    out["beta_logpdf"] = f(O.beta.logpdf(np.array([0.1, 0.3, 0.5, 0.9], np.float32), np.float32(2.0), np.float32(3.0)))
    out["assess_literal"] = float(np.float32(O.normal.logpdf(np.float32(1.0), np.float32(0.0), np.float32(1.0)) +
                                             O.normal.logpdf(np.float32(-1.0), np.float32(0.0), np.float32(1.0))))
    # resampling
    rng = np.random.default_rng(0)
    res = {}
    for n_ in (8, 1024, 1_000_000):
        lw = rng.normal(0, 2, n_).astype(np.float32)
        cdf, total, M, shift = O.weight_cdf(lw)
        ent = {"lw_sha256": h(lw), "total": str(total), "shift": shift, "max": float(M).hex(), "cdf_sha256": h(cdf)}
        for kind, name in ((O.SYSTEMATIC, "systematic"), (O.STRATIFIED, "stratified"), (O.MULTINOMIAL, "multinomial")):
            anc = O.ancestors(kind, O.key(99), cdf)
            ent[name] = anc.tolist() if n_ <= 1024 else None
            ent[name + "_sha256"] = h(anc)
        anc = O.ancestors_multinomial_tiled(O.key(99), cdf)          # the two-stage multinomial (round 3)
        ent["multinomial_tiled"] = anc.tolist() if n_ <= 1024 else None
        ent["multinomial_tiled_sha256"] = h(anc)
        anc = O.ancestors_multinomial_sorted(O.key(99), cdf) if n_ <= 4096 else O.ancestors_multinomial_sorted_c(O.key(99), cdf)
        ent["multinomial_sorted"] = anc.tolist() if n_ <= 1024 else None      # multinomial with sorted uniforms (round 3)
        ent["multinomial_sorted_sha256"] = h(anc)
        ent["sorted_exponentials_head"] = O.sorted_exponentials(O.key(99), 8).tolist()
        if n_ <= 1024:
            ent["lw"] = f(lw)
        res[str(n_)] = ent
    out["resample"] = res
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", os.path.join(HERE, "oracle_vectors.json"))


if __name__ == "__main__":
    main()
