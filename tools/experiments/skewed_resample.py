#!/usr/bin/env python3
"""gmx_resample (tile statistics + k_offspring_tile) on weight vectors of growing skew: us / launch.
(History: a per-thread slot loop cost the wave's largest offspring count — 2688 us at N(0, 4) log-weights; the LDS
fill, the only form since round 3, does not care about the weights.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctypes import c_uint32
import numpy as np
import torch
from genjax_amd import _lib
from genjax_amd.inference.smc import cdf_shift

be = _lib.get()
n = 1_000_000
dev = be.device
rng = np.random.default_rng(3)
cases = {
    "flat": np.zeros(n, np.float32),
    "normal(0, 1) log-weights": rng.normal(0, 1, n).astype(np.float32),
    "normal(0, 4) log-weights": rng.normal(0, 4, n).astype(np.float32),
    "one particle in 1000 carries the mass": np.where(rng.random(n) < 1e-3, 0.0, -50.0).astype(np.float32),
    "one particle carries everything": np.where(np.arange(n) == 123456, 0.0, -1e4).astype(np.float32),
}
shift = cdf_shift(n)
ws = torch.zeros(((be.c.gmx_resample_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
anc = torch.zeros((n,), dtype=torch.int32, device=dev)
mx = torch.zeros((1,), dtype=torch.float32, device=dev)
tot = torch.zeros((1,), dtype=torch.int64, device=dev)
kk = (c_uint32 * 2)(0, 42)
out = {}
for name, lw_h in cases.items():
    lw = torch.from_numpy(lw_h).to(dev)

    def go():
        be.check(be.c.gmx_resample(0, kk, be.ptr(lw), n, shift, None, 0, be.ptr(mx), be.ptr(tot), be.ptr(anc),
                                   be.ptr(ws), be.stream()), "gmx_resample")
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    out[name] = {"us_per_resample": 1e6 * (time.perf_counter() - t0) / reps,
                 "distinct_ancestors": int(torch.unique(anc).numel())}
# the large-n path (n > 2^21: gmx_weight_cdf + gmx_ancestors = k_offspring, one source per thread)
n2 = 4_000_000
shift2 = cdf_shift(n2)
ws2 = torch.zeros(((be.c.gmx_weight_cdf_workspace(n2) + 7) // 8,), dtype=torch.int64, device=dev)
cdf = torch.zeros((n2,), dtype=torch.int64, device=dev)
anc2 = torch.zeros((n2,), dtype=torch.int32, device=dev)
big = {}
for name, lw_h in cases.items():
    lw = torch.from_numpy(np.resize(lw_h, n2).copy()).to(dev)
    if name.startswith("one particle carries"):
        lw[:] = -1e4
        lw[123456] = 0.0

    mx.copy_(lw.max().reshape(1))           # the site program's block maxima normally provide it

    def go2():
        be.check(be.c.gmx_weight_cdf(be.ptr(lw), n2, shift2, None, 0, be.ptr(mx), be.ptr(cdf), be.ptr(tot), be.ptr(ws2),
                                     be.stream()), "gmx_weight_cdf")
        be.check(be.c.gmx_ancestors(0, kk, be.ptr(cdf), n2, 0, be.ptr(tot), n2, 0, n2, be.ptr(anc2), be.stream()),
                 "gmx_ancestors")
    for _ in range(2):
        go2()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        go2()
    torch.cuda.synchronize()
    big[name] = {"us_per_resample": 1e6 * (time.perf_counter() - t0) / reps,
                 "distinct_ancestors": int(torch.unique(anc2).numel())}
out["n = 4e6 (gmx_weight_cdf + gmx_ancestors)"] = big
print(json.dumps(out))
