#!/usr/bin/env python3
"""gmx_resample (tile statistics + k_offspring_tile) on weight vectors of growing skew: us / launch.
The per-thread slot loop (GENMI_RS_FILL=0) costs the wave's largest offspring count; the LDS fill does not care."""
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
out = {"GENMI_RS_FILL": os.environ.get("GENMI_RS_FILL", "(default: 1)")}
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
print(json.dumps(out))
