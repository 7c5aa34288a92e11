"""the ordered resampler alone, systematic against the sorted multinomial (table cold: 16 tables in turn; hot: one table):
us per launch over back-to-back launches on an idle GPU; and the table build per row."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ctypes import c_uint32
import genjax_amd as G
from genjax_amd import _lib
from genjax_amd.inference.smc import cdf_shift
be = _lib.get(); dev = be.device
n = int(os.environ.get("N", 1_000_000)); sigma = float(os.environ.get("SIGMA", 1.5)); NT = 16
rng = np.random.default_rng(0)
lw = torch.from_numpy(rng.normal(0, sigma, n).astype(np.float32)).to(dev)
shift = cdf_shift(n); tiles = (n + 1023) // 1024
tmax = torch.zeros((tiles,), dtype=torch.float32, device=dev); agg = torch.zeros((tiles,), dtype=torch.int64, device=dev)
be.check(be.c.gmx_tile_stats(be.ptr(lw), n, shift, be.ptr(tmax), be.ptr(agg), be.stream()), "stats")
words = int(be.c.gmx_sorted_uniforms_words(n))
keys = torch.from_numpy(np.stack([np.asarray(G.key(100 + r).host(), np.uint32) for r in range(NT)]).view(np.int32)).to(dev)
tables = torch.zeros((NT, words), dtype=torch.int32, device=dev)
be.check(be.c.gmx_sorted_uniforms(be.ptr(keys), NT, n, be.ptr(tables), 0, be.stream()), "tables")
mx = torch.zeros((1,), dtype=torch.float32, device=dev); tot = torch.zeros((1,), dtype=torch.int64, device=dev)
anc = torch.zeros((n,), dtype=torch.int32, device=dev)
kk = (c_uint32 * 2)(1, 2)
def t_(fn, reps=200):
    for i in range(10): fn(i)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
out = {"n": n, "sigma": sigma}
out["systematic"] = t_(lambda i: be.c.gmx_resample_tiles(0, kk, be.ptr(lw), n, shift, be.ptr(tmax), be.ptr(agg), be.ptr(mx), be.ptr(tot), be.ptr(anc), be.stream()))
out["stratified"] = t_(lambda i: be.c.gmx_resample_tiles(1, kk, be.ptr(lw), n, shift, be.ptr(tmax), be.ptr(agg), be.ptr(mx), be.ptr(tot), be.ptr(anc), be.stream()))
out["sorted_cold"] = t_(lambda i: be.c.gmx_resample_sorted(kk, be.ptr(lw), n, shift, be.ptr(tmax), be.ptr(agg), be.ptr(tables[i % NT]), 1, be.ptr(mx), be.ptr(tot), be.ptr(anc), be.stream()))
out["sorted_hot"] = t_(lambda i: be.c.gmx_resample_sorted(kk, be.ptr(lw), n, shift, be.ptr(tmax), be.ptr(agg), be.ptr(tables[0]), 1, be.ptr(mx), be.ptr(tot), be.ptr(anc), be.stream()))
out["table_build_per_row"] = t_(lambda i: be.c.gmx_sorted_uniforms(be.ptr(keys), NT, n, be.ptr(tables), 0, be.stream()), reps=20) / NT
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
