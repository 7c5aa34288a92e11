"""where BASELINE config 4's resampling step goes at k = 1e7 (beyond the fused resampler's 2^21): weight CDF, ancestors,
gather of the ten latents — us per call, HIP events"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import _lib, engine
from genjax_amd.inference import smc
be = _lib.get(); dev = be.device
k = int(os.environ.get("K", 10_000_000))
lw = torch.from_numpy(np.random.default_rng(0).normal(0, 2, k).astype(np.float32)).to(dev)
leaves = [torch.randn(k, device=dev) for _ in range(2)] + [torch.randn(k, 8, device=dev)]
def t_(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
out = {"k": k}
box = {}
def cdf(): box["c"] = smc.weight_cdf(lw)
out["weight_cdf_us"] = t_(cdf)
cdf_, total, mx, shift = box["c"]
def anc(): box["a"] = smc.ancestors_from_cdf(smc.SYSTEMATIC, G.key(3), cdf_, total)
out["ancestors_us"] = t_(anc)
out["gather_10_rows_us"] = t_(lambda: engine.gather_leaves(leaves, box["a"]))
if k <= 2048 * 1024:
    out["fused_resample_us"] = t_(lambda: smc.resample_fused(smc.SYSTEMATIC, G.key(3), lw))
print(json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in out.items()}))
