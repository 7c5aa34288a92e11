"""EXPERIMENT: what the dependent `ancestor -> state` gather costs the noise-ahead site program (J'): the same hoisted
step program with its state argument gathered through ancestors (what runs) and as a plain per-particle tensor (what a
state-pushing resampler would feed it), 100 back-to-back launches each in a hipGraph.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import genjax_amd as G
from genjax_amd import _lib, workloads
from genjax_amd.core.choice_map import ChoiceMap
from genjax_amd.engine import Gathered
from genjax_amd.inference.smc import cdf_shift
from genjax_amd.static import MinimalGenerate

n = 1_000_000
be = _lib.get()
dev = be.device
init, step = workloads.make_lgssm(G)
obs = ChoiceMap.empty().set("y", torch.tensor(0.3, device=dev))
x = torch.randn(n, device=dev)
anc = torch.sort(torch.randint(0, n, (n,), device=dev, dtype=torch.int32)).values
z = torch.randn(n, device=dev)
out = {}
for name, arg in (("gathered", Gathered(x, anc)), ("plain", x)):
    p = MinimalGenerate(step, (arg,), obs, (n,), hoist_noise=True)
    p.comp.specialize()
    xo, lw = torch.zeros((1, n), device=dev), torch.zeros((1, n), device=dev)
    part = torch.zeros((2, (n + 255) // 256), device=dev)
    agg = torch.zeros(((n + 1023) // 1024,), dtype=torch.int64, device=dev)
    bufs = [None] * len(p.comp.outputs)
    bufs[p.ro[1]], bufs[p.wo[1]] = xo, lw
    leaves = p.leaves((arg,), obs, [z])

    def launch():
        p.comp.run(leaves, (n,), None, red_out=part, out_buffers=bufs, tile_stats=(agg, cdf_shift(n)))
    launch(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100):
            launch()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    out[name] = {"us_per_launch": 1e6 * (time.perf_counter() - t0) / 1000}
print(json.dumps(out))
