"""BASELINE config 5 timing: K = 64 cluster-assignment update over N = 1e6 datapoints
(gibbs_categorical: one launch, no [N, K] matrix).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, workloads
from genjax_amd.inference import gibbs

n, K = int(os.environ.get("N", 1_000_000)), 64
x, guess, probs, z = workloads.mixture_data(n, K)
gd = workloads.make_mixture(G)
args = (torch.from_numpy(probs).cuda(), torch.from_numpy(guess).cuda())
chm = C["obs"].set(torch.from_numpy(x).cuda())
out = {}
from genjax_amd import engine
engine.JIT_MIN_PARTICLES = engine.JIT_MIN_WORK = 1 << 62   # no automatic specialisation: time the interpreter first
for mode in ("interpreter", "specialised"):
    if mode == "specialised":
        t0 = time.perf_counter()
        for comp, _ in gibbs._CACHE.values():
            ok = comp.specialize()
        out["hiprtc_compile_s"] = time.perf_counter() - t0
        out["specialised_ok"] = bool(ok)
    idx = gibbs.gibbs_categorical(G.key(1), gd, args, chm, "idx", K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for r in range(reps):
        idx = gibbs.gibbs_categorical(G.key(r), gd, args, chm, "idx", K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[mode] = {"ms": 1e3 * dt, "datapoints_per_s": n / dt, "gumbels_per_s": n * K / dt,
                 "algorithmic_GBps": 8.0 * n / dt / 1e9}
out["accuracy_vs_generating_component"] = float((idx.cpu().numpy() == z).mean())
print(json.dumps({"workload": "config 5: mixture assignments", "n": n, "K": K, **out}))
