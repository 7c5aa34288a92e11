"""A/B of BootstrapSweep builds that differ in ONE constructor keyword, in one process on one box (config 2 by default;
CONFIG=3: the nonlinear SSM with one MH move per step):

  python tools/bench_ab.py fuse_resample False True         (values are Python literals)
  python tools/bench_ab.py noise_ahead False True

Every variant is prepared and captured with the keyword set to its value, the variants are then timed in turn
(ROUNDS rounds of REPS graph replays each, so that clock / thermal drift hits all of them alike) and their final
particles, log-weights, ancestors and integer totals are compared.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.smc import BootstrapSweep

var, values = sys.argv[1], sys.argv[2:]
n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
cfg = int(os.environ.get("CONFIG", 2))
reps, rounds = int(os.environ.get("REPS", 10)), int(os.environ.get("ROUNDS", 4))
if cfg == 3:
    ys = workloads.nlssm_data(T)
    init, step = workloads.make_nlssm(G)
    kw = dict(rejuvenate=G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))}),
              step_extra=lambda t: (float(t),))
else:
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    kw = {}
if os.environ.get("RESAMPLE"):            # systematic (default) | stratified | multinomial
    kw["resample"] = os.environ["RESAMPLE"]
sweeps, finals = {}, {}
import ast
for v in values:
    G.clear_caches()
    sw = BootstrapSweep(init, step, n, T, **dict(kw, **{var: ast.literal_eval(v)})).prepare(G.key(314159), torch.from_numpy(ys))
    sw.capture()
    sw.launch()
    torch.cuda.synchronize()
    finals[v] = [t.clone() for t in sw.state()] + [sw.totals.clone(), sw.maxs.clone()]
    sweeps[v] = sw
times = {v: [] for v in values}
for _ in range(rounds):
    for v in values:
        sw = sweeps[v]
        sw.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sw.launch()
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) / reps)
out = {"keyword": var, "n": n, "T": T, "config": cfg, "resample": os.environ.get("RESAMPLE", "systematic"),
       "us_per_step": {v: [round(1e6 * t / T, 3) for t in ts] for v, ts in times.items()},
       "best_us_per_step": {v: round(1e6 * min(ts) / T, 3) for v, ts in times.items()},
       "bit_identical": all(all(torch.equal(a, b) for a, b in zip(finals[values[0]], finals[v])) for v in values[1:])}
print(json.dumps(out))
