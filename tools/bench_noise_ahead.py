"""A/B of BootstrapSweep's two forms on one box (config 2): the one-stream sweep and the noise-ahead (two-stream) one.
Checks that both leave the same particles / weights / ancestors / evidence, then times graph replays.  One JSON line."""
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

n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
out = {"n": n, "T": T, "group": os.environ.get("GENMI_NOISE_GROUP")}
state = {}
for name, na in (("one_stream", False), ("noise_ahead", True)):
    sw = BootstrapSweep(init, step, n, T, noise_ahead=na).prepare(G.key(314159), torch.from_numpy(ys))
    assert sw.noise_ahead == na
    sw.launch()                      # eager
    torch.cuda.synchronize()
    eager = [v.clone() for v in sw.state()] + [sw.totals.clone()]
    sw.capture()
    sw.launch()
    torch.cuda.synchronize()
    graph = [v.clone() for v in sw.state()] + [sw.totals.clone()]
    reps = int(os.environ.get("REPS", 20))
    t0 = time.perf_counter()
    for _ in range(reps):
        sw.launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    state[name] = graph
    out[name] = {"us_per_step": 1e6 * dt / T, "particle_steps_per_s": n * T / dt, "log_ml": sw.log_ml(),
                 "graph_equals_eager": all(torch.equal(a, b) for a, b in zip(eager, graph))}
out["bit_identical"] = all(torch.equal(a, b) for a, b in zip(state["one_stream"], state["noise_ahead"]))
print(json.dumps(out))
