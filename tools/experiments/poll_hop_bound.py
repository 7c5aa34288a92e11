"""VERDICT r4 item 5, bounded BEFORE building it: what removing the write-through + poll hop of the resample-first
prologue could save at most.  Runs config 2's captured sweep (and the chain alone) under the shipped library and under
the diagnostic `nopoll` build (tools/experiments/build_diag_lib.py: gathers through the identity instead of waiting for
its slots' ancestor words — wrong numbers, right amount of everything else), each in its own process.
Usage: python tools/experiments/poll_hop_bound.py > out.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import torch
import genjax_amd as G
from genjax_amd import workloads, _lib
from genjax_amd.inference.smc import BootstrapSweep
n, T = 1_000_000, 100
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
sw = BootstrapSweep(init, step, n, T).prepare(G.key(314159), torch.from_numpy(ys)).capture()
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best
out = {"lib": os.path.basename(_lib.LIB_PATH), "one_launch_per_step": bool(sw.fuse), "noise_ahead": bool(sw.noise_ahead)}
out["sweep_us_per_step"] = 1e6 * timed(sw.launch) / T
import bench
be = _lib.get()
r = bench.measure_roofline(be, sw, n, T, 1, True, n * T / (out["sweep_us_per_step"] * T * 1e-6))
out["sweep_us_per_step_event_timed"] = r["us_per_step_event_timed"]
out["chain_only_us_per_step_event_timed"] = r["chain_only_us_per_step_event_timed"]
try:
    out["log_ml"] = sw.log_ml()
except Exception as e:
    out["log_ml"] = repr(e)[:80]
print("RESULT " + json.dumps(out))
''' % ROOT
res = {}
for name, lib in (("shipped", None), ("nopoll", os.path.join(ROOT, "tools", "experiments", "_alt", "libgenmi_nopoll.so"))):
    env = dict(os.environ)
    if lib:
        env["GENMI_LIB"] = lib
        env["GENMI_JIT_CACHE"] = "0"
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    res[name] = json.loads(line[0][7:]) if line else {"error": (p.stderr or p.stdout)[-600:]}
if all("chain_only_us_per_step_event_timed" in v for v in res.values()):
    res["upper_bound_of_the_gain_us_per_step"] = {
        "chain_only": res["shipped"]["chain_only_us_per_step_event_timed"] - res["nopoll"]["chain_only_us_per_step_event_timed"],
        "sweep": res["shipped"]["sweep_us_per_step"] - res["nopoll"]["sweep_us_per_step"]}
print(json.dumps(res))
