"""BASELINE config 4 (8-schools, ImportanceK k = 1e7 + one systematic resample + the gathered theta), piece by piece"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
from genjax_amd.inference import smc
sig = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
ysch = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)
@G.gen
def schools():
    mu = G.normal(0.0, 5.0) @ "mu"
    log_tau = G.normal(0.0, 1.0) @ "log_tau"
    theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
    _ = G.normal(theta, jnp.array(sig)) @ "y"
    return theta
k = int(os.environ.get("K", 10_000_000))
alg = smc.ImportanceK(G.Target(schools, (), C["y"].set(ysch)), k_particles=k)
def t_(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
box = {}
def imp(): box["c"] = alg.run_smc(G.key(2))
def res(): box["r"] = smc.resample(G.key(3), box["c"], "systematic")
def mat(): box["theta"] = box["r"].get_particles().get_choices()["theta"]
out = {"k": k, "importance_us": t_(imp)}
out["resample_us"] = t_(res)
out["materialise_theta_us"] = t_(mat)
th = box["c"].get_particles().get_choices()["theta"]
out["theta_leaf"] = {"shape": list(th.shape), "stride": list(th.stride()), "contiguous": th.is_contiguous()}
out["theta_out"] = {"shape": list(box["theta"].shape), "stride": list(box["theta"].stride())}
print(json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in out.items()}))
