"""host-side profile of the functional SMC API (config 3: resample -> rejuvenate -> extend per step, eager launches)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference import smc
n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
ys = workloads.nlssm_data(T)
init, step = workloads.make_nlssm(G)
req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})


def sweep(key):
    for t in range(T):
        kp, kr, km = G.split(G.fold_in(key, t), 3)
        obs = G.ChoiceMap.kw(y=float(ys[t]))
        if t == 0:
            coll = smc.ImportanceK(G.Target(init, (), obs), k_particles=n).run_smc(kp)
        else:
            coll = smc.resample(kr, coll, "systematic")
            coll = smc.rejuvenate(km, coll, req)
            coll = smc.extend(kp, coll, step, lambda tr_: (tr_.get_retval(), float(t)), obs)
    return coll


sweep(G.key(7)); sweep(G.key(8)); torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(3):
    sweep(G.key(9 + r))
torch.cuda.synchronize()
print("us/step eager:", (time.perf_counter() - t0) / 3 / T * 1e6, flush=True)
pr = cProfile.Profile(); pr.enable()
for r in range(3):
    sweep(G.key(20 + r))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
