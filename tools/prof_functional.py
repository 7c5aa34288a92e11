"""Host-side profile of the functional SMC API (resample -> rejuvenate -> extend) at 1e6 particles."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference import smc
n, T = 1_000_000, 30
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


sweep(G.key(1)); torch.cuda.synchronize()
t0 = time.perf_counter(); sweep(G.key(2)); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/T:.1f} us/step, until done {1e6*(t2-t0)/T:.1f} us/step")
pr = cProfile.Profile(); pr.enable(); sweep(G.key(3)); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
