import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference import smc
kind = sys.argv[1]; n, T = 50_000, 4
ys = workloads.lgssm_data(T); init, step = workloads.make_lgssm(G)
def sweep(key):
    for t in range(T):
        kp, kr, _ = G.split(G.fold_in(key, t), 3)
        obs = G.ChoiceMap.kw(y=float(ys[t]))
        if t == 0: coll = smc.ImportanceK(G.Target(init, (), obs), k_particles=n).run_smc(kp)
        else:
            coll = smc.resample(kr, coll, kind)
            coll = smc.extend(kp, coll, step, lambda tr_: (tr_.get_retval(),), obs)
    return coll
a = sweep(G.key(3)); ax = a.get_particles().get_retval().clone()
b = sweep(G.key(3)); print("eager deterministic:", torch.equal(ax, b.get_particles().get_retval()))
for na in (False, True):
    cap = smc.capture(sweep, G.key(3), noise_ahead=na)
    for r in range(3):
        c = cap.replay(); torch.cuda.synchronize()
        x = c.get_particles().get_retval()
        print("noise_ahead", na, "replay", r, "mismatches", int((x != ax).sum()))
