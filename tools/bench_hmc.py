"""HMC edit timing: 1e6 chains of the reference's test model (x ~ N(0,1), y ~ N(x, 0.01) observed),
L = 10 leapfrog steps per edit, ONE launch per edit.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import genjax_amd as G
from genjax_amd import ChoiceMap, Diff, SelectionBuilder as S
from genjax_amd.inference.requests import HMC


@G.gen
def model():
    x = G.normal(0.0, 1.0) @ "x"
    y = G.normal(x, 0.01) @ "y"
    return y


n = int(os.environ.get("N", 1_000_000))
tr, _ = model.importance(G.split(G.key(0), n), ChoiceMap.kw(y=3.0), ())
req = HMC(S["x"], 1e-2, L=10)
cur = tr
for r in range(3):
    cur, *_ = req.edit(G.split(G.key(10 + r), n), cur, Diff.no_change(()))
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for r in range(reps):
    cur, w, *_ = req.edit(G.split(G.key(100 + r), n), cur, Diff.no_change(()))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": "HMC edit, L=10, 1-site normal model", "chains": n, "ms_per_edit": 1e3 * dt,
                  "leapfrog_steps_per_s": n * 10 / dt, "gradient_evals_per_s": n * 11 / dt,
                  "mean_x_after_23_edits": float(cur.get_choices()["x"].mean())}))
