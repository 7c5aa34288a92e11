"""Cost of one `IndexRequest` on a long scan held per particle: the O(1) form
(combinators._scan_edit_index_o1: slice t, slice t + 1) against the counted-loop form (all T steps re-run under a mask).
Usage: python tools/scan_index_request_cost.py [n] [T] > out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import genjax_amd as G  # noqa: E402
from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Update, _lib, numpy as jnp  # noqa: E402

_lib.install(None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = _lib.get().device


@G.gen
def step(c, x):
    z = G.normal(c * 0.5 + x, 1.25) @ "z"
    G.normal(z, 0.75) @ "y"
    return z, z * 2.0


sc = G.Scan(step, T)
args = (torch.zeros(n, device=dev), jnp.array(np.linspace(-0.5, 0.5, T).astype(np.float32)))
tr = sc.simulate(G.split(G.key(1), n), args)
torch.cuda.synchronize()
out = {"n": n, "T": T, "what": "seconds per IndexRequest(T // 2, Update(y = 0.25)).edit on a scan of T steps over n particles"}
ws = {}
for name, refuse in (("o1", False), ("loop", True)):
    if refuse:
        sc.__dict__["_o1_refused"] = True
    else:
        sc.__dict__.pop("_o1_refused", None)
    k = G.split(G.key(2), n)
    req = IndexRequest(T // 2, Update(C["y"].set(0.25)))
    new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10 if not refuse else 3
    for _ in range(reps):
        new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
    torch.cuda.synchronize()
    out[name] = (time.perf_counter() - t0) / reps
    ws[name] = (w.cpu().numpy(), new.get_score().cpu().numpy())
out["same_weights_and_scores"] = bool(np.array_equal(ws["o1"][0], ws["loop"][0]) and np.array_equal(ws["o1"][1], ws["loop"][1]))
out["speedup"] = out["loop"] / out["o1"]
print(json.dumps(out))
