"""Cost of one nested `IndexRequest(j, IndexRequest(t, sub))` on a plate of long scans held per particle (VERDICT r5 item
8): the O(1) form (combinators._vmap_edit_index_o1 around _scan_edit_index_o1: element j sliced out, steps t and t + 1 of
it edited, written back lazily) against the counted-loop form (all J x T steps of every particle re-run under a gate).
Usage: python tools/nested_index_request_cost.py [n] [J] [T] > out.json     (leaves are [n, J, T] f32: 4 n J T bytes each)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import genjax_amd as G  # noqa: E402
from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Update, _lib, numpy as jnp  # noqa: E402


def run(n=1000, J=64, T=4096, per_particle=False):
    dev = _lib.get().device

    @G.gen
    def step(c, x):
        z = G.normal(c * 0.5 + x, 1.25) @ "z"
        G.normal(z, 0.75) @ "y"
        return z, z * 2.0

    sc = G.Scan(step, T)
    pl = sc.vmap(in_axes=(0, None))
    args = (jnp.array(np.linspace(0, 1, J).astype(np.float32)), jnp.array(np.linspace(-0.5, 0.5, T).astype(np.float32)))
    tr = pl.simulate(G.split(G.key(1), n), args)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    sync()
    out = {"n": n, "J": J, "T": T, "per_particle_indices": per_particle, "leaf_bytes": 4 * n * J * T,
           "what": "seconds per IndexRequest(j, IndexRequest(t, Update(y = 0.25))).edit on a plate of J scans of T steps over n particles"}
    if per_particle:
        rng = np.random.default_rng(0)
        j = torch.from_numpy(rng.integers(0, J, n).astype(np.int32)).to(dev)
        t = torch.from_numpy(rng.integers(0, T, n).astype(np.int32)).to(dev)
    else:
        j, t = J // 2, T // 2
    ws = {}
    for name, refuse in (("o1", False), ("loop", True)):
        if refuse:
            sc.__dict__["_o1_refused"] = True
        else:
            sc.__dict__.pop("_o1_refused", None)
        k = G.split(G.key(2), n)
        req = IndexRequest(j, IndexRequest(t, Update(C["y"].set(0.25))))
        new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
        sync()
        t0 = time.perf_counter()
        reps = 10 if not refuse else 1
        for _ in range(reps):
            new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
            _ = w[:1].cpu()
        sync()
        out[name] = (time.perf_counter() - t0) / reps
        ws[name] = (w.cpu().numpy(), new.get_score().cpu().numpy(), new.get_choices()["y"][:, :, ::97].cpu().numpy())
        del new
    sc.__dict__.pop("_o1_refused", None)
    out["same_weights_scores_values"] = bool(all(np.array_equal(a, b) for a, b in zip(ws["o1"], ws["loop"])))
    out["speedup"] = out["loop"] / out["o1"]
    return out


if __name__ == "__main__":
    _lib.install(None)
    a = [int(x) for x in sys.argv[1:4]]
    print(json.dumps(run(*a, per_particle="--per-particle" in sys.argv)))
