"""What one IndexRequest on a long plate costs: a 4096-element plate x 1e5 particles, one element
regenerated / updated — milliseconds per edit, Python-int index and one index per particle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, Update, numpy as jnp

n, P = int(os.environ.get("N", 100_000)), int(os.environ.get("P", 4096))


@G.gen
def school(mu, tau, sigma):
    theta = G.normal(mu, tau) @ "theta"
    _ = G.normal(theta, sigma) @ "y"
    return theta


sig = torch.linspace(1.0, 3.0, P).cuda()
v = school.vmap(in_axes=(None, None, 0))
args = (1.0, 2.0, sig)
tr = v.simulate(G.split(G.key(1), n), args)
torch.cuda.synchronize()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {"n": n, "P": P}
keys = G.split(G.key(2), n)
from genjax_amd import static
loop = lambda req: static.run_edit(v, keys, tr, req, Diff.no_change(args))          # the counted-loop form (rounds 2-3)
out["ms_loop_form_regenerate_int_index"] = 1e3 * timed(lambda: loop(IndexRequest(7, Regenerate(S["theta"]))))
out["ms_with_score_regenerate_int_index"] = 1e3 * timed(lambda: IndexRequest(7, Regenerate(S["theta"])).edit(keys, tr, Diff.no_change(args))[0].get_score())
out["ms_regenerate_int_index"] = 1e3 * timed(lambda: IndexRequest(7, Regenerate(S["theta"])).edit(keys, tr, Diff.no_change(args)))
out["ms_update_int_index"] = 1e3 * timed(lambda: IndexRequest(7, Update(C["y"].set(0.5))).edit(keys, tr, Diff.no_change(args)))
idx = torch.randint(0, P, (n,), dtype=torch.int32).cuda()
out["ms_regenerate_index_per_particle"] = 1e3 * timed(lambda: IndexRequest(idx, Regenerate(S["theta"])).edit(keys, tr, Diff.no_change(args)))
out["ms_loop_form_regenerate_index_per_particle"] = 1e3 * timed(lambda: loop(IndexRequest(idx, Regenerate(S["theta"]))))
tr2 = tr
def sweep():
    global tr2
    tr2 = tr
    for j in range(64):
        tr2 = IndexRequest(j, Regenerate(S["theta"])).edit(keys, tr2, Diff.no_change(args))[0]
out["ms_per_edit_in_a_chain_of_64"] = 1e3 * timed(sweep, reps=1) / 64
out["ms_simulate"] = 1e3 * timed(lambda: v.simulate(keys, args))
print(json.dumps(out))
