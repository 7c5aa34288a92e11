"""The reference's `docs/cookbook/inactive/update/4_index_request.ipynb` (c3, c9, c12) on this build: ONE trace of a model
with three n-element bare-distribution plates and an observation of their sums; milliseconds for importance, for an
Update of the whole plate "a" and for `StaticRequest({"a": IndexRequest(3, Update(42.0))})`, n = 1e4 .. 1e8."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, StaticRequest, Update, numpy as jnp


def make(n):
    zeros, ones = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")

    @G.gen
    def model():
        x = G.normal(0.0, 1.0) @ "x"
        a = G.normal.vmap()(zeros, ones) @ "a"
        b = G.normal.vmap()(zeros, ones) @ "b"
        c = G.normal.vmap()(zeros, ones) @ "c"
        obs = G.normal(jnp.sum(a) + jnp.sum(b) + jnp.sum(c) + x, 5.0) @ "obs"
        return obs
    return model


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


for n in [int(float(a)) for a in (sys.argv[1:] or ["1e4", "1e5", "1e6", "1e7", "1e8"])]:
    G.clear_caches()
    model = make(n)
    tr, _ = model.importance(G.key(0), C["obs"].set(1.0), ())
    new_a = torch.ones(n, device="cuda")
    out = {"n": n,
           "ms_importance": timed(lambda: model.importance(G.key(0), C["obs"].set(1.0), ())),
           "ms_update_whole_plate": timed(lambda: tr.update(G.key(1), C["a"].set(new_a), Diff.no_change(()))),
           "ms_index_request": timed(lambda: StaticRequest({"a": IndexRequest(3, Update(C.v(42.0)))}).edit(G.key(2), tr, Diff.no_change(())))}
    new, w, _, _ = StaticRequest({"a": IndexRequest(3, Update(C.v(42.0)))}).edit(G.key(2), tr, Diff.no_change(()))
    out["one_value_changed"] = int((new.get_choices()["a"] == 42.0).sum().item()) == 1
    print(json.dumps(out), flush=True)
    del tr, new
    torch.cuda.empty_cache()
