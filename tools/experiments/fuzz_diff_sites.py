"""Which sites of a fuzz model differ from the oracle after `importance` on the HIP library (values and per-site scores):
python tools/experiments/fuzz_diff_sites.py <seed> [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import genjax_amd as G
from genjax_amd import _lib, numpy as jnp
from genjax_amd.engine import materialize
from oracle import genjax_oracle as O
O.build()
from tests import fuzz_models as F

seed = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 7
from genjax_amd import engine
_CHAINS = []
_oi = engine.Compiled.__init__


def _init(self, tr, chain=False):
    _oi(self, tr, chain)
    if self.links:
        _CHAINS.append(self)


engine.Compiled.__init__ = _init
rng = np.random.default_rng(seed)
spec = F.random_spec(rng)
dev = _lib.get().device
model, omodel = F.build(G, spec, float), F.build(O, spec, np.float32)
a, extra = F.spec_args(spec, rng, B)
ga = tuple([torch.from_numpy(a).to(dev)] + [jnp.array(e) if not (e.shape[:1] == (B,) and e.dtype == bool and e.ndim == 1) else torch.from_numpy(e).to(dev) for e in extra])
k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
tr, otr = model.simulate(k, ga), omodel.simulate(ok, (a,) + tuple(extra))
print("simulate score equal:", np.array_equal(F._np(tr.get_score()), otr.get_score()))
cons = F._pick_constraints(spec, rng, 0.5, B)
tri, w = model.importance(k, F._g_constraint(G, cons), ga)
otri, ow = omodel.importance(ok, F._o_constraint(cons), (a,) + tuple(extra))
print("importance weight equal:", np.array_equal(F._np(w), np.broadcast_to(ow, (B,))), "score equal:", np.array_equal(F._np(tri.get_score()), otri.get_score()))
print("constrained:", [(c[0][0], ("subset" if isinstance(c[1], tuple) else c[1].shape)) for c in cons])


def walk(t, ot, path=()):
    subs = getattr(t, "subtraces", None)
    if subs is None and hasattr(t, "inner"):
        return walk(t.inner, ot.inner, path + ("<inner>",))
    if subs is None:
        return
    for ad, st in subs.items():
        osubs = getattr(ot, "subtraces", None) or {}
        if ad not in osubs:
            print("  ", path + (ad,), "(no counterpart in the oracle's trace layout)")
            continue
        ost = osubs[ad]
        try:
            sc, osc = F._np(materialize(st.get_score())), np.asarray(ost.get_score() if hasattr(ost, "get_score") else ost.score)
            same = sc.shape == osc.shape and np.array_equal(sc, osc)
            if not same and sc.size == osc.size:
                same = np.array_equal(np.sort(sc.reshape(-1)), np.sort(osc.reshape(-1)))
                tag = "(same multiset)" if same else ""
            else:
                tag = ""
            print("  ", path + (ad,), "score", sc.shape, osc.shape, "EQUAL" if same else "DIFFERENT", tag)
        except Exception as e:      # noqa: BLE001
            print("  ", path + (ad,), "score: n/a", repr(e)[:80])
        walk(st, ost, path + (ad,))


walk(tri, otri)
sc, osc = F._np(tri.get_score()), np.asarray(otri.get_score())
print("total score product", sc, "oracle", osc, "diff", sc - osc)
tops = [F._np(materialize(st.get_score())) for st in tri.subtraces.values()]
acc = np.zeros_like(tops[0])
for t_ in tops:
    acc = (acc + t_).astype(np.float32)
print("sum of the product's own top-level scores in site order:", acc, "equal to its total:", np.array_equal(acc, sc), "to the oracle's:", np.array_equal(acc, osc))
# the code hashes of the specialised links of every chain (to find their code objects in GENMI_JIT_CACHE)
for i, c in enumerate(_CHAINS):
    print("chain", i, [(l.n_regs, "%016x" % int(_lib.get().c.gmx_program_code_hash(l.handle))) for l in c.links])
