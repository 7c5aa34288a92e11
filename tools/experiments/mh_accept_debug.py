import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import fuzz_models as F
import genjax_amd as G
from oracle import genjax_oracle as O
from genjax_amd import Diff, Regenerate, SelectionBuilder as S, static, _lib, numpy as jnp
seed, B = 40013, 1 << 18
rng = np.random.default_rng(seed)
spec = F.random_spec(rng)
print([(s["kind"], s.get("dist"), s.get("T")) for s in spec])
dev = _lib.get().device
model, omodel = F.build(G, spec, float), F.build(O, spec, np.float32)
a, extra = F.spec_args(spec, rng, B)
ga = (torch.from_numpy(a).to(dev),) + tuple(jnp.array(e) for e in extra)
k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
tri, w = model.importance(k, G.ChoiceMapBuilder.n(), ga)
otri, ow = omodel.importance(ok, O.ChoiceMap(), (a,) + tuple(extra))
print("scores equal", np.array_equal(tri.get_score().cpu().numpy(), otri.get_score()))
ads = F.addresses(spec)
for mask in range(1, 1 << len(ads)):
    picked = [ad for i, ad in enumerate(ads) if mask >> i & 1]
    sel = None
    for path, *_ in picked:
        sel = S[path] if sel is None else sel | S[path]
    mh, acc, wm = static.run_mh(model, G.split(G.key(seed + 4000), B), tri, Regenerate(sel), Diff.no_change(ga))
    osel = O.selection(*[okey for _, okey, *_ in picked])
    omh, oacc, owm = O.rejuvenate(O.key(seed + 4000), otri, lambda k_, tr_: omodel.regenerate(k_, tr_, osel, (a,) + tuple(extra))[:2])
    acc, wm = acc.cpu().numpy(), wm.cpu().numpy()
    bad = np.nonzero(acc != oacc)[0]
    wb = np.nonzero(wm != np.broadcast_to(owm, (B,)))[0]
    print([p[0] for p in picked], "accept mismatches", len(bad), "weight mismatches", len(wb))
    for i in list(bad[:3]) + list(wb[:3]):
        print("   particle", i, "w", repr(wm[i]), repr(np.broadcast_to(owm, (B,))[i]), acc[i], oacc[i])
