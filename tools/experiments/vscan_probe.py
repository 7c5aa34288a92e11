"""A scan whose step emits a long vector, importance with the emissions given as a launch-uniform [T, m] table: the
scan's total score, the per-step scores and the weight against the oracle — on the device interpreter (JIT=0) and
through the specialised kernel (JIT=1).   python tools/experiments/vscan_probe.py <JIT 0|1> [T] [m] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, _lib, engine, numpy as jnp
from genjax_amd.engine import materialize
from oracle import genjax_oracle as O
O.build()
jit = int(sys.argv[1]); Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 20; m = int(sys.argv[3]) if len(sys.argv) > 3 else 40
B = int(sys.argv[4]) if len(sys.argv) > 4 else 7
LIVE = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # scalar sites kept alive across the scan: more than 31 live values
if jit:
    engine.JIT_MIN_PARTICLES = 1
dev = _lib.get().device
tabm = np.linspace(0.0, 1.0, m).astype(np.float32)


@G.gen
def step(c, x):
    z = G.normal(c * 0.5 + x, 1.3) @ "z"
    G.normal(z * jnp.array(tabm), 1.25) @ "y"
    return z, z


@G.gen
def model(x0, us):
    vs = [G.normal(x0, 1.0) @ f"v{i}" for i in range(LIVE)]
    cT, _ = G.Scan(step, Tn)(x0, us) @ "s"
    for v in vs:
        cT = cT + v
    return cT


@O.gen
def ostep(c, x):
    z = O.normal((c * np.float32(0.5) + x).astype(np.float32), np.float32(1.3)) @ "z"
    O.normal((np.asarray(z, np.float32)[..., None] * tabm).astype(np.float32), np.float32(1.25)) @ "y"
    return z, z


@O.gen
def omodel(x0, us):
    vs = [O.normal(x0, np.float32(1.0)) @ f"v{i}" for i in range(LIVE)]
    cT, _ = O.Scan(ostep, Tn)(x0, us) @ "s"
    for v in vs:
        cT = (cT + v).astype(np.float32)
    return cT


rng = np.random.default_rng(0)
x0 = rng.normal(size=B).astype(np.float32); us = rng.normal(size=Tn).astype(np.float32)
ys = rng.normal(size=(Tn, m)).astype(np.float32)
k, ok = G.split(G.key(1), B), O.split(O.key(1), B)
tri, w = model.importance(k, C["s", "y"].set(ys), (torch.from_numpy(x0).to(dev), jnp.array(us)))
otri, ow = omodel.importance(ok, O.C.d({("s", "y"): ys}), (x0, us))
npy = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
print("JIT", jit, "T", Tn, "m", m, "B", B, "LIVE", LIVE, "regs", "| weight", np.array_equal(npy(w), ow), "| total score", np.array_equal(npy(tri.get_score()), otri.get_score()))
sub = tri.subtraces["s"]
print("   scan score", np.array_equal(npy(materialize(sub.get_score())), np.asarray(otri.subtraces["s"].get_score())),
      npy(materialize(sub.get_score()))[:3], np.asarray(otri.subtraces["s"].get_score())[:3])
for ad, st in sub.inner.subtraces.items():
    print("   per-step", ad, np.array_equal(npy(materialize(st.get_score())), np.asarray(otri.subtraces["s"].inner.subtraces[ad].score)))
