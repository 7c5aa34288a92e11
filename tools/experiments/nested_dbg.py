"""debug driver: a plate of scans on the device, mode by mode (simulate / importance / assess), interpreter and JIT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import engine, ChoiceMapBuilder as C, numpy as jnp
jit = os.environ.get("DBG_JIT", "1") == "1"
engine.JIT_MIN_PARTICLES = 1024 if jit else 1 << 40
engine.JIT_MIN_WORK = 1024 if jit else 1 << 40
n, no, T = int(os.environ.get("DBG_N", 5000)), int(os.environ.get("DBG_NO", 3)), int(os.environ.get("DBG_T", 40))
rng = np.random.default_rng(0)
ys = rng.normal(0, 1, (no, T)).astype(np.float32)
x0s = np.zeros(no, np.float32)

@G.gen
def step(x, _):
    xn = G.normal(0.9 * x, 0.5) @ "x"
    G.normal(xn, 1.0) @ "y"
    return xn, xn * 2.0

@G.gen
def series(x0):
    xT, dbl = step.scan(n=T)(x0, None) @ "steps"
    return xT, dbl
model = series.vmap(in_axes=(0,))
keys = G.split(G.key(1), n)
tr = model.simulate(keys, (jnp.array(x0s),))
torch.cuda.synchronize(); print("simulate ok", tr.get_score()[:2].tolist(), flush=True)
tri, w = model.importance(keys, C["steps", "y"].set(ys), (jnp.array(x0s),))
torch.cuda.synchronize(); print("importance ok", w[:2].tolist(), flush=True)
ch = tri.get_choices()
print({k: tuple(v.shape) for k, v in [("x", ch["steps", "x"]), ("y", ch["steps", "y"])]}, flush=True)
s, _ = model.assess(ch, (jnp.array(x0s),))
torch.cuda.synchronize(); print("assess ok", s[:2].tolist(), tri.get_score()[:2].tolist(), flush=True)
