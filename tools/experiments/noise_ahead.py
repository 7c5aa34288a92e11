"""EXPERIMENT (the product form of this is BootstrapSweep's noise-ahead path, inference/smc.py; the measurements that
led to it: profiles/r02f_experiment_noise_ahead_*.txt): does drawing step t+1's standard-normal noise in a SEPARATE kernel on a second stream —
concurrently with step t's resampling — shorten the bootstrap sweep?  The whole step is bound by vector-instruction
issue, so this only helps if the runtime overlaps the noise kernel with the kernel boundaries and
load / store phases of the dependent chain  [site program' -> offspring].

  chain  (stream A):  J_t: x = 0.9 x_prev[anc] + 0.5 z_t ; lw = log N(y_t; x, 1) ; tile stats   ->  O_t: ancestors
  noise  (stream B):  N_t: z_t = sqrt(2) erfinv(bits(fold_in(split(k_prop_t, n)[i], 1)))        (keys only)

Same keys, same float operations as the fused site program (x = z * sx, then + a * x_prev), so states / ancestors /
evidence must equal BootstrapSweep's bit for bit — checked below.  Prints one JSON line with both timings.
"""
import json
import os
import sys
import time
from ctypes import c_uint32

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import genjax_amd as G
from genjax_amd import _lib, workloads
from genjax_amd.core.choice_map import ChoiceMap
from genjax_amd.engine import Gathered
from genjax_amd.inference.smc import BootstrapSweep, cdf_shift
from genjax_amd.random import fold_in, lazy_split, split
from genjax_amd.static import MinimalGenerate

n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
be = _lib.get()
dev = be.device
ys_np = workloads.lgssm_data(T)
ys = torch.from_numpy(ys_np).to(dev)
key = G.key(314159)
init, step = workloads.make_lgssm(G)

# reference: the product's fused sweep
ref = BootstrapSweep(init, step, n, T).prepare(key, torch.from_numpy(ys_np))
ref.capture()
ref.launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ref.launch()
torch.cuda.synchronize()
dt_ref = (time.perf_counter() - t0) / 20
x_ref, lw_ref, anc_ref = [v.clone() for v in ref.state()]


@G.gen
def noise():
    return G.normal(0.0, 1.0) @ "x"


@G.gen
def j_init(z):
    x = z * 1.0
    G.normal(x, 1.0) @ "y"
    return x


@G.gen
def j_step(x_prev, z):
    v = z * 0.5
    x = v + 0.9 * x_prev
    G.normal(x, 1.0) @ "y"
    return x


shift = cdf_shift(n)
RING = int(os.environ.get("RING", 2))     # noise buffers: the noise stream may run RING steps ahead of the chain
z = [torch.zeros((1, n), dtype=torch.float32, device=dev) for _ in range(RING)]
xs = [torch.zeros((1, n), dtype=torch.float32, device=dev) for _ in range(2)]
lw = torch.zeros((n,), dtype=torch.float32, device=dev)
wdummy = torch.zeros((1, n), dtype=torch.float32, device=dev)
anc = torch.zeros((n,), dtype=torch.int32, device=dev)
partials = torch.zeros((2, (n + 255) // 256), dtype=torch.float32, device=dev)
npart = torch.zeros((2, (n + 255) // 256), dtype=torch.float32, device=dev)
tile_agg = torch.zeros(((n + 1023) // 1024,), dtype=torch.int64, device=dev)
maxs = torch.zeros((T,), dtype=torch.float32, device=dev)
totals = torch.zeros((T,), dtype=torch.int64, device=dev)
obs0 = ChoiceMap.empty().set("y", ys[0])
pN = MinimalGenerate(noise, (), ChoiceMap.empty(), (n,))
pJ0 = MinimalGenerate(j_init, (z[0].reshape(n),), obs0, (n,))
pJ = MinimalGenerate(j_step, (Gathered(xs[0].reshape(n), anc), z[0].reshape(n)), obs0, (n,))
# PAD: the noise program as a BACKGROUND program (gmx_program_set_background): wave priority 0 and PAD bytes of dynamic
# LDS every noise workgroup asks for and never touches — a cap on how many noise workgroups a CU holds (160 KB of LDS
# per CU), so that the chain's kernels always find wave slots.  PAD=0: an ordinary program.
PAD = int(os.environ.get("PAD", 0))
if PAD:
    pN.comp.set_background(PAD)
pN.comp.specialize()
for p in (pJ0, pJ):
    p.comp.specialize()
assert pJ.comp.writes_tile_stats() and pJ0.comp.writes_tile_stats()
keys = [split(fold_in(key, t), 3) for t in range(T)]


def launch_noise(t):
    bufs = [None] * len(pN.comp.outputs)
    bufs[pN.ro[1]] = z[t % RING]
    if pN.wo[0] == "out":
        bufs[pN.wo[1]] = wdummy
    pN.comp.run(pN.leaves((), ChoiceMap.empty()), (n,), lazy_split(keys[t][0], n), red_out=npart, out_buffers=bufs)


zfull = None


def zsrc(t):
    return zfull[t] if zfull is not None else z[t % RING]


def launch_j(t):
    obs = ChoiceMap.empty().set("y", ys[t])
    if t == 0:
        prog, leaves = pJ0, pJ0.leaves((zsrc(0).reshape(n),), obs)
    else:
        prog = pJ
        leaves = prog.leaves((Gathered(xs[(t - 1) % 2].reshape(n), anc), zsrc(t).reshape(n)), obs)
    bufs = [None] * len(prog.comp.outputs)
    bufs[prog.ro[1]] = xs[t % 2]
    bufs[prog.wo[1]] = lw.reshape(1, n)
    prog.comp.run(leaves, (n,), None, red_out=partials, out_buffers=bufs, tile_stats=(tile_agg, shift))


def launch_o(t):
    kh = keys[t][1].host()
    kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
    be.check(be.c.gmx_resample_tiles(0, kk, be.ptr(lw), n, shift, be.ptr(partials), be.ptr(tile_agg),
                                     be.ptr(maxs[t:t + 1]), be.ptr(totals[t:t + 1]), be.ptr(anc), be.stream()), "resample")


BATCH = int(os.environ.get("BATCH", 1))   # > 1: the noise of BATCH steps is one group of launches, one event pair per group
GROUP = int(os.environ.get("GROUP", 0))   # 1: ... and ONE launch per group (BATCH * n rows, keys split(step key, n)[i] by row)
if GROUP:
    from genjax_amd.random import Key
    assert RING == 2 * BATCH and T % BATCH == 0
    zg = [torch.zeros((1, BATCH * n), dtype=torch.float32, device=dev) for _ in range(2)]
    z = [zg[(r // BATCH) % 2][:, (r % BATCH) * n:(r % BATCH + 1) * n] for r in range(RING)]
    wdummy_g = torch.zeros((1, BATCH * n), dtype=torch.float32, device=dev)
    npart_g = torch.zeros((2, (BATCH * n + 255) // 256), dtype=torch.float32, device=dev)
    gkeys, gkeys_dev = [], []
    for g in range(T // BATCH):
        rows = np.stack([keys[t][0].host() for t in range(g * BATCH, (g + 1) * BATCH)]).astype(np.uint32)
        gkeys_dev.append(torch.from_numpy(rows.view(np.int32)).to(dev))
        gkeys.append(Key(lazy=("rowsplit", Key(dev=torch.from_numpy(rows.view(np.int32)).to(dev)), n), split_last=True))


def launch_noise_group(g):
    bufs = [None] * len(pN.comp.outputs)
    bufs[pN.ro[1]] = zg[g % 2]
    if pN.wo[0] == "out":
        bufs[pN.wo[1]] = wdummy_g
    pN.comp.run(pN.leaves((), ChoiceMap.empty()), (BATCH * n,), gkeys[g], red_out=npart_g, out_buffers=bufs)



def enqueue_batched():
    # RING = 2 * BATCH buffers: group g of noise launches fills half g % 2 while the chain consumes the other half
    assert RING == 2 * BATCH and T % BATCH == 0
    A = torch.cuda.current_stream()
    B = torch.cuda.Stream(priority=0)
    B.wait_stream(A)
    groups = T // BATCH
    done = [None] * groups
    ready = [None] * groups

    def noise_group(g):
        with torch.cuda.stream(B):
            if g >= 2:
                B.wait_event(done[g - 2])
            if GROUP:
                launch_noise_group(g)
            else:
                for t in range(g * BATCH, (g + 1) * BATCH):
                    launch_noise(t)
            ready[g] = torch.cuda.Event()
            ready[g].record(B)

    noise_group(0)
    for g in range(groups):
        if g + 1 < groups:
            noise_group(g + 1)
        A.wait_event(ready[g])
        for t in range(g * BATCH, (g + 1) * BATCH):
            launch_j(t)
            launch_o(t)
        done[g] = torch.cuda.Event()
        done[g].record(A)
    A.wait_stream(B)


def enqueue(two_streams: bool):
    if two_streams and BATCH > 1:
        return enqueue_batched()
    A = torch.cuda.current_stream()
    B = torch.cuda.Stream() if two_streams else A
    done_j = [None] * T
    ready_z = [None] * T
    if two_streams:
        B.wait_stream(A)
    for t in range(T):
        with torch.cuda.stream(B):
            if two_streams and t >= RING:
                B.wait_event(done_j[t - RING])          # z[t % RING] is free once J_{t-RING} has read it
            launch_noise(t)
            if two_streams:
                ready_z[t] = torch.cuda.Event()
                ready_z[t].record(B)
        if two_streams:
            A.wait_event(ready_z[t])
        launch_j(t)
        if two_streams:
            done_j[t] = torch.cuda.Event()
            done_j[t].record(A)
        launch_o(t)
    if two_streams:
        A.wait_stream(B)


out = {"n": n, "T": T, "ring": RING, "batch": BATCH, "group_launch": GROUP,  "lds_pad": PAD, "fused_sweep_us_per_step": 1e6 * dt_ref / T}
for two in (False, True):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enqueue(two)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    same = bool(torch.equal(xs[(T - 1) % 2].reshape(n), x_ref) and torch.equal(anc, anc_ref) and torch.equal(lw, lw_ref))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enqueue(two)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["split_two_streams" if two else "split_one_stream"] = {"us_per_step": 1e6 * dt / T, "bit_identical_to_fused": same}
if int(os.environ.get("CHAIN_ONLY", 0)):
    # the dependent chain alone (J_t, O_t on one stream), every step's noise already in memory: what the two-stream
    # form would cost if the noise were free
    zfull = [torch.zeros((1, n), dtype=torch.float32, device=dev) for _ in range(T)]
    zsave, z = z, zfull
    RING_save, RING = RING, T
    for t in range(T):
        launch_noise(t)
    torch.cuda.synchronize()

    def chain():
        for t in range(T):
            launch_j(t)
            launch_o(t)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    same = bool(torch.equal(xs[(T - 1) % 2].reshape(n), x_ref) and torch.equal(anc, anc_ref))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    out["chain_only"] = {"us_per_step": 1e6 * (time.perf_counter() - t0) / 20 / T, "bit_identical_to_fused": same}
print(json.dumps(out))
