"""After every link of every chain of launches of one fuzz seed: the outputs and scratch rows that link wrote, saved to
gpurun_out/snap_<backend>_<seed>.npz — run once on the HIP library and once on the CPU mirror, then diff
(tools/experiments/chain_snapshot.py diff <seed>): the first differing (program, link, buffer) names the launch that
computes something else on the device.   python tools/experiments/chain_snapshot.py <hip|mirror|diff> <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

mode, seed = sys.argv[1], int(sys.argv[2])
OUT = os.path.join(ROOT, "gpurun_out")
if mode == "diff":
    a = np.load(os.path.join(OUT, f"snap_hip_{seed}.npz"))
    b = np.load(os.path.join(OUT, f"snap_mirror_{seed}.npz"))
    bad = 0
    for k in sorted(a.files, key=lambda s: [int(x) for x in s.split("_")[1::2]]):
        if k not in b.files:
            print("only on the device:", k); continue
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        if not same:
            bad += 1
            if bad <= 12:
                d = np.argwhere(a[k] != b[k])
                print("DIFFERENT", k, a[k].shape, "first at", d[:2].tolist() if d.size else None, "hip", a[k].reshape(-1)[:4], "mirror", b[k].reshape(-1)[:4])
    print("buffers compared", len(a.files), "different", bad)
    sys.exit(0)
import torch
if mode == "mirror":
    import tests.hostsim as hs
    hs.install()
from genjax_amd import _lib, engine
from tests import fuzz_models as F
be = _lib.get()
snap = {}
n_prog = [0]
orun = engine.Compiled.run


def run(self, leaves, batch, key, *a, **k):
    if not self.links:
        return orun(self, leaves, batch, key, *a, **k)
    n_prog[0] += 1
    n, args, keep, outs = self.bind(leaves, batch, key, *a, **k)
    scratch = keep[-1]
    slot_to_out = {}
    for ko, (dt, event, slots) in enumerate(self.outputs):
        sl = slots[1] if isinstance(slots, tuple) else slots
        for s in (sl if isinstance(sl, list) else [sl]):
            slot_to_out[s] = ko
    for li, (link, A) in enumerate(zip(self.links, args)):
        be.check(be.c.gmx_program_run(link.handle, n, A, be.stream()), "gmx_program_run")
        if be.uses_streams:
            torch.cuda.synchronize()
        done = set()
        for kind, x in link.out_dst:
            if kind == "spill":
                snap[f"p_{n_prog[0]}_l_{li}_spill_{x}"] = scratch[x].cpu().numpy().copy()
            elif slot_to_out[x] not in done:
                done.add(slot_to_out[x])
                snap[f"p_{n_prog[0]}_l_{li}_out_{slot_to_out[x]}"] = outs[slot_to_out[x]].cpu().numpy().copy()
    return outs


engine.Compiled.run = run
try:
    F.run_one(seed)
    print("run_one ok")
except Exception as e:      # noqa: BLE001
    print("run_one:", repr(e)[:120])
np.savez(os.path.join(OUT, f"snap_{'hip' if mode == 'hip' else 'mirror'}_{seed}.npz"), **snap)
print("saved", len(snap), "buffers")
