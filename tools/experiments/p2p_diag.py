"""diagnostic: the sharded sweep at world size 1 over the peer-mapped communicator"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["GENMI_COMM"] = "p2p"
import numpy as np, torch
import genjax_amd as G
from genjax_amd import _lib, workloads
from genjax_amd.inference.sharded import ShardedBootstrapSweep
class Solo:
    @staticmethod
    def get_rank(): return 0
    @staticmethod
    def get_world_size(): return 1
n, T = 300_000, 5
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
for na in (False, True):
    sw = ShardedBootstrapSweep(init, step, n, T, Solo, always_communicate=True, noise_ahead=na).prepare(G.key(314159), torch.from_numpy(ys))
    sw.launch(); torch.cuda.synchronize()
    print("na", na, "state", sw.cx.state.tolist(), "log_ml", sw.log_ml(), "landing sizes", list(sw.cx._landing))
    t0 = time.perf_counter()
    for _ in range(5): sw.launch()
    torch.cuda.synchronize()
    print("  eager us/step", 1e6 * (time.perf_counter() - t0) / 5 / T, sw.cx.state.tolist())
    x_eager = sw.state().clone()
    sw.capture()
    sw.launch(); sw.finish(); torch.cuda.synchronize()
    print("  graph equal", bool(torch.equal(sw.state(), x_eager)), sw.cx.state.tolist(), sw.log_ml())
    t0 = time.perf_counter()
    for _ in range(5): sw.launch()
    torch.cuda.synchronize()
    print("  graph us/step", 1e6 * (time.perf_counter() - t0) / 5 / T, sw.cx.state.tolist())
    sw.close()
