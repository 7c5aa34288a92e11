"""World-size-1 A/B of the sharded sweep over the fused peer exchange: one launch per step (the gathering program routes
the previous step first, gmx_run_args.sh) against two (site program + gmx_shard_step_peer).  Config 2, one hipGraph each.
Usage: GENMI_COMM=peer python tools/bench_sharded_fuse_ab.py > out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GENMI_COMM", "peer")
import torch  # noqa: E402
import genjax_amd as G  # noqa: E402
from genjax_amd import workloads  # noqa: E402
from genjax_amd.inference.sharded import ShardedBootstrapSweep  # noqa: E402


class _Solo:
    @staticmethod
    def get_rank(): return 0
    @staticmethod
    def get_world_size(): return 1


n, T = int(os.environ.get("N", 1_000_448)), 100
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
out = {"n": n, "T": T}
cx = None
only = os.environ.get("ONLY")             # ONLY=one_launch REPS=1: the run rocprofv3 wraps for that form's counters
reps = int(os.environ.get("REPS", "4"))
for name, fuse in (("two_launches", False), ("one_launch", None)):
    if only and name != only:
        continue
    sw = ShardedBootstrapSweep(init, step, n, T, _Solo, always_communicate=True, fuse_step=fuse, comm=cx).prepare(
        G.key(314159), torch.from_numpy(ys))
    cx = sw.cx
    sw.capture()
    sw.launch(); sw.finish()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(10):
            sw.launch()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    sw.finish()
    out[name] = {"us_per_step": 1e6 * best / T, "fused": bool(sw.fuse_sh), "noise_ahead": bool(sw.noise_ahead),
                 "peer_mode": bool(sw.peer_mode), "log_ml": sw.log_ml(),
                 "resident_particles": int(sw.p_step.comp.resident_particles())}
if not only:
    out["same_log_ml"] = out["two_launches"]["log_ml"] == out["one_launch"]["log_ml"]
print(json.dumps(out))
