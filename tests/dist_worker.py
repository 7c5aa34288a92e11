"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo
job running ShardedBootstrapSweep on the CPU harness; rank 0 gathers the
resampled particles and writes them for the parent test to compare."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, n_per_rank, T, capacity=None, mh=False):
    dist.init_process_group("gloo")
    import tests.hostsim as hs
    hs.install()
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference.sharded import ShardedBootstrapSweep
    if mh:      # BASELINE config 3: nonlinear SSM with one Gaussian-drift MH move per step
        ys = workloads.nlssm_data(T)
        init, step = workloads.make_nlssm(G)
        req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
        sw = ShardedBootstrapSweep(init, step, n_per_rank, T, dist, capacity=capacity, rejuvenate=req,
                                   step_extra=lambda t: (float(t),)).prepare(G.key(7), torch.from_numpy(ys))
    else:
        ys = workloads.lgssm_data(T)
        init, step = workloads.make_lgssm(G)
        sw = ShardedBootstrapSweep(init, step, n_per_rank, T, dist, capacity=capacity).prepare(G.key(314159), torch.from_numpy(ys))
    sw.launch()
    xs = [torch.empty_like(sw.state()) for _ in range(dist.get_world_size())]
    dist.all_gather(xs, sw.state())
    if dist.get_rank() == 0:
        np.save(out_path + ".npy", torch.cat(xs).numpy())
        json.dump({"log_ml": sw.log_ml(), "totals": [str(t) for t in sw.totals.numpy().view(np.uint64).tolist()],
                   "maxs": sw.maxs.tolist(), "reruns": sw.reruns, "capacity": sw.capacity},
                  open(out_path + ".json", "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    cap = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else None
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), cap, mh=len(sys.argv) > 5 and sys.argv[5] == "mh")
