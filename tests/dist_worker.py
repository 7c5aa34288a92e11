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
# what the parent test asks of this run (one JSON object): on_gpu, noise_ahead, capture, cdf_form, fused
OPTS = json.loads(os.environ.get("GENMI_TEST_OPTS", "{}"))


def schools_main(out_path, k_per_rank, capacity):
    """BASELINE config 4 sharded: 8-schools ImportanceK + one global systematic resample."""
    dist.init_process_group("gloo")
    import tests.hostsim as hs
    hs.install()
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    from genjax_amd.inference.sharded import sharded_importance_resample
    from tests import parity

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        log_tau = G.normal(0.0, 1.0) @ "log_tau"
        theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
        _ = G.normal(theta, jnp.array(parity.SCHOOL_SIGMA)) @ "y"
        return theta
    info = {}
    comm = None
    if os.environ.get("GENMI_COMM") == "p2p":
        from genjax_amd.inference.comm import make_comm
        comm = make_comm(dist, G._lib.get().device)
    coll, lw = sharded_importance_resample(G.Target(schools, (), C["y"].set(parity.SCHOOL_Y)), k_per_rank, G.key(2), dist,
                                           kind=OPTS.get("resample", "systematic"), capacity=capacity, stats=info,
                                           comm=comm)
    ch = coll.get_particles().get_choices()
    outs = {}
    for name in ("theta", "mu"):
        v = ch[name].contiguous()
        parts = [torch.empty_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, v)
        outs[name] = torch.cat(parts).numpy()
    lws = [torch.empty_like(lw) for _ in range(dist.get_world_size())]
    dist.all_gather(lws, lw)
    if dist.get_rank() == 0:
        np.savez(out_path + ".npz", lw=torch.cat(lws).numpy(), log_ml=float(coll.get_log_marginal_likelihood_estimate()), **outs)
        json.dump(info, open(out_path + ".info.json", "w"))
    dist.destroy_process_group()


def main(out_path, n_per_rank, T, capacity=None, mh=False, vec=False, vecmh=False):
    dist.init_process_group("gloo")
    on_gpu = bool(OPTS.get("on_gpu"))
    if on_gpu:
        # every rank on THE one GPU of the box (tests/test_gpu_parity.py: the peer-mapped exchange between two
        # processes through IPC handles, on real device memory): the HIP library, no CPU mirror
        torch.cuda.set_device(0)
    else:
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
                                   step_extra=lambda t: (float(t),), resample=OPTS.get("resample", "systematic")).prepare(
            G.key(7), torch.from_numpy(ys))
    elif vecmh:  # a D-vector state in one vector-valued site + one MH move per step: 2 x D routed leaves
        from genjax_amd import numpy as jnp
        from tests import parity
        init, step = parity.make_vec_mh(G, lambda *v: jnp.stack(list(v)), jnp.ones(vecmh))
        req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.2))})
        sw = ShardedBootstrapSweep(init, step, n_per_rank, T, dist, capacity=capacity, rejuvenate=req,
                                   step_extra=lambda t: (float(t),)).prepare(G.key(11), torch.from_numpy(parity.tracker_data(T)))
    elif vec:   # a 2-vector state (position, velocity): one routed leaf per component
        from genjax_amd import numpy as jnp
        from tests import parity
        init, step = parity.make_tracker(G, lambda a, b: jnp.stack([a, b]))
        sw = ShardedBootstrapSweep(init, step, n_per_rank, T, dist, capacity=capacity).prepare(
            G.key(5), torch.from_numpy(parity.tracker_data(T)))
    else:
        ys = workloads.lgssm_data(T)
        init, step = workloads.make_lgssm(G)
        # noise_ahead: the step's draws by background programs keyed by the global particle index
        na = True if OPTS.get("noise_ahead") else (False if on_gpu else None)
        sw = ShardedBootstrapSweep(init, step, n_per_rank, T, dist, capacity=capacity, noise_ahead=na,
                                   cdf_form=bool(OPTS.get("cdf_form")), fused=bool(OPTS.get("fused", 1)),
                                   resample=OPTS.get("resample", "systematic")).prepare(
            G.key(314159), torch.from_numpy(ys))
        assert sw.noise_ahead == bool(na)
    sw.launch()
    sw.finish()
    if OPTS.get("capture"):      # the same sweep again as ONE captured graph, replayed twice
        first = sw.state().clone()
        sw.capture()
        for _ in range(2):
            sw.launch()
            sw.finish()
        assert torch.equal(first, sw.state())
    st = sw.state().cpu()
    xs = [torch.empty_like(st) for _ in range(dist.get_world_size())]
    dist.all_gather(xs, st)
    if dist.get_rank() == 0:
        np.save(out_path + ".npy", torch.cat(xs).numpy())
        json.dump({"log_ml": sw.log_ml(), "totals": [str(t) for t in sw.totals.cpu().numpy().view(np.uint64).tolist()],
                   "maxs": sw.maxs.cpu().tolist(), "reruns": sw.reruns, "capacity": sw.capacity,
                   "communicator": sw.cx.name if sw.cx is not None else None,
                   "one_launch_per_step": bool(getattr(sw, "fuse_sh", False)),
                   "chained_mh": bool(getattr(sw, "chain_mh", False))},
                  open(out_path + ".json", "w"))
    if on_gpu and sw.cx is not None:
        sw.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 5 and sys.argv[5] == "schools":
        schools_main(sys.argv[1], int(sys.argv[2]), int(sys.argv[4]) if int(sys.argv[4]) > 0 else None)
        sys.exit(0)
    cap = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else None
    mode = sys.argv[5] if len(sys.argv) > 5 else ""
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), cap, mh=mode == "mh", vec=mode == "vec", vecmh={"vecmh": 2, "vec6mh": 6}.get(mode, 0))
