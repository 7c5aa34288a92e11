# NOTE: needs the multi-step launch of profiles/r04l_multi_step_attempt.patch applied (git apply); the product does not carry it
# (measured slower than one launch per step: profiles/r04l_multi_step_results.json).
"""Where a step's time goes with several steps per launch (BootstrapSweep(steps_per_launch=K)): the whole sweep and the
chain alone (no noise launches: the chain then reads whatever the noise buffers hold), each as a captured graph, for
K in argv (default 1 10).  One JSON line per K."""
import json, os, sys, time
from ctypes import c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import genjax_amd as G
from genjax_amd import _lib, workloads
from genjax_amd.inference.smc import BootstrapSweep

n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
be = _lib.get()


def graph_time(fn, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        be.check(be.c.gmx_capture_begin(be.stream()), "capture")
        try:
            fn()
        finally:
            g = c_void_p()
            rc = be.c.gmx_capture_end(be.stream(), g)
        be.check(rc, "capture_end")
        be.check(be.c.gmx_graph_launch(g, be.stream()), "graph"); side.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            be.check(be.c.gmx_graph_launch(g, be.stream()), "graph"); side.synchronize()
            best = min(best, time.perf_counter() - t0)
        be.c.gmx_graph_destroy(g)
    torch.cuda.current_stream().wait_stream(side)
    return best


for K in [int(a) for a in (sys.argv[1:] or ["1", "10"])]:
    G.clear_caches()
    sw = BootstrapSweep(init, step, n, T, steps_per_launch=K).prepare(G.key(314159), torch.from_numpy(ys))
    out = {"steps_per_launch": K, "n": n, "T": T,
           "sweep_us_per_step": 1e6 * graph_time(sw.enqueue) / T,
           "chain_only_us_per_step": 1e6 * graph_time(lambda: sw._enqueue_noise_ahead(skip_noise=True)) / T}
    print(json.dumps(out), flush=True)
