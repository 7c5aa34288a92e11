"""chain-only timings of a noise-ahead sweep (the background launches left out; their buffers keep the last full run's
contents, which are the right ones): RESAMPLE=<kind> python tools/experiments/chain_only.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.smc import BootstrapSweep
n, T = 1_000_000, 100
kind = os.environ.get("RESAMPLE", "multinomial_sorted")
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
sw = BootstrapSweep(init, step, n, T, resample=kind).prepare(G.key(314159), torch.from_numpy(ys))
from ctypes import c_void_p
from genjax_amd import _lib
be = _lib.get()
def t_(fn, reps=5):
    """`fn` captured into a hipGraph (the host out of the loop), replayed `reps` times: us per step"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        be.check(be.c.gmx_capture_begin(be.stream()), "capture")
        try:
            fn()
        finally:
            g = c_void_p(); rc = be.c.gmx_capture_end(be.stream(), g)
        be.check(rc, "capture_end")
        be.check(be.c.gmx_graph_launch(g, be.stream()), "graph"); side.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps): be.check(be.c.gmx_graph_launch(g, be.stream()), "graph")
        e1.record(side); side.synchronize()
        be.c.gmx_graph_destroy(g)
    torch.cuda.current_stream().wait_stream(side)
    return 1e3 * e0.elapsed_time(e1) / reps / T
out = {"kind": kind}
out["sweep_eager"] = t_(lambda: sw.enqueue())
out["chain_only"] = t_(lambda: sw._enqueue_noise_ahead(skip_noise=True))
out["chain_only_without_site_program"] = t_(lambda: sw._enqueue_noise_ahead(skip_vm=True, skip_noise=True))
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
