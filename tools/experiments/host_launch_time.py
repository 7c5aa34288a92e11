import os, sys, time, json
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/genjax_amd') else os.getcwd())
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.smc import BootstrapSweep
n, T = 1_000_000, 100
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
out = {}
for name, na in (("one_stream", False), ("noise_ahead", True)):
    sw = BootstrapSweep(init, step, n, T, noise_ahead=na).prepare(G.key(314159), torch.from_numpy(ys)).capture()
    sw.launch(); torch.cuda.synchronize()
    host = []
    tot = []
    for _ in range(10):
        t0 = time.perf_counter()
        sw.launch()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    out[name] = {"host_launch_call_us": 1e6 * sorted(host)[5], "launch_to_done_us": 1e6 * sorted(tot)[5]}
print(json.dumps(out))
