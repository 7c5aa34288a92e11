"""us / step of the captured config-2 sweep for the resampling kinds (KINDS=a,b,...; one JSON line)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.smc import BootstrapSweep
n, T = 1_000_000, 100
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
out = {}
for kind in (os.environ.get("KINDS", "systematic,stratified,multinomial,multinomial_tiled,multinomial_sorted")).split(","):
    sw = BootstrapSweep(init, step, n, T, resample=kind).prepare(G.key(314159), torch.from_numpy(ys)).capture()
    for _ in range(3): sw.launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sw.launch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    out[kind] = {"us_per_step": 1e6 * dt / T, "log_ml": sw.log_ml(), "kalman": workloads.kalman_log_ml(ys)}
print(json.dumps(out))
