"""debug driver: tiny populations on the device, one-stream / noise-ahead, against the oracle and the Kalman evidence"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.smc import BootstrapSweep
from tests import parity
for n in (32, 1000, 1024, 3000):
    for na in (False, True):
        res = parity.check_lgssm_sweep(n=n, T=8, noise_ahead=na)
        print(n, na, {k: res[k] for k in ("ancestors_equal", "x_equal", "totals_equal", "lw_max_abs_diff", "log_ml", "log_ml_oracle")}, flush=True)
T, N = 8, 32
ys = workloads.lgssm_data(T); kal = workloads.kalman_log_ml(ys)
init, step = workloads.make_lgssm(G)
sw = BootstrapSweep(init, step, N, T)
for r in range(4):
    sw.prepare(G.key(5000 + r), torch.from_numpy(ys)); sw.launch()
    print("repeat prepare", r, sw.log_ml() - kal, sw.noise_ahead, flush=True)
for r in range(4):
    sw2 = BootstrapSweep(init, step, N, T).prepare(G.key(5000 + r), torch.from_numpy(ys)); sw2.launch()
    print("fresh sweep   ", r, sw2.log_ml() - kal, flush=True)
