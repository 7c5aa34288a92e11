"""Child process of tests/test_sanitizers.py: started with the ASan runtime preloaded and GENMI_HOSTSIM_SO /
GENMI_ORACLE_SO pointing at the -fsanitize=address,undefined builds of tests/hostsim/hostsim.cpp and
oracle/orc_core.c.  Drives the C code through the same parity workloads the CPU suite uses (site-program
interpreter, samplers, integer CDF, exact ancestors, routing, MH, plates, Dirichlet) plus the OpenMP sweep;
any sanitizer report aborts the process with a non-zero status."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import tests.hostsim as hs  # noqa: E402
from oracle import genjax_oracle as O  # noqa: E402
from tests import parity  # noqa: E402

hs.install()
assert "san" in os.environ["GENMI_HOSTSIM_SO"] and "san" in os.environ["GENMI_ORACLE_SO"]

res = parity.check_lgssm_sweep(n=3000, T=4)
assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"], res
res = parity.check_lgssm_sweep(n=2049, T=3)
assert res["ancestors_equal"] and res["x_equal"], res
parity.check_nlssm_mh(n=700, T=3)
parity.check_nlssm_mh_sweep(n=1100, T=4, want_chained=True)     # MH move chained into the extension (OP_KSPLITU)
parity.check_nlssm_mh_sweep(n=1100, T=4, want_chained=False, chain_mh=False)     # ... and as two launches
parity.check_plates(n=129)
parity.check_plate_of_scans(n=33, no=24, T=40)         # two nested counted loops, [n, A, T] step leaves (GMX_F_FLAT)
parity.check_plate_of_scans(n=33, no=3, T=40)          # an unrolled plate around its elements' loops
parity.check_multinomial_sorted(n=3333, seed=7, spike=30.0, rows=2)      # the order-statistics table and its reader
parity.check_multinomial_sorted(n=1, seed=9)
parity.check_multinomial_tiled(n=2500, seed=8)
parity.check_nested_index_edits(3, 24)                     # gated edits inside two nested loops
parity.check_csmc(k=65)
parity.check_nested_marginal(k=33)
parity.check_dirichlet(n=300)
parity.check_shard_route(n=1024, world=2)
parity.check_mixture_assignments(n=300, K=8)

# the fixed-point weight (gmx_exp_fixed, through the mirror's tile statistics) on hostile log-weights
import torch  # noqa: E402
from genjax_amd import _lib  # noqa: E402
be = _lib.get()
bad = np.array([np.nan, np.inf, -np.inf, 1e38, -1e38, 0.0, -0.0, 88.0, -87.4, -103.0, 1e-45, -1e-45] * 100, np.float32)
tmax = torch.zeros(2); agg = torch.zeros(2, dtype=torch.int64)
be.check(be.c.gmx_tile_stats(be.ptr(torch.from_numpy(bad)), bad.size, 40, be.ptr(tmax), be.ptr(agg), be.stream()), "tile_stats")

# resampling with hostile weights (NaN / inf / nothing / one-hot), ragged sizes
rng = np.random.default_rng(0)
for n in (1, 5, 1023, 1025, 4097):
    for kind in (O.SYSTEMATIC, O.STRATIFIED, O.MULTINOMIAL):
        lw = rng.normal(size=n).astype(np.float32) * 30
        lw[rng.integers(0, n, size=max(1, n // 50))] = np.float32("nan")
        lw[rng.integers(0, n)] = -np.float32("inf")
        cdf, total, M, shift = O.weight_cdf(lw)
        O.ancestors(kind, O.key(n), cdf)
    O.weight_cdf(np.full(n, -np.inf, np.float32))

# the OpenMP sweep (bench.py's cpu_baseline) under the sanitizers
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liborc_sweep_san.so"))
n, T = 5000, 5
from genjax_amd import workloads  # noqa: E402
ys = workloads.lgssm_data(T)
f32, u64, i32 = np.float32, np.uint64, np.int32
x, x2, lw = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, f32)
cdf, anc = np.zeros(n, u64), np.zeros(n, i32)
maxs, totals = np.zeros(T, f32), np.zeros(T, u64)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rc = lib.orc_lgssm_sweep(ctypes.c_int64(n), ctypes.c_int64(T), P(ys), ctypes.c_uint32(0), ctypes.c_uint32(314159),
                         ctypes.c_float(0.9), ctypes.c_float(0.5), ctypes.c_float(1.0), ctypes.c_float(1.0),
                         ctypes.c_int(O.cdf_shift(n)), P(x), P(x2), P(lw), P(cdf), P(anc), P(maxs), P(totals))
assert rc == 0
print("sanitize_driver ok")
