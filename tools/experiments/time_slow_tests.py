"""times the pieces of the two slowest GPU tests (where do 260 s and 168 s go?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genjax_amd import _lib
_lib.install(None)
from tests import cookbook
def t(name, fn):
    t0 = time.time(); fn(); torch.cuda.synchronize(); print(f"{name}: {time.time() - t0:.1f} s", flush=True)
for k, n in ((12, 40), (20, 100), (40, 500), (64, 1000)):
    t(f"mixture k={k} n={n} B=5", lambda: cookbook.check_mixture_notebook_under_a_batch(k=k, n=n))
t("mixture k=20 n=100 B=3000", lambda: cookbook.check_mixture_notebook_under_a_batch(k=20, n=100, B=3000, seed=4))
for npts, J in ((100, 40), (500, 200), (5000, 1000)):
    t(f"hmc npts={npts} J={J}", lambda: cookbook.check_hmc_through_long_vector_sites(npts=npts, J=J))
t("hmc J=200 K=300000 L=2", lambda: cookbook.check_hmc_through_long_vector_sites(npts=500, J=200, K=300_000, L=2))
