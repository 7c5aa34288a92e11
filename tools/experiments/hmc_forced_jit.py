"""tests/cookbook.py::check_hmc_through_long_vector_sites with EVERY program sent through hiprtc (engine.JIT_MIN_PARTICLES =
1): the HMC / Regenerate / Rejuvenate programs over long vector sites — loops, gradient contributions, sums, the gather's
scatter-add — as specialised kernels against the oracle, at a size the oracle finishes in seconds.  One-off (minutes of
hiprtc); the default GPU suite runs these programs on the interpreter and one selection at 3e5 particles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import _lib, engine
_lib.install(None)
engine.JIT_MIN_PARTICLES = 1
from tests import cookbook
t0 = time.time()
cookbook.check_hmc_through_long_vector_sites(npts=100, J=40)
print(f"J = 40: every check passed with specialised kernels, {time.time() - t0:.0f} s", flush=True)
print("jit rejected by the first-launch cross-check:", int(_lib.get().c.gmx_jit_rejected_count()))
