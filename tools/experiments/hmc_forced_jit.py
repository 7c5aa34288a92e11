"""tests/cookbook.py::check_hmc_through_long_vector_sites with EVERY program sent through hiprtc (engine.JIT_MIN_PARTICLES =
1): the HMC / Regenerate / Rejuvenate programs over long vector sites — loops, gradient contributions, sums, the gather's
scatter-add — as specialised kernels against the oracle, at a size the oracle finishes in seconds.  Counts the programs
that were specialised and launched as such."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import _lib, engine
_lib.install(None)
engine.JIT_MIN_PARTICLES = 1
from tests import cookbook
made, launched = [0], [0]
_spec, _launch = engine.Compiled.specialize, engine.Compiled.launch

def spec(self, *a, **k):
    was = self.is_specialized() or self.is_partly_specialized()
    ok = _spec(self, *a, **k)
    if ok and not was:
        made[0] += 1
    return ok

def launch(self, bound):
    if self.is_specialized() or self.is_partly_specialized():
        launched[0] += 1
    return _launch(self, bound)
engine.Compiled.specialize, engine.Compiled.launch = spec, launch
t0 = time.time()
cookbook.check_hmc_through_long_vector_sites(npts=100, J=40)
print(f"J = 40: every check passed, {time.time() - t0:.0f} s; programs specialised: {made[0]}, launches of specialised programs: {launched[0]}", flush=True)
print("jit rejected by the first-launch cross-check:", int(_lib.get().c.gmx_jit_rejected_count()))
