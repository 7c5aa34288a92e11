"""Where does BASELINE config 4's time go on the HOST?  (ImportanceK k = 1e7 + systematic resample + gather of theta: the
kernels add up to 0.665 ms, a run takes 0.78.)  Times the enqueue loop with and without the final synchronize and prints
cProfile's top entries of the enqueue loop."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, _lib, numpy as jnp
from genjax_amd.inference import smc
_lib.install(None)
sig = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
ysch = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)

@G.gen
def schools():
    mu = G.normal(0.0, 5.0) @ "mu"
    log_tau = G.normal(0.0, 1.0) @ "log_tau"
    theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
    _ = G.normal(theta, jnp.array(sig)) @ "y"
    return theta
tgt = G.Target(schools, (), C["y"].set(ysch))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_384
alg = smc.ImportanceK(tgt, k_particles=k)
box = {}

def run1():
    c = alg.run_smc(G.key(2))
    r = smc.resample(G.split(G.key(2))[0], c, "systematic")
    box["theta1"] = r.get_particles().get_choices()["theta"]
for _ in range(3):
    run1()
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    run1()
t_host = (time.perf_counter() - t0) / reps
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / reps
print(f"k = {k}: enqueue loop {1e3 * t_host:.3f} ms per run on the host, {1e3 * t_all:.3f} ms per run with the final synchronize")
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    run1()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
