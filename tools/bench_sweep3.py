"""BASELINE config 3 as the native captured sweep (BootstrapSweep(rejuvenate=...)): us / step and the
size of each site program."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference import smc

n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
ys = workloads.nlssm_data(T)
init, step = workloads.make_nlssm(G)
req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
out = {}
states = []
for name, na in (("one_stream", False), ("noise_ahead", True)):
    sw = smc.BootstrapSweep(init, step, n, T, step_extra=lambda t: (float(t),), rejuvenate=req, noise_ahead=na).prepare(
        G.key(7), torch.from_numpy(ys)).capture()
    for _ in range(3):
        sw.launch()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        sw.launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    states.append([v.clone() for v in sw.state()] + [sw.accept.clone(), sw.totals.clone()])
    out[name] = {"us_per_step": 1e6 * dt / T, "particle_steps_per_s": n * T / dt, "log_ml": sw.log_ml(),
                 "programs": {k: {"instr": int(p.comp.blob[2]), "regs": int(p.comp.blob[3])}
                              for k, p in (("init", sw.p_init), ("mhvm_step", sw.p_mhvm_step)) if p is not None}}
out["bit_identical"] = all(torch.equal(a, b) for a, b in zip(*states))
out["us_per_step"] = out["noise_ahead"]["us_per_step"]
out["particle_steps_per_s"] = out["noise_ahead"]["particle_steps_per_s"]
print(json.dumps(out))
