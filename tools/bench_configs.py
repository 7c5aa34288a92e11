"""Timings of BASELINE configs 3 and 4 on one MI355X through the functional SMC API
(config 2 is bench.py; config 5 is tools/bench_mixture.py).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import genjax_amd as G
from genjax_amd import ChoiceMapBuilder as C, numpy as jnp, workloads
from genjax_amd.inference import smc

out = {}
# ---- config 3: nonlinear SSM, N = 1e6, T = 100, one Gaussian-drift MH sweep per step ----
n, T = int(os.environ.get("N", 1_000_000)), int(os.environ.get("T", 100))
ys = workloads.nlssm_data(T)
init, step = workloads.make_nlssm(G)
req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})


def sweep(key):
    acc = []
    for t in range(T):
        kp, kr, km = G.split(G.fold_in(key, t), 3)
        obs = G.ChoiceMap.kw(y=float(ys[t]))
        if t == 0:
            coll = smc.ImportanceK(G.Target(init, (), obs), k_particles=n).run_smc(kp)
        else:
            coll = smc.resample(kr, coll, "systematic")
            coll = smc.rejuvenate(km, coll, req)
            acc.append(coll.accept)
            coll = smc.extend(kp, coll, step, lambda tr_: (tr_.get_retval(), float(t)), obs)
    return coll, acc


coll, acc = sweep(G.key(7))
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for r in range(reps):
    coll, acc = sweep(G.key(7 + r))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
out["config3"] = {"workload": "nonlinear SSM + 1 MH (Gaussian drift 0.5) sweep per step, functional API (eager launches)",
                  "particles": n, "T": T, "ms_per_sweep": 1e3 * dt, "particle_steps_per_s": n * T / dt,
                  "us_per_step": 1e6 * dt / T,
                  "log_ml": float(coll.get_log_marginal_likelihood_estimate()),
                  "mean_accept_rate": float(torch.stack([a.float().mean() for a in acc]).mean())}

# the same functional sweep captured ONCE into a hipGraph (torch.cuda.graph sees the C-ABI launches too:
# they go to torch's current stream) and replayed: device time without the Python dispatch
try:
    ref, _ = sweep(G.key(7))
    ref_x = ref.get_particles().get_retval().clone()
    for tag, na in (("graph_replay_one_stream", False), ("graph_replay", True)):
        cap = smc.capture(sweep, G.key(7), noise_ahead=na)
        gcoll, gacc = cap.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(ref_x, gcoll.get_particles().get_retval()))
        t0 = time.perf_counter()
        for r in range(10):
            cap.replay()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / 10
        out["config3"][tag] = {"how": "smc.capture(sweep, key): the functional loop captured once, replayed"
                                      + ("; its draws by background programs on a second stream (noise ahead)" if na else ""),
                               "ms_per_sweep": 1e3 * dtg, "particle_steps_per_s": n * T / dtg,
                               "us_per_step": 1e6 * dtg / T, "same_particles_as_eager": same,
                               "log_ml": float(gcoll.get_log_marginal_likelihood_estimate())}
        del cap, gcoll, gacc
except Exception as e:
    out["config3"]["graph_replay"] = {"error": repr(e)[:300]}

# the native form: BootstrapSweep(rejuvenate=...) = k_vm -> resample -> fused MH per step, pre-bound
# persistent buffers, one hipGraph for the sweep
try:
    sw = smc.BootstrapSweep(init, step, n, T, step_extra=lambda t: (float(t),), rejuvenate=req).prepare(
        G.key(7), torch.from_numpy(ys))
    sw.capture()
    sw.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(20):
        sw.launch()
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / 20
    out["config3"]["native_sweep"] = {"ms_per_sweep": 1e3 * dts, "particle_steps_per_s": n * T / dts,
                                      "us_per_step": 1e6 * dts / T, "log_ml": sw.log_ml(),
                                      "accept_rate_last_step": float(sw.accept.float().mean())}
except Exception as e:
    out["config3"]["native_sweep"] = {"error": repr(e)[:300]}

# ---- config 4: 8-schools, ImportanceK k = 1e7 + one global systematic resample ----
sig, ysch = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0], np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)


@G.gen
def schools():
    mu = G.normal(0.0, 5.0) @ "mu"
    log_tau = G.normal(0.0, 1.0) @ "log_tau"
    theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
    _ = G.normal(theta, jnp.array(sig)) @ "y"
    return theta


k = int(os.environ.get("K4", 10_000_000))
alg = smc.ImportanceK(G.Target(schools, (), C["y"].set(ysch)), k_particles=k)


def run4(seed):
    c = alg.run_smc(G.key(seed))
    r = smc.resample(G.key(seed + 1), c, "systematic")
    th = r.get_particles().get_choices()["theta"]          # materialises the gathered latents
    return c, r, th


run4(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r_ in range(reps):
    c, r, th = run4(2 + r_)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
t1 = time.perf_counter()
for r_ in range(reps):
    c = alg.run_smc(G.key(20 + r_))
torch.cuda.synchronize()
dti = (time.perf_counter() - t1) / reps
out["config4"] = {"workload": "8-schools ImportanceK + one global systematic resample + gather of theta",
                  "k_particles": k, "ms_total": 1e3 * dt, "ms_importance": 1e3 * dti, "particles_per_s": k / dt,
                  "importance_GBps_algorithmic(48B/particle)": 48.0 * k / dti / 1e9,
                  "log_ml": float(c.get_log_marginal_likelihood_estimate()),
                  "posterior_mean_mu": float(r.get_particles().get_choices()["mu"].float().mean())}
try:
    out["config3"]["captured_functional_over_native"] = (out["config3"]["graph_replay"]["us_per_step"]
                                                         / out["config3"]["native_sweep"]["us_per_step"])
except Exception:
    pass
print(json.dumps(out))
