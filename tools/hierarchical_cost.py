#!/usr/bin/env python3
"""8-schools at J schools under ImportanceK with K particles: the model computes with the values of a long LATENT vector
site (`theta ~ normal(mu 1_J, tau 1_J); y ~ normal(theta, sigma_J)`): each of the two sites is one counted loop per
particle, the second reading the first's values back from the launch's own output (engine.StepAlias).
    python tools/hierarchical_cost.py [K]   ->   seconds per ImportanceK.run_smc at J = 8 (unrolled), 500 and 5 000"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import genjax_amd as G  # noqa: E402
from genjax_amd import ChoiceMapBuilder as C, numpy as jnp  # noqa: E402
from genjax_amd.inference.smc import ImportanceK  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
out = {"K": K, "what": "seconds per ImportanceK.run_smc of the 8-schools model at J schools; bytes = the 2 J + 4 stored "
                       "f32 leaves per particle"}
for J in (8, 500, 5000):
    sig = np.linspace(9, 18, J).astype(np.float32)
    ys = np.linspace(-3, 28, J).astype(np.float32)

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        log_tau = G.normal(0.0, 1.0) @ "log_tau"
        theta = G.normal(mu * jnp.ones(J), jnp.exp(log_tau) * jnp.ones(J)) @ "theta"
        G.normal(theta, jnp.array(sig)) @ "y"
        return mu
    alg = ImportanceK(G.Target(schools, (), C["y"].set(ys)), k_particles=K)
    t0 = time.perf_counter()
    c = alg.run_smc(G.key(1))
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        c = alg.run_smc(G.key(1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[str(J)] = {"seconds": dt, "first_call_seconds": first, "draws_per_s": K * (J + 2) / dt,
                   "stored_GB_per_s": 4.0 * K * (J + 4) / dt / 1e9,
                   "log_ml": float(c.get_log_marginal_likelihood_estimate())}
print(json.dumps(out))
