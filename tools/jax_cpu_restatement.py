#!/usr/bin/env python3
"""A self-contained jax.numpy restatement of BASELINE config 2 — the scalable form of what the reference's
`ImportanceK` / `ParticleCollection` do (`/root/reference` src/genjax/_src/inference/smc.py:298-315, 96-109):
vmapped importance over split keys (per-site key = fold_in(key, counter), static.py:260-263), logsumexp for the
evidence, inverse-CDF systematic resampling, written by this build (it is NOT reference code and
imports nothing from the reference) so that `bench.py`'s cpu_baseline leg can time "the jax[cpu] path" whenever a
box happens to have jax installed.  The build container and the GPU boxes of this round do
not have jax: there `available()` is False and bench.py says so.

    python tools/jax_cpu_restatement.py [n] [T]      -> one JSON line
"""
from __future__ import annotations

import json
import os
import sys
import time


def available() -> bool:
    try:
        import jax  # noqa: F401
        return True
    except Exception:
        return False


def time_sweep(n: int, T: int, ys, seed: int = 314159, budget_s: float = 20.0, a=0.9, sx=0.5, sy=1.0, s0=1.0):
    """particle-steps/s of a bootstrap SMC sweep under jax on the host CPU (JAX_PLATFORMS=cpu), bounded sample."""
    os.environ.setdefault("JAX_PLATFORMS", "cpu")
    import jax
    import jax.numpy as jnp
    import numpy as np

    half_log_2pi = 0.5 * float(np.log(2.0 * np.pi))

    def normal_logpdf(x, loc, scale):                       # tfd.Normal.log_prob
        return -0.5 * jnp.square(x / scale - loc / scale) - (half_log_2pi + jnp.log(scale))

    def particle_step(key, x_prev, y, first):
        site = jax.random.fold_in(key, 1)                   # site counter 1: "x"; "y" is constrained
        z = jax.random.normal(site, (), jnp.float32)
        x = jnp.where(first, z * s0, z * sx + a * x_prev)
        return x, normal_logpdf(y, x, sy)

    run_key = jax.random.key(seed)

    @jax.jit
    def smc_step(x, t, y):
        ks = jax.random.split(jax.random.fold_in(run_key, t), 3)
        keys = jax.random.split(ks[0], n)
        x_new, lw = jax.vmap(particle_step, in_axes=(0, 0, None, None))(keys, x, y, t == 0)
        m = jnp.max(lw)
        w = jnp.exp(lw - m)
        c = jnp.cumsum(w)
        total = c[-1]
        u0 = jax.random.uniform(ks[1], (), jnp.float32)
        pos = (jnp.arange(n, dtype=jnp.float32) + u0) * (total / n)
        anc = jnp.clip(jnp.searchsorted(c, pos, side="right"), 0, n - 1)
        inc = m + jnp.log(total) - jnp.log(jnp.float32(n))
        return x_new[anc], inc

    x = jnp.zeros((n,), jnp.float32)
    ys = jnp.asarray(ys, jnp.float32)
    x, inc = smc_step(x, jnp.int32(0), ys[0])               # compile + first step
    inc.block_until_ready()
    t0 = time.perf_counter()
    x, inc = smc_step(x, jnp.int32(1), ys[1 % len(ys)])
    inc.block_until_ready()
    one = max(time.perf_counter() - t0, 1e-6)
    Tn = int(max(2, min(T, budget_s / one)))
    log_ml = 0.0
    x = jnp.zeros((n,), jnp.float32)
    t0 = time.perf_counter()
    for t in range(Tn):
        x, inc = smc_step(x, jnp.int32(t), ys[t % len(ys)])
        log_ml += float(inc)
    dt = time.perf_counter() - t0
    return {"value": n * Tn / dt, "unit": "particle-steps/s", "kind": "jax[cpu] restatement (tools/jax_cpu_restatement.py)",
            "jax": jax.__version__, "devices": [str(d) for d in jax.devices()], "cores": os.cpu_count(),
            "sample": f"{Tn} of {T} SMC steps x {n} particles", "log_ml_of_sample": log_ml}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if not available():
        print(json.dumps({"error": "jax is not importable here"}))
        sys.exit(0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from genjax_amd import workloads
    print(json.dumps(time_sweep(n, T, workloads.lgssm_data(T))))
