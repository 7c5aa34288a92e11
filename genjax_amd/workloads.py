"""Synthetic state-space workloads of BASELINE.json's configs, written as
ordinary `@gen` models, plus their closed-form references.

`make_lgssm(G)` / `make_nlssm(G)` take the namespace that provides `gen`,
`normal`, ... so the SAME model source runs on this package (G = genjax_amd)
and on the CPU oracle (G = oracle.genjax_oracle) in the parity tests.
"""
from __future__ import annotations

import math

import numpy as np

# config 2: x_0 ~ N(0,1); x_t ~ N(0.9 x_{t-1}, 0.5); y_t ~ N(x_t, 1.0)   (scale params)
LGSSM = dict(a=0.9, sx=0.5, sy=1.0, s0=1.0)


def make_lgssm(G, p=LGSSM):
    a, sx, sy, s0 = p["a"], p["sx"], p["sy"], p["s0"]

    @G.gen
    def init():
        x = G.normal(0.0, s0) @ "x"
        _ = G.normal(x, sy) @ "y"
        return x

    @G.gen
    def step(x_prev):
        x = G.normal(a * x_prev, sx) @ "x"
        _ = G.normal(x, sy) @ "y"
        return x

    return init, step


def lgssm_data(T: int, seed: int = 2024, p=LGSSM) -> np.ndarray:
    """Observations y_1..y_T simulated once on the host (numpy Generator; the
    data are inputs, not part of the parity surface)."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, p["s0"])
    ys = []
    for t in range(T):
        if t > 0:
            x = p["a"] * x + p["sx"] * rng.normal()
        ys.append(x + p["sy"] * rng.normal())
    return np.asarray(ys, dtype=np.float32)


def kalman_log_ml(ys, p=LGSSM) -> float:
    """Exact log p(y_1..T) of the linear-Gaussian model (float64 Kalman filter)."""
    a, q, r = p["a"], p["sx"] ** 2, p["sy"] ** 2
    m, P = 0.0, p["s0"] ** 2
    ll = 0.0
    for t, y in enumerate(np.asarray(ys, dtype=np.float64)):
        if t > 0:
            m, P = a * m, a * a * P + q
        S = P + r
        ll += -0.5 * (math.log(2 * math.pi * S) + (y - m) ** 2 / S)
        K = P / S
        m, P = m + K * (y - m), (1 - K) * P
    return ll


# config 3: x_t ~ N(0.5 x + 25 x/(1+x^2) + 8 cos(1.2 t), sqrt(10)); y_t ~ N(x_t^2/20, 1)
def make_nlssm(G, cos=None):
    if cos is None:
        cos = G.numpy.cos if hasattr(G, "numpy") else G.cos
    s10 = math.sqrt(10.0)

    @G.gen
    def init():
        x = G.normal(0.0, s10) @ "x"
        _ = G.normal(x * x / 20.0, 1.0) @ "y"
        return x

    @G.gen
    def step(x_prev, t):
        mean = 0.5 * x_prev + 25.0 * x_prev / (1.0 + x_prev * x_prev) + 8.0 * cos(1.2 * t)
        x = G.normal(mean, s10) @ "x"
        _ = G.normal(x * x / 20.0, 1.0) @ "y"
        return x

    return init, step


def nlssm_data(T: int, seed: int = 2025) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, math.sqrt(10.0))
    ys = []
    for t in range(T):
        if t > 0:
            x = 0.5 * x + 25.0 * x / (1.0 + x * x) + 8.0 * math.cos(1.2 * t) + math.sqrt(10.0) * rng.normal()
        ys.append(x * x / 20.0 + rng.normal())
    return np.asarray(ys, dtype=np.float32)


# ---------------------------------------------------------------------------
# BASELINE config 5: mixture-model cluster assignments
# ---------------------------------------------------------------------------
def make_mixture(g, jnp=None, obs_scale=1.0):
    """`generate_datapoint` of 7_application_dirichlet_mixture_model.ipynb (cell 6): one datapoint's
    cluster index and observation given the mixture weights and the cluster means."""
    if jnp is None:
        from . import numpy as jnp

    @g.gen
    def generate_datapoint(probs, clusters):
        idx = g.categorical(jnp.log(probs)) @ "idx"
        obs = g.normal(clusters[idx], obs_scale) @ "obs"
        return obs
    return generate_datapoint


def mixture_data(n: int, K: int = 64, seed: int = 7):
    """n points from K unit-variance Gaussians with means on a grid (spacing 4), plus a perturbed
    guess of the means and non-uniform weights for the assignment step.  numpy Philox: the data are
    an input, not part of the parity claim."""
    rng = np.random.default_rng(seed)
    means = (4.0 * (np.arange(K) - (K - 1) / 2.0)).astype(np.float32)
    z = rng.integers(0, K, size=n)
    x = (means[z] + rng.standard_normal(n)).astype(np.float32)
    guess = (means + 0.3 * rng.standard_normal(K)).astype(np.float32)
    probs = rng.dirichlet(np.full(K, 5.0)).astype(np.float32)
    return x, guess, probs, z.astype(np.int32)
