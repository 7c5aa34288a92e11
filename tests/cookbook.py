"""The reference's COOKBOOK as a parity corpus: the models and inference code of four notebooks, typed in
once against this package (`G`) and once against the CPU oracle (`O`), at the notebooks' own sizes, compared bit for bit.

  docs/cookbook/inactive/update/3_speed_gains.ipynb                         c4 (model), c6, c8 (SIR), c13, c15 (MH move), c10 / c17 (sizes)
  docs/cookbook/inactive/inference/mcmc.ipynb                               c4 (model), c8 (MH move), c10 (a proposal that takes a TRACE), c12-c16
  docs/cookbook/inactive/inference/importance_sampling.ipynb                c16 (SIR from model.importance + categorical)
  docs/cookbook/inactive/expressivity/... 7_application_dirichlet_mixture_model.ipynb   c6 (model), c10 (Gibbs updates), under a BATCH of keys

What the notebooks need from the build beyond round 5 (all reference behaviour, file:line in the code they exercise):
  * a plate's / a long scan's RETURN values are plain stacked arrays the model computes with (`jnp.sum(a)`, `clusters[idx]`):
    vmap.py:180-191, scan.py:221-233;
  * `Vmap.edit` of a bare distribution's plate (`normal.vmap()`) under `model.update`: vmap.py:237-275;
  * a Trace is a pytree and can be an argument of a `@gen` function: generative_function.py:72-230, static.py:80-119;
  * `categorical(logits, sample_shape=n)` for any n: tensorflow_probability/__init__.py:52-55;
  * any pytree as a scan's carry: scan.py:200-294.

This module is test infrastructure: `tests/test_host_logic.py` runs it on the CPU mirror, `tests/test_gpu_parity.py` through the
C-ABI on the GPU."""
from __future__ import annotations

import numpy as np
import torch

from oracle import genjax_oracle as O

f32 = np.float32


def npv(v):
    return v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)


def same(a, b):
    a, b = npv(a), np.asarray(b)
    if a.dtype.kind == "f":
        b = b.astype(np.float32)
    return bool(np.array_equal(a, np.broadcast_to(b, a.shape) if b.shape != a.shape and b.size in (1, a.size) and b.ndim <= a.ndim else b))


def inorder_sum(x):
    """`jnp.sum` as the build defines it inside a site program: element order (oracle/genjax_oracle.py::sum_vector below 4096
    elements; under a batch of keys at any length)"""
    x = np.asarray(x, np.float32)
    acc = np.zeros(x.shape[:-1], np.float32)
    for j in range(x.shape[-1]):
        acc = (acc + x[..., j]).astype(np.float32)
    return acc


def _keys(seed, N):
    import genjax_amd as G
    return (G.key(seed), O.key(seed)) if N is None else (G.split(G.key(seed), N), O.split(O.key(seed), N))


# =====================================================================================================================
# 3_speed_gains.ipynb
# =====================================================================================================================
def speed_gains_models(n):
    import genjax_amd as G
    from genjax_amd import numpy as jnp

    @G.gen
    def model(size_model):                                    # c4, as written
        size_model = size_model.unwrap()
        x = G.normal(0.0, 1.0) @ "x"
        a = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "a"
        b = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "b"
        c = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "c"
        obs = G.normal(jnp.sum(a) + jnp.sum(b) + jnp.sum(c) + x, 5.0) @ "obs"
        return obs

    @G.gen
    def default_proposal(size_model):                         # c6
        size_model = size_model.unwrap()
        _ = G.normal(0.0, 1.0) @ "x"
        _ = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "a"
        _ = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "b"
        _ = G.normal.vmap()(jnp.zeros(size_model), jnp.ones(size_model)) @ "c"
        return None

    @G.gen
    def rejuv_x(x):                                           # c13
        x = G.normal(x, 1.0) @ "x"
        return x

    z, o = np.zeros(n, f32), np.ones(n, f32)
    big = n >= 4096          # ONE trace: the build sums a large plate's values in its fixed tree (numpy.sum / sum_vector)

    def osum(x, one):
        return O.sum_vector(x) if (one and big) else inorder_sum(x)

    def o_model(one):
        @O.gen
        def m():
            x = O.normal(f32(0.0), f32(1.0)) @ "x"
            a = O.Vmap(O.normal)(z, o) @ "a"
            b = O.Vmap(O.normal)(z, o) @ "b"
            c = O.Vmap(O.normal)(z, o) @ "c"
            tot = ((osum(a, one) + osum(b, one)).astype(f32) + osum(c, one)).astype(f32) + x
            return O.normal(tot.astype(f32), f32(5.0)) @ "obs"
        return m

    @O.gen
    def o_proposal():
        _ = O.normal(f32(0.0), f32(1.0)) @ "x"
        _ = O.Vmap(O.normal)(z, o) @ "a"
        _ = O.Vmap(O.normal)(z, o) @ "b"
        _ = O.Vmap(O.normal)(z, o) @ "c"
        return None

    @O.gen
    def o_rejuv(x):
        return O.normal(x, f32(1.0)) @ "x"
    return (model, default_proposal, rejuv_x), (o_model, o_proposal, o_rejuv)


def check_speed_gains_sir(n=100, N=100, seed=0):
    """c8: `sir(key, N, use_fast, size_model)` both ways — `vmap(model.importance)` over N keys (fast), and
    `default_proposal.simulate` + two `assess` (slow) — then `categorical.simulate(key, (weights,))` and the gather of the
    chosen particle's choices.  Choices, weights and the drawn index bit-exact."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C
    from genjax_amd.core.pytree import Const
    (model, proposal, _), (o_model, o_proposal, _) = speed_gains_models(n)
    om = o_model(False)
    size = (Const(n),)
    key, okey = G.key(seed), O.key(seed)
    obs, oobs = C["obs"].set(1.0), O.C.d({"obs": f32(1.0)})
    # fast
    traces, weights = G.vmap(model.importance, in_axes=(0, None, None))(G.split(key, N), obs, size)
    otr, ow = om.importance(O.split(okey, N), oobs, ())
    for a in ("x", "a", "b", "c"):
        assert same(traces.get_choices()[a], otr.get_choices()[a]), a
    assert same(weights, ow)
    idx = G.categorical.simulate(key, (weights,)).get_retval()
    oidx = O.categorical._sample(okey, (np.asarray(ow, f32),))
    assert int(idx) == int(oidx)
    resampled = traces.get_choices()["a"][int(idx)]
    assert same(resampled, np.asarray(otr.get_choices()["a"])[int(oidx)])
    # slow: simulate from the proposal, score under proposal and model (the notebook maps `assess` over the particles;
    # here the same calls are batch polymorphic)
    ptr = G.vmap(proposal.simulate, in_axes=(0, None))(G.split(key, N), size)
    optr = o_proposal.simulate(O.split(okey, N), ())
    chm = ptr.get_choices()
    q, _ = proposal.assess(chm, size)
    oq, _ = o_proposal.assess(optr.get_choices(), (), (N,))
    # (`C["obs"].set(jnp.ones(N) * obs["obs"])` in the notebook, sliced per particle by its vmap over idx: ONE observation
    #  per particle — here a device tensor with the particle axis; a host array would be a launch-uniform VECTOR)
    chm_model = chm | C["obs"].set(torch.ones(N, device=weights.device))
    p, _ = model.assess(chm_model, size)
    ochm = optr.get_choices().merge(O.C.d({"obs": np.ones(N, f32)})) if hasattr(optr.get_choices(), "merge") else None
    op, _ = om.assess(ochm, (), (N,))
    assert same(q, oq) and same(p, op)
    return dict(idx=int(idx))


def check_speed_gains_mh(n=1000, N=None, seed=3):
    """c15 `metropolis_hastings_move(key, trace, use_fast)`: propose `x` from `rejuv_x`, then the ratio through `model.update`
    (fast: argdiffs no_change, only `x` and `obs` are visited — the three plates are carried over untouched) and through two
    `model.assess` (slow), the backward `rejuv_x.assess` of the discard, the accept draw.  ONE trace (the notebook's c17:
    n = 1000 ... 1e8) or a batch of N chains."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff
    from genjax_amd.core.pytree import Const
    (model, _, rejuv_x), (o_model, _, o_rejuv) = speed_gains_models(n)
    one = N is None
    om = o_model(one)
    size = (Const(n),)
    k, ok_ = _keys(seed, N)
    trace, _ = model.importance(k, C["obs"].set(1.0), size)
    otrace, _ = om.importance(ok_, O.C.d({"obs": f32(1.0)}), ())
    for a in ("x", "a", "b", "c"):
        assert same(trace.get_choices()[a], otrace.get_choices()[a]), a
    key, okey = _keys(seed + 1, N)
    # -- the move, as written
    model_args = trace.get_args()
    key, subkey = G.split(key)
    okey, osub = (O.split(okey)[..., 0, :], O.split(okey)[..., 1, :])
    fwd_choice, fwd_weight, _ = rejuv_x.propose(subkey, (trace.get_choices()["x"],))
    ofc, ofw, _ = o_rejuv.propose(osub, (np.asarray(otrace.get_choices()["x"], f32),))
    assert same(fwd_choice["x"], ofc["x"]) and same(fwd_weight, ofw)
    key, subkey = G.split(key)
    okey, osub = (O.split(okey)[..., 0, :], O.split(okey)[..., 1, :])
    # fast
    argdiffs = Diff.no_change(model_args)
    new_tr, weight, _, discard = model.update(subkey, trace, fwd_choice, argdiffs)
    onew, oweight, odisc = om.update(osub, otrace, ofc, ())
    assert same(weight, oweight), (npv(weight), np.asarray(oweight))
    assert same(discard["x"], odisc["x"])
    assert same(new_tr.get_score(), onew.get_score())
    bwd, _ = rejuv_x.assess(discard, (fwd_choice["x"],))
    obwd, _ = o_rejuv.assess(odisc, (np.asarray(ofc["x"], f32),), () if one else (N,))
    assert same(bwd, obwd)
    alpha_fast = weight - fwd_weight + bwd
    # slow
    chm = trace.get_choices()
    w_old, _ = model.assess(chm, model_args)
    w_new, _ = model.assess(fwd_choice | chm, model_args)
    ow_old, _ = om.assess(otrace.get_choices(), (), () if one else (N,))
    ow_new, _ = om.assess(ofc.merge(otrace.get_choices()), (), () if one else (N,))
    assert same(w_old, ow_old) and same(w_new, ow_new)
    old_x = C["x"].set(chm["x"])
    bwd2, _ = rejuv_x.assess(old_x, (fwd_choice["x"],))
    assert same(bwd2, obwd)
    alpha_slow = w_new - w_old - fwd_weight + bwd2
    # the accept draw: jnp.log(uniform(subkey)) < alpha
    key, subkey = G.split(key)
    okey, osub = (O.split(okey)[..., 0, :], O.split(okey)[..., 1, :])
    u = G.random.uniform(subkey)
    assert same(u, O.random_uniform(osub, ()))
    # jax.lax.cond(jnp.log(u) < alpha, lambda: fwd_choice, lambda: old_choice)
    from genjax_amd import numpy as jnp
    ret = jnp.lax.cond(jnp.log(u) < alpha_fast, lambda: fwd_choice, lambda: old_x)
    oacc = O.log(np.asarray(O.random_uniform(osub, ()), f32)) < ((np.asarray(oweight, f32) - np.asarray(ofw, f32)).astype(f32) + np.asarray(obwd, f32)).astype(f32)
    assert same(ret["x"], np.where(oacc, np.asarray(ofc["x"], f32), np.asarray(otrace.get_choices()["x"], f32)))
    return dict(alpha_fast=npv(alpha_fast), alpha_slow=npv(alpha_slow))


# =====================================================================================================================
# mcmc.ipynb
# =====================================================================================================================
def check_mcmc_notebook(N=None, steps=6, seed=0):
    """c4-c16: linear model `y ~ normal(a x + b, 1)`, a custom proposal `prop(tr, *_)` that reads `tr.get_choices()["a"]` from
    the TRACE it is handed, `metropolis_hastings_move` (propose -> update -> assess the discard on the NEW trace -> accept),
    chained `steps` times as `jax.lax.scan` does; every step's weights, the accept decisions and the final choices bit-exact.
    N: None = the notebook's one chain, else a batch of chains."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff
    from genjax_amd import numpy as jnp

    @G.gen
    def model(x):
        a = G.normal(0.0, 5.0) @ "a"
        b = G.normal(0.0, 1.0) @ "b"
        y = G.normal(a * x + b, 1.0) @ "y"
        return y

    @G.gen
    def prop(tr, *_):
        orig_a = tr.get_choices()["a"]
        a = G.normal(orig_a, 1.0) @ "a"
        return a

    @O.gen
    def o_model(x):
        a = O.normal(f32(0.0), f32(5.0)) @ "a"
        b = O.normal(f32(0.0), f32(1.0)) @ "b"
        return O.normal((a * x).astype(f32) + b, f32(1.0)) @ "y"

    @O.gen
    def o_prop(tr, *_):
        orig_a = np.asarray(tr.get_choices()["a"], f32)
        return O.normal(orig_a, f32(1.0)) @ "a"
    bshape = () if N is None else (N,)
    k, ok_ = _keys(seed, N)
    trace, _ = model.importance(k, C["y"].set(4.0), (5.0,))
    otrace, _ = o_model.importance(ok_, O.C.d({"y": f32(4.0)}), (f32(5.0),))
    assert same(trace.get_choices()["a"], otrace.get_choices()["a"])
    key, okey = _keys(seed + 1, N)
    accepted = 0
    for step in range(steps):
        # mh_keys = split(key, num_updates): one key per move; inside the move: key, subkey = split(key)
        kk = G.fold_in(key, step) if N is None else G.split(G.fold_in(G.key(seed + 1), step), N)
        okk = O.fold_in(okey, step) if N is None else O.split(O.fold_in(O.key(seed + 1), step), N)
        kk, subkey = G.split(kk)
        okk, osub = (O.split(okk)[..., 0, :], O.split(okk)[..., 1, :])
        model_args = trace.get_args()
        argdiffs = Diff.no_change(model_args)
        fwd_choices, fwd_weight, _ = prop.propose(kk, (trace,))
        ofc, ofw, _ = o_prop.propose(okk, (otrace,))
        assert same(fwd_choices["a"], ofc["a"]) and same(fwd_weight, ofw), step
        new_trace, weight, _, discard = model.update(subkey, trace, fwd_choices, argdiffs)
        onew, ow, odisc = o_model.update(osub, otrace, ofc, (f32(5.0),))
        assert same(weight, ow), (step, npv(weight), np.asarray(ow))
        bwd_weight, _ = prop.assess(discard, (new_trace,))
        obwd, _ = o_prop.assess(odisc, (onew,), bshape)
        assert same(bwd_weight, obwd), step
        alpha = weight - fwd_weight + bwd_weight
        oalpha = ((np.asarray(ow, f32) - np.asarray(ofw, f32)).astype(f32) + np.asarray(obwd, f32)).astype(f32)
        assert same(alpha, oalpha)
        kk, subkey = G.split(kk)
        okk, osub = (O.split(okk)[..., 0, :], O.split(okk)[..., 1, :])
        u = G.random.uniform(subkey)
        ou = O.random_uniform(osub, ())
        assert same(u, ou)
        acc = jnp.log(u) < alpha
        oacc = O.log(np.asarray(ou, f32)) < oalpha
        assert same(acc, oacc)
        # jax.lax.cond(accept, lambda: new_trace, lambda: trace): a select over the whole trace
        trace = jnp.lax.cond(acc, lambda: new_trace, lambda: trace)
        otrace = O.trace_where(oacc, onew, otrace)
        accepted += int(np.sum(npv(acc)))
        for a in ("a", "b"):
            assert same(trace.get_choices()[a], otrace.get_choices()[a]), (step, a)
        assert same(trace.get_score(), otrace.get_score())
    return dict(accepted=accepted)


# =====================================================================================================================
# importance_sampling.ipynb c16
# =====================================================================================================================
def check_importance_sampling_sir(N=1000, K=50, seed=2):
    """c16 `sir(N, K, model, chm)`: `vmap(model.importance)` over N keys, normalise the weights, `categorical.vmap()` K draws
    (`split(key, K)`), gather the samples — here with the notebook's model family (a latent, a noisy observation)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C

    @G.gen
    def model():
        x = G.normal(0.0, 1.0) @ "x"
        y = G.normal(x, 0.5) @ "y"
        return y

    @O.gen
    def o_model():
        x = O.normal(f32(0.0), f32(1.0)) @ "x"
        return O.normal(x, f32(0.5)) @ "y"
    key, okey = G.key(seed), O.key(seed)
    traces, weights = G.vmap(model.importance, in_axes=(0, None, None))(G.split(key, N), C["y"].set(1.3), ())
    otr, ow = o_model.importance(O.split(okey, N), O.C.d({"y": f32(1.3)}), ())
    assert same(weights, ow)
    key2, okey2 = G.key(seed + 1), O.key(seed + 1)
    lw = weights.reshape(1, N).expand(K, N) if hasattr(weights, "expand") else weights
    idxs = G.vmap(lambda k_, l_: G.categorical.simulate(k_, (l_,)).get_retval())(G.split(key2, K), lw.contiguous())
    oidx = O.categorical._sample(O.split(okey2, K), (np.broadcast_to(np.asarray(ow, f32), (K, N)),))
    assert same(idxs, oidx)
    xs = traces.get_choices()["x"][idxs.long()]
    assert same(xs, np.asarray(otr.get_choices()["x"])[oidx])
    return dict(mean=float(npv(xs).mean()))


# =====================================================================================================================
# 7_application_dirichlet_mixture_model.ipynb under a BATCH of keys
# =====================================================================================================================
PRIOR_MEAN, PRIOR_VARIANCE, OBS_VARIANCE = 50.0, 10.0, 1.0


def mixture_models(k, n, alpha):
    import genjax_amd as G
    from genjax_amd import numpy as jnp

    @G.gen
    def generate_cluster(mean, var):
        return G.normal(mean, var) @ "mean"

    @G.gen
    def generate_cluster_weight(alphas):
        return G.dirichlet(alphas) @ "probs"

    @G.gen
    def generate_datapoints(probs, clusters, n_datapoints):
        idx = G.categorical(jnp.log(probs), sample_shape=n_datapoints) @ "idx"
        return G.normal(clusters[idx], OBS_VARIANCE) @ "obs"

    @G.gen
    def generate_data(n_clusters, n_datapoints, alpha_):        # c6, as written
        clusters = generate_cluster.repeat(n=n_clusters.unwrap())(PRIOR_MEAN, PRIOR_VARIANCE) @ "clusters"
        probs = generate_cluster_weight.inline(alpha_ / n_clusters.unwrap() * jnp.ones(n_clusters.unwrap()))
        return generate_datapoints(probs, clusters, n_datapoints) @ "datapoints"

    @O.gen
    def o_cluster(mean, var):
        return O.normal(mean, var) @ "mean"

    @O.gen
    def o_datapoints(probs, clusters):
        idx = O.categorical(logits=O.log(np.asarray(probs, f32)), sample_shape=n) @ "idx"
        cl = np.asarray(clusters, f32)
        mu = np.take_along_axis(cl, idx.astype(np.int64), axis=-1) if cl.ndim == 2 else cl[idx]
        return O.normal(mu, f32(OBS_VARIANCE)) @ "obs"

    @O.gen
    def o_data():
        clusters = O.Repeat(o_cluster, k)(f32(PRIOR_MEAN), f32(PRIOR_VARIANCE)) @ "clusters"
        probs = O.dirichlet(((f32(alpha) / f32(k)) * np.ones(k, f32)).astype(f32)) @ "probs"
        return o_datapoints(probs, clusters) @ "datapoints"
    return generate_data, o_data


def check_mixture_notebook_under_a_batch(k=40, n=500, B=5, seed=0):
    """c6 `generate_data` and the three `trace.update` calls c10's Gibbs moves end with, under B keys at once (the notebook
    runs one chain; a particle ensemble over the same model is BASELINE config 5's setting): `repeat` of k clusters whose
    means are READ BACK (`clusters[idx]`: a gather at traced indices from the plate's stored values), the inlined Dirichlet of
    k weights, `categorical(jnp.log(probs), sample_shape=n)` as one counted loop over the n draws (k logits spilled to
    memory beyond 24), `normal(clusters[idx], 1)` as a vector site looping over the n observations."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff
    from genjax_amd import numpy as jnp
    from genjax_amd.core.pytree import Const
    alpha = float(n / (k * 10))
    args = (Const(k), Const(n), alpha)
    gd, od = mixture_models(k, n, alpha)
    keys, okeys = G.split(G.key(seed), B), O.split(O.key(seed), B)
    tr = G.vmap(gd.simulate, in_axes=(0, None))(keys, args)
    otr = od.simulate(okeys, ())
    for a in (("clusters", "mean"), "probs", ("datapoints", "idx"), ("datapoints", "obs")):
        assert same(tr.get_choices()[a], otr.get_choices()[a]), a
    assert same(tr.get_score(), otr.get_score())
    pts = np.linspace(10.0, 90.0, n).astype(f32)
    uni = (np.ones(k, f32) / f32(k)).astype(f32)
    cons = C["datapoints", "obs"].set(jnp.array(pts)) | C["probs"].set(jnp.ones(k) / k)
    tr2, w = G.vmap(gd.importance, in_axes=(0, None, None))(keys, cons, args)
    otr2, ow = od.importance(okeys, O.C.d({("datapoints", "obs"): pts, "probs": uni}), ())
    assert same(tr2.get_choices()["datapoints", "idx"], otr2.get_choices()["datapoints", "idx"])
    assert same(w, ow) and same(tr2.get_score(), otr2.get_score())
    new_means = np.linspace(20.0, 80.0, k).astype(f32)
    tr3, w3, _, _ = tr2.update(G.split(G.key(seed + 2), B), C["clusters", "mean"].set(jnp.array(new_means)), Diff.no_change(args))
    otr3, ow3, _ = od.update(O.split(O.key(seed + 2), B), otr2, O.C.d({("clusters", "mean"): new_means}), ())
    assert same(w3, ow3) and same(tr3.get_score(), otr3.get_score()), (npv(w3), np.asarray(ow3))
    new_probs = np.linspace(1.0, 2.0, k).astype(f32)
    new_probs = (new_probs / new_probs.sum(dtype=f32)).astype(f32)
    tr4, w4, _, _ = tr3.update(G.split(G.key(seed + 3), B), C["probs"].set(jnp.array(new_probs)), Diff.no_change(args))
    otr4, ow4, _ = od.update(O.split(O.key(seed + 3), B), otr3, O.C.d({"probs": new_probs}), ())
    assert same(w4, ow4) and same(tr4.get_score(), otr4.get_score())
    new_idx = np.random.default_rng(seed).integers(0, k, size=(B, n)).astype(np.int32)
    dev = npv(w4) is not None and (w4.device if hasattr(w4, "device") else None)
    t_idx = torch.from_numpy(new_idx)
    if dev is not None:
        t_idx = t_idx.to(dev)
    tr5, w5, _, _ = tr4.update(G.split(G.key(seed + 4), B), C["datapoints", "idx"].set(t_idx), Diff.no_change(args))
    otr5, ow5, _ = od.update(O.split(O.key(seed + 4), B), otr4, O.C.d({("datapoints", "idx"): new_idx}), ())
    assert same(w5, ow5) and same(tr5.get_score(), otr5.get_score())
    return dict(w=npv(w5))


# =====================================================================================================================
# scan carries and stacked outputs (scan.py:200-294)
# =====================================================================================================================
def check_scan_outputs_and_array_carries(T_=40, N=6, seed=1):
    """`step.scan(n=T)(jnp.zeros(2), None)` — an ARRAY as the initial carry — and a model that computes with a long scan's
    stacked outputs (`jnp.sum(xs)`, `xs[3]`), importance and update, against the oracle."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff
    from genjax_amd import numpy as jnp

    @G.gen
    def step(c, _):
        x = G.normal(c[0] * 0.5 + c[1] * 0.25, 1.0) @ "x"
        return jnp.stack([x, c[0]]), x

    @G.gen
    def model():
        last, xs = step.scan(n=T_)(jnp.zeros(2), None) @ "s"
        y = G.normal(jnp.sum(xs) + last[0] + xs[3], 2.0) @ "y"
        return y

    @O.gen
    def o_step(c, _):
        c = np.asarray(c, f32)
        x = O.normal(((c[..., 0] * f32(0.5)).astype(f32) + (c[..., 1] * f32(0.25)).astype(f32)).astype(f32), f32(1.0)) @ "x"
        return np.stack([np.broadcast_to(x, np.broadcast_shapes(np.shape(x), c[..., 0].shape)),
                         np.broadcast_to(c[..., 0], np.broadcast_shapes(np.shape(x), c[..., 0].shape))], axis=-1).astype(f32), x

    @O.gen
    def o_model():
        last, xs = O.Scan(o_step, T_)(np.zeros(2, f32), None) @ "s"
        xs = np.asarray(xs, f32)
        tot = ((inorder_sum(xs) + last[..., 0]).astype(f32) + xs[..., 3]).astype(f32)
        return O.normal(tot, f32(2.0)) @ "y"
    k, ok_ = _keys(seed, N)
    tr, w = model.importance(k, C["y"].set(1.5), ())
    otr, ow = o_model.importance(ok_, O.C.d({"y": f32(1.5)}), ())
    assert same(tr.get_choices()["s", "x"], otr.get_choices()["s", "x"])
    assert same(w, ow)
    k2, ok2 = _keys(seed + 1, N)
    newx = np.linspace(-1, 1, T_).astype(f32)
    tr2, w2, _, _ = tr.update(k2, C["s", "x"].set(jnp.array(newx)), Diff.no_change(()))
    otr2, ow2, _ = o_model.update(ok2, otr, O.C.d({("s", "x"): newx}), ())
    assert same(w2, ow2) and same(tr2.get_score(), otr2.get_score())
    return dict(w=npv(w2))


# =====================================================================================================================
# HMC / Regenerate through LONG vector sites (requests/hmc.py:138-211, distribution.py:258-300)
# =====================================================================================================================
def _col(v):
    return O.Dual(v.v[..., None], v.t[..., None]) if isinstance(v, O.Dual) else np.asarray(v, f32)[..., None]


def check_hmc_through_long_vector_sites(npts=500, J=200, K=7, L=3, seed=1):
    """`HMC` on scalars that feed LONG vector sites, in ONE launch: Bayesian linear regression `y ~ normal(a xs + b, 0.5)` with
    npts observations, `HMC(S["a"] | S["b"])`; 8-schools at J schools, `HMC(S["mu"])` and `HMC(S["mu"] | S["log_tau"])`.  The
    vector site's counted loop accumulates d score / d w beside the score (static._vector_site_loop, autodiff.grad's custom
    derivative of a loop-carried sum); the oracle differentiates in forward mode (Dual numbers) and adds a vector site's
    tangents in element order — the new values and the weight agree BIT FOR BIT.  `Regenerate(S["theta"])` ON the long site
    runs as a counted loop as well, and so does `HMC(S["theta"])` ON the long site — alone and together with the scalars; a
    latent vector read through elementwise arithmetic by two later sites, through `jnp.sum` / `jnp.mean`, and GATHERED at a
    table of group indices (`theta[group]`: a random-effects likelihood); ONE trace.  A long vector read at ONE index is
    refused, naming the site."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S
    from genjax_amd import numpy as jnp
    from genjax_amd.inference.requests import HMC
    xs = np.linspace(-1, 1, npts).astype(f32)
    ys = (f32(0.7) * xs + f32(0.2)).astype(f32)

    @G.gen
    def reg():
        a = G.normal(0.0, 2.0) @ "a"
        b = G.normal(0.0, 2.0) @ "b"
        G.normal(a * jnp.array(xs) + b, 0.5) @ "y"
        return a

    @O.gen
    def oreg():
        a = O.normal(f32(0.0), f32(2.0)) @ "a"
        b = O.normal(f32(0.0), f32(2.0)) @ "b"
        O.normal(_col(a) * xs + _col(b), f32(0.5)) @ "y"
        return a
    # (at many particles the oracle runs the FIRST 64 of them — particles are independent, the product's first rows are theirs)
    Kr = min(K, 64)
    tr, _ = reg.importance(G.split(G.key(seed), K), C["y"].set(jnp.array(ys)), ())
    otr, _ = oreg.importance(O.split(O.key(seed), K)[:Kr], O.C.d({"y": ys}), ())
    new, w, _, _ = HMC(S["a"] | S["b"], 1e-3, L=L).edit(G.split(G.key(seed + 1), K), tr, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.split(O.key(seed + 1), K)[:Kr], otr, ["a", "b"], 1e-3, L, ())
    for a_ in ("a", "b"):
        assert same(npv(new.get_choices()[a_])[:Kr], onew.get_choices()[a_]), ("regression", a_)
    assert same(npv(w)[:Kr], ow), ("regression weight", npv(w)[:Kr], np.asarray(ow))
    # 8-schools at J schools
    sig = np.linspace(9, 18, J).astype(f32)
    yj = np.linspace(-3, 28, J).astype(f32)

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        log_tau = G.normal(0.0, 1.0) @ "log_tau"
        theta = G.normal(mu * jnp.ones(J), jnp.exp(log_tau) * jnp.ones(J)) @ "theta"
        G.normal(theta, jnp.array(sig)) @ "y"
        return mu

    @O.gen
    def oschools():
        mu = O.normal(f32(0.0), f32(5.0)) @ "mu"
        log_tau = O.normal(f32(0.0), f32(1.0)) @ "log_tau"
        theta = O.normal(_col(mu) * np.ones(J, f32), _col(O.exp(log_tau)) * np.ones(J, f32)) @ "theta"
        O.normal(theta, sig) @ "y"
        return mu
    Ko = min(K, 64)
    okeys = lambda s_: O.split(O.key(s_), K)[:Ko]
    cut = lambda v: npv(v)[:Ko]
    tr, _ = schools.importance(G.split(G.key(seed), K), C["y"].set(jnp.array(yj)), ())
    otr, _ = oschools.importance(okeys(seed), O.C.d({"y": yj}), ())
    many = K > 64          # (specialised kernels: every distinct request is a hiprtc compile of a long program — fewer of them)
    for sel, osel in ((S["mu"], ["mu"]), (S["mu"] | S["log_tau"], ["mu", "log_tau"]))[1 if many else 0:]:
        new, w, _, _ = HMC(sel, 1e-3, L=L).edit(G.split(G.key(seed + 2), K), tr, Diff.no_change(()))
        onew, ow = O.hmc_edit(okeys(seed + 2), otr, osel, 1e-3, L, ())
        for a_ in osel:
            assert same(cut(new.get_choices()[a_]), onew.get_choices()[a_]), ("schools", osel, a_)
        assert same(cut(w), ow), ("schools weight", osel)
    new, wr, _, _ = Regenerate(S["theta"]).edit(G.split(G.key(seed + 3), K), tr, Diff.no_change(()))
    onew, owr = oschools.regenerate(okeys(seed + 3), otr, O.selection("theta"), ())[:2]
    assert same(cut(new.get_choices()["theta"]), onew.get_choices()["theta"]) and np.array_equal(cut(wr), np.asarray(owr, f32), equal_nan=True)
    # Rejuvenate ON the long vector site (rejuvenate.py:70-94): the proposal's J draws, its forward / backward densities and
    # the re-scoring of the proposed vector each run as ONE counted loop (before round 6: unrolled — 173 launches at J = 1 000)
    from genjax_amd import StaticRequest
    rq = StaticRequest({"theta": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
    orq = {"theta": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), f32(0.5)))}
    new, wj, _, _ = rq.edit(G.split(G.key(seed + 10), K), tr, Diff.no_change(()))
    onew, owj = oschools.edit_static(okeys(seed + 10), otr, orq, ())
    assert same(cut(new.get_choices()["theta"]), onew.get_choices()["theta"]), "rejuvenate on the long site: values"
    assert np.array_equal(cut(wj), np.asarray(owj, f32), equal_nan=True) and same(cut(new.get_score()), onew.get_score()), "rejuvenate on the long site"
    # HMC ON the long vector site (round 6): positions / momenta / gradients as vectors in memory, every leapfrog stage one
    # counted loop per vector-valued site that reads them; alone, together with the scalars, at a larger step
    # (the oracle differentiates a vector element by element — J forward passes per gradient: at many particles it runs the
    #  first 64 of them, which are the first 64 rows of the product's result: particles are independent)
    otr_v = otr
    for sel, osel, eps in ((S["theta"], ["theta"], 1e-2), (S["mu"] | S["theta"], ["mu", "theta"], 1e-3),
                           (S["mu"] | S["log_tau"] | S["theta"], ["mu", "log_tau", "theta"], 5e-2))[2 if many else 0:]:
        new, w, _, _ = HMC(sel, eps, L=L).edit(G.split(G.key(seed + 4), K), tr, Diff.no_change(()))
        onew, ow = O.hmc_edit(O.split(O.key(seed + 4), K)[:Ko], otr_v, osel, eps, L, ())
        if 64 < J <= 200 and K <= 16 and osel == ["theta"]:
            # the oracle's one-pass gradient of a long vector (tangent 1 everywhere, kept per element) against its J
            # one-hot passes: the same numbers
            oslow, owslow = O.hmc_edit(O.split(O.key(seed + 4), K)[:Ko], otr_v, osel, eps, L, (), one_hot_max=10 ** 9)
            assert np.array_equal(oslow.get_choices()["theta"], onew.get_choices()["theta"]) and np.array_equal(owslow, ow)
        for a_ in osel:
            assert same(npv(new.get_choices()[a_])[:Ko], onew.get_choices()[a_]), ("schools, vector", osel, a_)
        assert same(npv(w)[:Ko], ow), ("schools weight, vector", osel, npv(w)[:Ko], np.asarray(ow))
        assert same(npv(new.get_score())[:Ko], onew.get_score()), ("schools score, vector", osel)
        assert np.array_equal(npv(new.get_choices()["y"])[:Ko], np.broadcast_to(yj, (Ko, J)))
    if K > 64:
        return
    if J <= 16:
        return
    # the latent vector read through elementwise arithmetic by TWO later sites (three contributions to its gradient)
    xs2 = np.linspace(0.5, 1.5, J).astype(f32)

    @G.gen
    def two():
        theta = G.normal(jnp.zeros(J), 2.0 * jnp.ones(J)) @ "theta"
        G.normal(theta * jnp.array(xs2) + 0.25, jnp.array(sig)) @ "y"
        G.normal(jnp.tanh(theta), 0.5) @ "z"
        return None

    @O.gen
    def otwo():
        theta = O.normal(np.zeros(J, f32), f32(2.0) * np.ones(J, f32)) @ "theta"
        O.normal(theta * xs2 + f32(0.25), sig) @ "y"
        O.normal(O.tanh(theta), f32(0.5)) @ "z"
        return None
    zj = np.linspace(-0.5, 0.5, J).astype(f32)
    tr2, _ = two.importance(G.split(G.key(seed + 5), K), C["y"].set(jnp.array(yj)) | C["z"].set(jnp.array(zj)), ())
    otr2, _ = otwo.importance(O.split(O.key(seed + 5), K), O.C.d({"y": yj, "z": zj}), ())
    new, w, _, _ = HMC(S["theta"], 2e-2, L=L).edit(G.split(G.key(seed + 6), K), tr2, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.split(O.key(seed + 6), K), otr2, ["theta"], 2e-2, L, ())
    assert same(new.get_choices()["theta"], onew.get_choices()["theta"]) and same(w, ow) and same(new.get_score(), onew.get_score())
    # ... and ONE trace (no particle batch)
    tr1, _ = schools.importance(G.key(seed + 7), C["y"].set(jnp.array(yj)), ())
    otr1, _ = oschools.importance(O.key(seed + 7), O.C.d({"y": yj}), ())
    new, w, _, _ = HMC(S["theta"], 1e-2, L=L).edit(G.key(seed + 8), tr1, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.key(seed + 8), otr1, ["theta"], 1e-2, L, ())
    assert same(new.get_choices()["theta"], onew.get_choices()["theta"]) and same(w, ow)
    # ... Rejuvenate on the long site of ONE trace (site by site from 65 elements: the proposal's draws and densities on the
    # launch axis — sitewise.vector_site_update)
    new, wj, _, _ = rq.edit(G.key(seed + 16), tr1, Diff.no_change(()))
    onew, owj = oschools.edit_static(O.key(seed + 16), otr1, orq, ())
    assert same(new.get_choices()["theta"], onew.get_choices()["theta"]) and same(wj, owj) and same(new.get_score(), onew.get_score())
    # the latent vector read through SUMS as well (`normal(jnp.sum(theta), 3)`, `normal(jnp.mean(theta * theta), 0.5)`): the
    # sum's own loop stores d element / d theta_j, scaled afterwards by the adjoint the sum reaches the score with
    @G.gen
    def summed():
        theta = G.normal(jnp.zeros(J), 2.0 * jnp.ones(J)) @ "theta"
        G.normal(theta, jnp.array(sig)) @ "y"
        G.normal(jnp.sum(theta), 3.0) @ "tot"
        G.normal(jnp.sum(theta * theta), 25.0) @ "m2"
        return None

    @O.gen
    def osummed():
        theta = O.normal(np.zeros(J, f32), f32(2.0) * np.ones(J, f32)) @ "theta"
        O.normal(theta, sig) @ "y"
        O.normal(O.sum_vector(theta), f32(3.0)) @ "tot"
        O.normal(O.sum_vector(theta * theta), f32(25.0)) @ "m2"
        return None
    y3 = np.linspace(-3, 3, J).astype(f32)
    # (bit for bit while a sum's result feeds the next site AS IT IS: one multiplication per contribution on either side; a
    #  scaled sum — `jnp.mean` — multiplies in a different order in reverse mode (the build, jax) than in forward mode (the
    #  oracle): the last bit of a gradient may differ there, checked to 1e-6 relative below)
    tr3, _ = summed.importance(G.split(G.key(seed + 9), K), C["y"].set(jnp.array(y3)) | C["tot"].set(0.5) | C["m2"].set(float(4 * J)), ())
    otr3, _ = osummed.importance(O.split(O.key(seed + 9), K), O.C.d({"y": y3, "tot": f32(0.5), "m2": f32(4 * J)}), ())
    new, w, _, _ = HMC(S["theta"], 1e-2, L=L).edit(G.split(G.key(seed + 11), K), tr3, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.split(O.key(seed + 11), K), otr3, ["theta"], 1e-2, L, ())
    assert same(new.get_choices()["theta"], onew.get_choices()["theta"]) and same(w, ow) and same(new.get_score(), onew.get_score())
    @G.gen
    def meaned():
        theta = G.normal(jnp.zeros(J), 2.0 * jnp.ones(J)) @ "theta"
        G.normal(jnp.mean(jnp.tanh(theta)), 0.5) @ "m"
        return None

    @O.gen
    def omeaned():
        theta = O.normal(np.zeros(J, f32), f32(2.0) * np.ones(J, f32)) @ "theta"
        O.normal(O.sum_vector(O.tanh(theta)) / f32(J), f32(0.5)) @ "m"
        return None
    tr5, _ = meaned.importance(G.split(G.key(seed + 12), K), C["m"].set(0.1), ())
    otr5, _ = omeaned.importance(O.split(O.key(seed + 12), K), O.C.d({"m": f32(0.1)}), ())
    new, w, _, _ = HMC(S["theta"], 1e-2, L=L).edit(G.split(G.key(seed + 13), K), tr5, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.split(O.key(seed + 13), K), otr5, ["theta"], 1e-2, L, ())
    assert np.allclose(npv(new.get_choices()["theta"]), onew.get_choices()["theta"], rtol=1e-6, atol=1e-6)
    assert np.allclose(npv(w), ow, rtol=1e-4, atol=1e-5)
    # a random-effects likelihood: `normal(theta[group] + x, 0.5) @ "y"` with `group` a table of N = 5 J group labels — the
    # gather's adjoint is a scatter-add (static._scatter_add: the consuming loop stores d term_i, two more loops add them up
    # per group); the oracle differentiates element by element (its one-pass mode does not cover gathers)
    if J > 200:          # (J x N compares per particle and pass, and J forward passes of the oracle per gradient)
        return
    Ng = 5 * J if J <= 40 else 2 * J
    rng = np.random.default_rng(seed + 20)
    grp = rng.integers(0, J, Ng).astype(np.int32)
    xg = rng.normal(size=Ng).astype(f32)

    @G.gen
    def grouped():
        theta = G.normal(jnp.zeros(J), 2.0 * jnp.ones(J)) @ "theta"
        G.normal(theta[jnp.array(grp)] + jnp.array(xg), 0.5) @ "y"
        return None

    @O.gen
    def ogrouped():
        theta = O.normal(np.zeros(J, f32), f32(2.0) * np.ones(J, f32)) @ "theta"
        O.normal(theta[..., grp] + xg, f32(0.5)) @ "y"
        return None
    yg = rng.normal(size=Ng).astype(f32)
    Kg = min(K, 5)
    tr6, _ = grouped.importance(G.split(G.key(seed + 14), Kg), C["y"].set(jnp.array(yg)), ())
    otr6, _ = ogrouped.importance(O.split(O.key(seed + 14), Kg), O.C.d({"y": yg}), ())
    new, w, _, _ = HMC(S["theta"], 1e-3, L=L).edit(G.split(G.key(seed + 15), Kg), tr6, Diff.no_change(()))
    onew, ow = O.hmc_edit(O.split(O.key(seed + 15), Kg), otr6, ["theta"], 1e-3, L, (), one_hot_max=10 ** 9)
    assert same(new.get_choices()["theta"], onew.get_choices()["theta"]) and same(w, ow) and same(new.get_score(), onew.get_score()), "gather"
    # what stays refused: one element picked out of the vector (a static or a traced index)
    @G.gen
    def picked():
        theta = G.normal(jnp.zeros(J), jnp.ones(J)) @ "theta"
        G.normal(theta[3], 1.0) @ "y"
        return None
    tr4 = picked.simulate(G.split(G.key(seed + 9), K), ())
    try:
        HMC(S["theta"], 1e-3, L=L).edit(G.split(G.key(seed + 4), K), tr4, Diff.no_change(()))
        raise AssertionError("HMC on a long vector read at one index should raise")
    except NotImplementedError as e:
        assert "vector-valued site" in str(e) and "theta" in str(e), str(e)
