"""Shared parity drivers: run the same workload through genjax_amd (HIP, or the
CPU hostsim harness in `-m "not gpu"` tests) and through the CPU oracle."""
from __future__ import annotations

import math

import numpy as np
import torch

from oracle import genjax_oracle as O


def oracle_bootstrap_sweep(init, step, n, T, ys, run_key, kind=O.SYSTEMATIC, step_extra=None):
    """The oracle's statement of smc.BootstrapSweep (same build-defined key
    schedule: step key = fold_in(run_key, t); (k_prop, k_res, k_mh) = split(., 3))."""
    step_extra = step_extra or (lambda t: ())
    x, anc, log_ml = None, None, 0.0
    hist = []
    for t in range(T):
        ks = O.split(O.fold_in(run_key, t), 3)
        k_prop, k_res = ks[0], ks[1]
        keys = O.split(k_prop, n)
        obs = O.C.d({"y": np.float32(ys[t])})
        if t == 0:
            tr, w = init.importance(keys, obs, ())
        else:
            tr, w = step.importance(keys, obs, (x[anc],) + tuple(step_extra(t)))
        x = np.asarray(tr.get_retval(), np.float32)
        lw = np.asarray(w, np.float32)
        cdf, total, M, shift = O.weight_cdf(lw)
        anc = O.ancestors_of_kind(kind, k_res, cdf)
        log_ml += O.log_ml_increment(M, total, shift, n)
        hist.append(dict(x=x, lw=lw, cdf=cdf, total=total, M=M, anc=anc))
    return dict(log_ml=log_ml, x=x, lw=lw, anc=anc, hist=hist)


def check_lgssm_sweep(n=4096, T=5, seed=314159, capture=False, specialize=True, resample="systematic",
                      noise_ahead=None, fuse_resample=None):
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference.smc import BootstrapSweep
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    sw = BootstrapSweep(init, step, n, T, specialize=specialize, resample=resample, noise_ahead=noise_ahead,
                        fuse_resample=fuse_resample).prepare(G.key(seed), torch.from_numpy(ys))
    if noise_ahead is not None:
        assert sw.noise_ahead == noise_ahead, "the sweep did not take the requested (one- / two-stream) form"
    if fuse_resample is not None:
        assert sw.fuse == fuse_resample, "the sweep did not take the requested (one- / two-launch) form"
    if capture:
        sw.capture()
        sw.launch()              # a second replay must not see what the first one left (count buffers, ring slots)
    sw.launch()
    log_ml = sw.log_ml()
    x, lw, anc = sw.state()
    oi, os_ = workloads.make_lgssm(O)
    ref = oracle_bootstrap_sweep(oi, os_, n, T, ys, O.key(seed),
                                 kind={"systematic": O.SYSTEMATIC, "stratified": O.STRATIFIED,
                                       "multinomial_tiled": O.MULTINOMIAL_TILED,
                                       "multinomial_sorted": O.MULTINOMIAL_SORTED}[resample])
    return dict(
        log_ml=log_ml, log_ml_oracle=ref["log_ml"], kalman=workloads.kalman_log_ml(ys),
        ancestors_equal=bool(np.array_equal(anc.cpu().numpy(), ref["anc"])),
        x_equal=bool(np.array_equal(x.cpu().numpy(), ref["x"])),
        lw_max_abs_diff=float(np.max(np.abs(lw.cpu().numpy() - ref["lw"]))),
        totals_equal=bool(np.array_equal(sw.totals.cpu().numpy().view(np.uint64),
                                         np.array([h["total"] for h in ref["hist"]], dtype=np.uint64))),
    )


def check_tuple_state_sweep(n=3000, T=5, seed=13, capture=False, specialize=False, noise_ahead=None):
    """BootstrapSweep over a step model with THREE latent sites whose state is a tuple of two of them
    (the sweep is not limited to `trace = {state, obs}`)."""
    import genjax_amd as G
    from genjax_amd.inference.smc import BootstrapSweep
    ys = np.random.default_rng(seed).normal(size=T).astype(np.float32)

    def mk(g):
        @g.gen
        def init():
            a = g.normal(0.0, 1.0) @ "a"
            b = g.normal(a, 1.0) @ "b"
            g.normal(a + b, 1.0) @ "y"
            return (a, b)

        @g.gen
        def step(s):
            a0, b0 = s
            a = g.normal(0.9 * a0, 0.5) @ "a"
            b = g.normal(0.5 * b0 + 0.1 * a, 0.5) @ "b"
            c = g.normal(0.0, 1.0) @ "c"
            g.normal((a + b) + 0.1 * c, 1.0) @ "y"
            return (a, b)
        return init, step
    (init, step), (oi, os_) = mk(G), mk(O)
    sw = BootstrapSweep(init, step, n, T, specialize=specialize, noise_ahead=noise_ahead).prepare(G.key(seed), torch.from_numpy(ys))
    if noise_ahead is not None:
        assert sw.noise_ahead == noise_ahead
        if noise_ahead:
            assert len(sw.p_step.noise) == 3 and len(sw.p_init.noise) == 2     # every latent normal site's draw
    if capture:
        sw.capture()
    sw.launch()
    # the oracle's statement of the same sweep with a tuple state
    x, anc, log_ml = None, None, 0.0
    for t in range(T):
        ks = O.split(O.fold_in(O.key(seed), t), 3)
        keys = O.split(ks[0], n)
        obs = O.C.d({"y": np.float32(ys[t])})
        if t == 0:
            tr, w = oi.importance(keys, obs, ())
        else:
            tr, w = os_.importance(keys, obs, (tuple(v[anc] for v in x),))
        x = tuple(np.asarray(v, np.float32) for v in tr.get_retval())
        lw = np.asarray(w, np.float32)
        cdf, total, M, shift = O.weight_cdf(lw)
        anc = O.ancestors(O.SYSTEMATIC, ks[1], cdf)
        log_ml += O.log_ml_increment(M, total, shift, n)
    xs, lw_d, anc_d = sw.state()
    assert np.array_equal(anc_d.cpu().numpy(), anc)
    for d in range(2):
        assert np.array_equal(xs[d].cpu().numpy(), x[d])
    assert np.array_equal(lw_d.cpu().numpy(), lw)
    assert sw.log_ml() == log_ml
    return dict(log_ml=log_ml)


def check_nlssm_mh(n=2000, T=4, seed=7):
    """BASELINE config 3 in miniature: nonlinear SSM, bootstrap SMC with one
    Rejuvenate (Gaussian drift, sigma 0.5) MH sweep on x_t after each resample,
    through the functional API (resample -> rejuvenate -> extend) vs the oracle."""
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference import smc
    ys = workloads.nlssm_data(T)
    init, step = workloads.make_nlssm(G)
    oi, ost = workloads.make_nlssm(O)
    req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
    oreq = {"x": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.5)))}
    key, okey = G.key(seed), O.key(seed)
    out = dict(ok=True, steps=[])
    coll = otr = olw = None
    for t in range(T):
        kp, kr, km = G.split(G.fold_in(key, t), 3)
        oks = O.split(O.fold_in(okey, t), 3)
        obs, oobs = G.ChoiceMap.kw(y=float(ys[t])), O.C.kw(y=np.float32(ys[t]))
        if t == 0:
            coll = smc.ImportanceK(G.Target(init, (), obs), k_particles=n).run_smc(kp)
            oc = O.ImportanceK(O.Target(oi, (), oobs), n).run_smc(oks[0])
            otr, olw = oc.get_particles(), oc.get_log_weights()
        else:
            coll = smc.resample(kr, coll, "systematic")
            cdf, total, M, shift = O.weight_cdf(olw)
            anc = O.ancestors(O.SYSTEMATIC, oks[1], cdf)
            otr = O.gather_trace(otr, anc)
            anc_ok = bool(np.array_equal(coll.ancestors.cpu().numpy(), anc))
            coll = smc.rejuvenate(km, coll, req)
            gf = otr.get_gen_fn()
            otr, oacc, _ = O.rejuvenate(oks[2], otr, lambda k, tr_: gf.edit_static(k, tr_, oreq, tr_.get_args()))
            acc_ok = bool(np.array_equal(coll.accept.cpu().numpy(), oacc))
            coll = smc.extend(kp, coll, step, lambda tr_: (tr_.get_retval(), float(t)), obs)
            otr, olw = ost.importance(O.split(oks[0], n), oobs, (np.asarray(otr.get_retval(), np.float32), np.float32(t)))
            out["steps"].append(dict(anc=anc_ok, acc=acc_ok, acc_rate=float(oacc.mean())))
            out["ok"] &= anc_ok and acc_ok
        x_ok = bool(np.array_equal(coll.get_particles().get_retval().cpu().numpy(), np.asarray(otr.get_retval(), np.float32)))
        w_ok = bool(np.array_equal(coll.get_log_weights().cpu().numpy(), np.asarray(olw, np.float32)))
        out["ok"] &= x_ok and w_ok
    return out


# ---------------------------------------------------------------------------
# plates: eight schools written with Vmap / repeat
# ---------------------------------------------------------------------------
SCHOOL_SIGMA = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
SCHOOL_Y = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)


def _school(g):
    @g.gen
    def school(mu, tau, sigma):
        theta = g.normal(mu, tau) @ "theta"
        g.normal(theta, sigma) @ "y"
        return theta
    return school


def check_plates(n=257, seed=4):
    """importance / simulate / assess of plate models: product == oracle, bit for bit."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    school, oschool = _school(G), _school(O)

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        log_tau = G.normal(0.0, 1.0) @ "log_tau"
        return school.vmap(in_axes=(None, None, 0))(mu, jnp.exp(log_tau), jnp.array(SCHOOL_SIGMA)) @ "schools"

    @O.gen
    def o_schools():
        mu = O.normal(0.0, 5.0) @ "mu"
        log_tau = O.normal(0.0, 1.0) @ "log_tau"
        return O.Vmap(oschool, in_axes=(None, None, 0))(mu, O.exp(log_tau), np.array(SCHOOL_SIGMA, np.float32)) @ "schools"

    tr, w = schools.importance(G.split(G.key(seed), n), C["schools", :, "y"].set(SCHOOL_Y), ())
    tro, wo = o_schools.importance(O.split(O.key(seed), n), O.C.d({("schools", "y"): SCHOOL_Y}), ())
    th = tr.get_choices()["schools", "theta"]
    assert tuple(th.shape) == (n, 8)
    assert np.array_equal(th.cpu().numpy(), tro.get_choices()["schools", "theta"])
    assert np.array_equal(w.cpu().numpy(), wo)
    assert np.array_equal(tr.get_score().cpu().numpy(), tro.get_score())
    assert np.array_equal(tr.get_retval().cpu().numpy(), tro.get_retval())
    # per-element scores of the inner trace keep the plate axis (VmapTrace.inner)
    sub = tr.get_subtrace("schools")
    assert tuple(sub.inner.get_score().shape) == (n, 8)
    assert np.array_equal(sub.get_score().cpu().numpy(), tro.get_subtrace("schools").get_score())

    @G.gen
    def mixed():
        mu = G.normal(0.0, 5.0) @ "mu"
        G.normal.vmap(in_axes=(0, None))(jnp.array([1.0, 2.0, 3.0]) + mu, 1.0) @ "xs"
        return school.repeat(n=3)(mu, 2.0, 1.0) @ "zs"

    @O.gen
    def o_mixed():
        mu = O.normal(0.0, 5.0) @ "mu"
        O.Vmap(O.normal, in_axes=(0, None))(np.array([1.0, 2.0, 3.0], np.float32) + mu[..., None], 1.0) @ "xs"
        return O.Repeat(oschool, 3)(mu, 2.0, 1.0) @ "zs"

    tr = mixed.simulate(G.split(G.key(seed + 5), n), ())
    tro = o_mixed.simulate(O.split(O.key(seed + 5), n), ())
    for a in [("mu",), ("xs",), ("zs", "theta"), ("zs", "y")]:
        assert np.array_equal(np.asarray(tr.get_choices()[a].cpu()), tro.get_choices()[a]), a
    assert np.array_equal(tr.get_score().cpu().numpy(), tro.get_score())
    assert np.array_equal(tr.get_retval().cpu().numpy(), tro.get_retval())
    s, _ = mixed.assess(tr.get_choices(), ())
    so, _ = o_mixed.assess(tro.get_choices(), (), (n,))
    assert np.array_equal(s.cpu().numpy(), so)
    assert np.array_equal(s.cpu().numpy(), tr.get_score().cpu().numpy())      # assess(simulate) == score

    # a Vmap used directly: keys are split(key, n) of the caller's key (vmap.py:186)
    v = school.vmap(in_axes=(None, None, 0))
    t2 = v.simulate(G.split(G.key(seed + 1), 16), (0.0, 1.0, jnp.array(SCHOOL_SIGMA)))
    t2o = O.Vmap(oschool, in_axes=(None, None, 0)).simulate(O.split(O.key(seed + 1), 16),
                                                             (0.0, 1.0, np.array(SCHOOL_SIGMA, np.float32)))
    assert np.array_equal(np.asarray(t2.get_choices()["theta"].cpu()), t2o.get_choices()["theta"])
    assert np.array_equal(t2.get_score().cpu().numpy(), t2o.get_score())


# ---------------------------------------------------------------------------
# global resampling across ranks, emulated in ONE process through the C-ABI:
# every "rank" runs gmx_weight_cdf / gmx_shard_plan / gmx_shard_route on its
# shard; the all-to-all is a block copy.  Result must equal the oracle's
# single-population resample, for any world size and capacity >= need.
# ---------------------------------------------------------------------------
def check_shard_route(n=1000, world=4, capacity=None, kind=None, seed=0, skew=0.0, dead=False, fused=False, spike=0.0):
    from ctypes import c_uint32
    from genjax_amd import _lib
    be = _lib.get()
    dev = be.device
    kind = O.SYSTEMATIC if kind is None else kind
    N = n * world
    C = n if capacity is None else capacity
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, 2, N).astype(np.float32)
    lw += (skew * (np.arange(N) // n)).astype(np.float32)          # later ranks heavier
    if spike:
        lw[n + n // 2 + 7] += np.float32(spike)                    # one heavy particle on rank 1: its run spans ranks
    if dead:
        lw[:] = -np.inf
    x = rng.normal(0, 1, N).astype(np.float32)
    k = O.key(seed + 11)
    if dead:
        anc_ref = np.full(N, N - 1, dtype=np.int64)
        shift = O.cdf_shift(N)
    else:
        cdf_ref, total_ref, M_ref, shift = O.weight_cdf(lw)
        anc_ref = O.ancestors(kind, k, cdf_ref)
    want = x[anc_ref]

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    kk = (c_uint32 * 2)(int(k[0]), int(k[1]))
    max_d = T(np.array([lw.max()], np.float32))
    totals = torch.zeros((world,), dtype=torch.int64, device=dev)
    cdfs, sends, idxs, xexts, plans = [], [], [], [], []
    ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
    lws = [T(lw[r * n:(r + 1) * n]) for r in range(world)]
    if fused in ("tiles", "stats"):        # the two-collective form: tile statistics -> totals + global max, no CDF array
        sb = int(be.c.gmx_shard_stats_bytes(n))
        pad = (n + 1023) // 1024
        pad += pad & 1
        stats_all = torch.zeros((world * sb,), dtype=torch.uint8, device=dev)
        for r in range(world):
            blk = stats_all[r * sb:(r + 1) * sb]
            be.check(be.c.gmx_tile_stats(be.ptr(lws[r]), n, shift, be.ptr(blk[pad * 8:]), be.ptr(blk), be.stream()),
                     "gmx_tile_stats")
        max_d = torch.zeros((1,), dtype=torch.float32, device=dev)
        be.check(be.c.gmx_shard_totals(be.ptr(stats_all), world, n, be.ptr(totals), be.ptr(max_d), be.stream()),
                 "gmx_shard_totals")
        if not dead:
            assert float(max_d.item()) == M_ref
    else:
        for r in range(world):
            cdf = torch.zeros((n,), dtype=torch.int64, device=dev)
            be.check(be.c.gmx_weight_cdf(be.ptr(lws[r]), n, shift, None, 0, be.ptr(max_d), be.ptr(cdf),
                                         be.ptr(totals[r:r + 1]), be.ptr(ws), be.stream()), "gmx_weight_cdf")
            cdfs.append(cdf)
    for r in range(world):
        plan = torch.zeros((int(be.c.gmx_shard_plan_words(world)),), dtype=torch.int64, device=dev)
        tot = torch.zeros((1,), dtype=torch.int64, device=dev)
        xe = torch.zeros((n + world * C,), dtype=torch.float32, device=dev)
        xe[:n] = T(x[r * n:(r + 1) * n])
        send = torch.full((world * C,), float("nan"), dtype=torch.float32, device=dev)
        idx = torch.full((n,), -1, dtype=torch.int32, device=dev)
        if fused == "stats":     # gmx_shard_step_fused: totals + plan + route straight from the gathered table
            mx = torch.zeros((1,), dtype=torch.float32, device=dev)
            be.check(be.c.gmx_shard_step_fused(kind, kk, be.ptr(stats_all), be.ptr(plan), be.ptr(tot), be.ptr(lws[r]),
                                               be.ptr(mx), shift, r, world, n, C, be.ptr(xe), be.ptr(send), be.ptr(idx),
                                               be.stream()), "gmx_shard_step_fused")
            if not dead:
                assert float(mx.item()) == M_ref
                assert int(tot.item()) & 0xFFFFFFFFFFFFFFFF == total_ref
        elif fused == "tiles":
            be.check(be.c.gmx_shard_step_tiles(kind, kk, be.ptr(totals), be.ptr(plan), be.ptr(tot), be.ptr(lws[r]),
                                               be.ptr(stats_all[r * sb:(r + 1) * sb]), be.ptr(max_d), shift, r, world,
                                               n, C, be.ptr(xe), be.ptr(send), be.ptr(idx), be.stream()),
                     "gmx_shard_step_tiles")
        elif fused:     # gmx_shard_step: plan + route in one launch
            be.check(be.c.gmx_shard_step(kind, kk, be.ptr(totals), be.ptr(plan), be.ptr(tot), be.ptr(cdfs[r]), r, world,
                                         n, C, be.ptr(xe), be.ptr(send), be.ptr(idx), be.stream()), "gmx_shard_step")
        else:
            be.check(be.c.gmx_shard_plan(kind, kk, be.ptr(totals), r, world, n, be.ptr(plan), be.ptr(tot), be.stream()),
                     "gmx_shard_plan")
            be.check(be.c.gmx_shard_route(kind, kk, be.ptr(plan), be.ptr(cdfs[r]), r, world, n, C, be.ptr(xe),
                                          be.ptr(send), be.ptr(idx), be.stream()), "gmx_shard_route")
        plans.append(plan); sends.append(send); idxs.append(idx); xexts.append(xe)
    overflow = any(int(p[2].item()) for p in plans)
    if not dead:
        assert int(plans[0][0].item()) & 0xFFFFFFFFFFFFFFFF == total_ref
    b = plans[0][4:].cpu().numpy()
    assert b[0] == 0 and b[-1] == N and np.all(np.diff(b) >= 0)
    for p in plans[1:]:
        assert np.array_equal(p[4:].cpu().numpy(), b)
    # slots sourced by rank s = number of oracle ancestors living on rank s
    assert np.array_equal(np.diff(b), np.bincount(anc_ref // n, minlength=world))
    if overflow:
        return {"overflow": True, "bounds": b}
    out = []
    for d in range(world):
        for s_ in range(world):                                   # the all-to-all
            xexts[d][n + s_ * C:n + (s_ + 1) * C] = sends[s_][d * C:(d + 1) * C]
        i = idxs[d].cpu().numpy()
        assert i.min() >= 0
        out.append(xexts[d].cpu().numpy()[i])
    got = np.concatenate(out)
    assert np.array_equal(got, want)
    return {"overflow": False, "bounds": b}


# ---------------------------------------------------------------------------
# conditional SMC / estimate_logpdf / proposals
# ---------------------------------------------------------------------------
def check_csmc(k=257, seed=3):
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Target
    from genjax_amd.inference.smc import ChangeTarget, Importance, ImportanceK
    from genjax_amd.inference.sp import marginal

    def mk(g):
        @g.gen
        def model():
            p = g.beta(2.0, 2.0) @ "p"
            v = g.flip(p) @ "v"
            g.normal(p, 1.0) @ "z"
            return v
        return model
    m, om = mk(G), mk(O)
    tgt, otgt = Target(m, (), C["v"].set(True)), O.Target(om, (), O.C.d({"v": True}))
    alg, oalg = ImportanceK(tgt, k_particles=k), O.ImportanceK(otgt, k)
    ret, oret = C.d({"p": 0.3, "z": 0.5}), O.C.d({"p": np.float32(0.3), "z": np.float32(0.5)})
    # run_csmc: K-1 fresh particles (keys split(sub, K-1)) + the retained choices in slot K-1
    pc, opc = alg.run_csmc(G.key(seed), ret), O.importancek_run_csmc(oalg, O.key(seed), oret)
    assert np.array_equal(pc.get_log_weights().cpu().numpy(), opc.get_log_weights())
    for a in ("p", "z"):
        assert np.array_equal(np.asarray(pc.get_particles().get_choices()[a].cpu()), opc.get_particles().get_choices()[a])
    assert float(pc.get_particles().get_choices()["p"][-1]) == np.float32(0.3)
    # ChangeTarget.run_csmc re-weights with split(key, K)
    ct, oct_ = ChangeTarget(alg, tgt), O.ChangeTarget(oalg, otgt)
    pc2, opc2 = ct.run_csmc(G.key(seed + 1), ret), O.changetarget_run_csmc(oct_, O.key(seed + 1), oret)
    assert np.array_equal(pc2.get_log_weights().cpu().numpy(), opc2.get_log_weights())
    # estimate_logpdf: same sampled particle; the estimate differs only by the summation
    # order of logsumexp (tree on the device, sequential in the oracle): |diff| <= 1e-5
    e, oe = alg.estimate_logpdf(G.key(seed + 2), ret, tgt), O.estimate_logpdf(oalg, O.key(seed + 2), oret, otgt)
    assert abs(float(e) - float(oe)) <= 1e-5
    # Importance.run_csmc: one particle = the retained choices
    pc1 = Importance(tgt).run_csmc(G.key(seed), ret)
    s1, _ = m.assess(C.d({"p": 0.3, "z": 0.5, "v": True}), ())
    assert tuple(pc1.get_log_weights().shape) == (1,)
    assert abs(float(pc1.get_log_weights()[0]) - float(s1)) < 1e-6        # everything constrained: w = score
    # a nested algorithm as the proposal q (smc.py:301-305): weights = target score - q's estimate
    nested = ImportanceK(tgt, q=ImportanceK(tgt, k_particles=3), k_particles=k)
    pcq = nested.run_smc(G.key(seed + 4))
    assert tuple(pcq.get_log_weights().shape) == (k,) and bool(torch.isfinite(pcq.get_log_weights()).all())
    pcq2 = nested.run_csmc(G.key(seed + 5), ret)
    assert float(pcq2.get_particles().get_choices()["p"][-1]) == np.float32(0.3)

    # a Marginal as the proposal, reading the Target it is given (sp.py:217-240, literally:
    # with the default full selection the returned weight is project(~all) = 0)
    @marginal()
    @G.gen
    def proposal(target):
        a = 3.0 if bool(target.constraint["v"]) else 2.0
        return G.beta(a, 5.0 - a) @ "p"
    tgt2 = Target(mk2(G), (), C["v"].set(True))
    w, chm = proposal.random_weighted(G.split(G.key(seed), 8), tgt2)
    assert np.array_equal(w.cpu().numpy(), np.zeros(8, np.float32)) and tuple(chm["p"].shape) == (8,)
    lp = proposal.estimate_logpdf(G.key(seed), C["p"].set(0.4), tgt2)
    assert abs(float(lp) - float(O.beta.assess(O.C.choice(np.float32(0.4)), (np.float32(3.0), np.float32(2.0)))[0])) < 1e-6
    ImportanceK(tgt2, q=proposal, k_particles=64).run_smc(G.key(seed))


def check_change_target_from_subset_constraints(n=5, k=33, seed=3):
    """A7 / A9: `ChangeTarget.run_smc` (ref smc.py:370-396) away from a first target that constrains a SUBSET of a plate's
    elements, `C["ys", idx, "y"].set(v)`.  `filter_to_unconstrained` (sp.py:89-91) = `filter(~constraint.get_selection())`;
    the selection of an `Indexed` layer selects nothing below it (`ChmSel.get_subselection` -> `Indexed.get_inner_map(static)`
    = empty, choice_map.py:658-663, 1494-1496), so the WHOLE site stays among the latents; merged under the new target's own
    subset constraint (`Or`, first operand wins per element: choice_map.py:1714-1717, 1740-1743; Mask.__or__,
    functional_types.py:309-319) the listed elements take the new values and every other element keeps its old one.
    Checked three ways: product == oracle bit for bit; the elements are the ones the rule names; the weights equal
    `new_w - old_score + old_lw` recomputed in f64 from the particles themselves (scipy-free closed form)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Target, numpy as jnp
    from genjax_amd.inference.smc import ChangeTarget, ImportanceK

    def mk(g, f32):
        @g.gen
        def elem(m, x):
            return g.normal(m + x, f32(1.5)) @ "y"

        @g.gen
        def model(a, xs):
            mu = g.normal(a, f32(1.0)) @ "mu"
            g.Vmap(elem, in_axes=(None, 0))(mu, xs) @ "ys"
            return mu
        return model
    m, om = mk(G, float), mk(O, np.float32)
    xs = np.linspace(0.0, 1.0, n).astype(np.float32)
    idx1, v1 = np.array([1, 3]), np.array([0.5, -0.5], np.float32)
    idx2, v2 = np.array([0, 3]), np.array([1.5, -1.5], np.float32)
    t1 = Target(m, (0.25, jnp.array(xs)), C["ys", idx1, "y"].set(v1))
    t2 = Target(m, (0.25, jnp.array(xs)), C["ys", idx2, "y"].set(v2))
    ot1 = O.Target(om, (np.float32(0.25), xs), O.C.d({("ys", "y"): O.indexed(v1, idx1, n)}))
    ot2 = O.Target(om, (np.float32(0.25), xs), O.C.d({("ys", "y"): O.indexed(v2, idx2, n)}))
    first, ofirst = ImportanceK(t1, k_particles=k), O.ImportanceK(ot1, k)
    c1 = first.run_smc(G.key(seed))
    # the latents: the whole plate (and mu) — nothing of the site is dropped
    lat = t1.filter_to_unconstrained(c1.get_particles().get_choices())
    assert tuple(lat["ys", "y"].shape) == (k, n) and "mu" in lat
    olat = ot1.filter_to_unconstrained(ofirst.run_smc(O.key(seed)).get_particles().get_choices())
    assert np.array_equal(lat["ys", "y"].cpu().numpy(), olat[("ys", "y")])
    ct, oct_ = ChangeTarget(first, t2).run_smc(G.key(seed)), O.ChangeTarget(ofirst, ot2).run_smc(O.key(seed))
    lw, ys, mu = (ct.get_log_weights().cpu().numpy(), ct.get_particles().get_choices()["ys", "y"].cpu().numpy(),
                  ct.get_particles().get_choices()["mu"].cpu().numpy())
    assert np.array_equal(lw, oct_.get_log_weights())
    assert np.array_equal(ys, oct_.get_particles().get_choices()[("ys", "y")])
    assert np.array_equal(ct.get_particles().get_score().cpu().numpy(), oct_.get_particles().get_score())
    old = c1.get_particles().get_choices()["ys", "y"].cpu().numpy()
    want = old.copy()
    want[:, idx2] = v2
    assert np.array_equal(ys, want) and np.array_equal(old[:, idx1], np.broadcast_to(v1, (k, 2)))
    assert np.array_equal(mu, c1.get_particles().get_choices()["mu"].cpu().numpy())
    # weights in f64: every element and mu are constrained under the new target (the latents carry them all)
    lp = lambda x, loc, sd: -0.5 * ((x - loc) / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)
    mu64 = mu.astype(np.float64)
    new_w = lp(mu64, 0.25, 1.0) + lp(ys.astype(np.float64), mu64[:, None] + xs[None, :], 1.5).sum(-1)
    old_score = lp(mu64, 0.25, 1.0) + lp(old.astype(np.float64), mu64[:, None] + xs[None, :], 1.5).sum(-1)
    old_lw = lp(old[:, idx1].astype(np.float64), mu64[:, None] + xs[None, idx1], 1.5).sum(-1)
    assert np.allclose(lw, new_w - old_score + old_lw, rtol=0, atol=2e-5)
    assert np.allclose(c1.get_log_weights().cpu().numpy(), old_lw, rtol=0, atol=2e-5)
    # a first target whose index is known only at RUN time (one per particle): the site stays among the latents too
    dyn = torch.full((k,), 2, dtype=torch.int32, device=c1.get_log_weights().device)
    t1d = Target(m, (0.25, jnp.array(xs)), C["ys", G.dynamic_index(dyn), "y"].set(0.75)) if hasattr(G, "dynamic_index") else None
    if t1d is not None:
        cd = ImportanceK(t1d, k_particles=k).run_smc(G.key(seed + 1))
        latd = t1d.filter_to_unconstrained(cd.get_particles().get_choices())
        assert tuple(latd["ys", "y"].shape) == (k, n)


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def check_many_sites(ns=40, B=9, seed=3, kinds=("normal", "uniform", "flip", "normal", "beta")):
    """A5 / A6 beyond ONE launch: a static model of `ns` sites (ref static.py:254-380 walks any number of them) needs
    2 * ns stored leaves — more than the launch ABI's 64 from 32 sites on — and is cut into a chain of launches
    (program.split_graph): simulate / importance / assess / update (new constraints + a changed argument) / regenerate
    / a StaticRequest of Rejuvenate moves, all against the oracle bit for bit (an old uniform value that an edit upstream
    leaves outside its new support scores -inf, and -inf - -inf = NaN on both sides: compared as equal).  Site i reads site i - 1 (a chain of
    dependencies crossing every cut) and sites 0 and ns // 2 (long-lived values: spilled once, reloaded where read)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S, StaticRequest, Update, numpy as jnp

    def mk(g, lit, where):
        @g.gen
        def model(a, b):
            x, first, mid = a, a, a
            for i in range(ns):
                kind = kinds[i % len(kinds)]
                m = x * lit(0.5) + first * lit(0.125) - mid * lit(0.25) + b
                if kind == "normal":
                    x = g.normal(m, lit(1.25)) @ f"s{i}"
                elif kind == "uniform":
                    x = g.uniform(m - lit(6.0), m + lit(7.0)) @ f"s{i}"      # (wide: an edit upstream keeps the old value inside)
                elif kind == "flip":
                    f = g.flip(lit(0.4)) @ f"s{i}"
                    x = where(f, m, x)
                else:
                    p_ = g.beta(lit(2.0), lit(3.0)) @ f"s{i}"
                    x = p_ * lit(2.0) + m * lit(0.5)
                if i == 0:
                    first = x
                if i == ns // 2:
                    mid = x
            return x + first
        return model
    m = mk(G, float, jnp.where)
    om = mk(O, np.float32, lambda c, x, y: np.where(c, x, y).astype(np.float32))
    rng = np.random.default_rng(seed)
    dev = G._lib.get().device
    b = rng.normal(size=B).astype(np.float32) * np.float32(0.1)
    args, oargs = (0.3, torch.from_numpy(b).to(dev)), (np.float32(0.3), b)
    keys, okeys = G.split(G.key(seed), B), O.split(O.key(seed), B)
    names = [f"s{i}" for i in range(ns)]
    cont = [nm for i, nm in enumerate(names) if kinds[i % len(kinds)] == "normal"]

    def same(tr, otr, what):
        ch, och = tr.get_choices(), otr.get_choices()
        for nm in names:
            assert np.array_equal(_np(ch[nm]), np.broadcast_to(och[nm], _np(ch[nm]).shape)), (what, nm)
        assert np.array_equal(_np(tr.get_score()), otr.get_score(), equal_nan=True), (what, "score")
        assert np.array_equal(*np.broadcast_arrays(_np(tr.get_retval()), otr.get_retval())), (what, "retval")   # (launch-uniform: a scalar)
    tr, otr = m.simulate(keys, args), om.simulate(okeys, oargs)
    same(tr, otr, "simulate")
    # importance: a third of the normal sites constrained (launch-uniform values)
    obs = {nm: np.float32(rng.normal()) for nm in cont[::3]}
    tri, w = m.importance(keys, C.d({k_: float(v) for k_, v in obs.items()}), args)
    otri, ow = om.importance(okeys, O.C.d(obs), oargs)
    assert np.array_equal(_np(w), ow), "importance weight"
    same(tri, otri, "importance")
    # assess of all the choices
    sc, _ = m.assess(tr.get_choices(), args)
    osc, _ = om.assess(otr.get_choices(), oargs, batch_shape=(B,))
    assert np.array_equal(_np(sc), osc) and np.array_equal(_np(sc), _np(tr.get_score())), "assess"
    # update: other constraints (one value per particle) and a changed first argument
    upd = {nm: rng.normal(size=B).astype(np.float32) for nm in cont[1::4]}
    k2, ok2 = G.split(G.key(seed + 1), B), O.split(O.key(seed + 1), B)
    new, wu, _, bwd = Update(C.d({k_: torch.from_numpy(v).to(dev) for k_, v in upd.items()})).edit(
        k2, tri, (Diff(0.7, G.UnknownChange), Diff.no_change(args[1])))
    onew, owu, odisc = om.update(ok2, otri, O.C.d(upd), (np.float32(0.7), b))
    assert np.array_equal(_np(wu), owu, equal_nan=True), "update weight"
    same(new, onew, "update")
    for nm in upd:
        assert np.array_equal(np.broadcast_to(_np(bwd.constraint[nm]), (B,)), np.broadcast_to(odisc[nm], (B,))), ("discard", nm)
    # regenerate a selection spread over the chain
    pick = names[2::7]
    sel = S[pick[0]]
    for nm in pick[1:]:
        sel = sel | S[nm]
    k3, ok3 = G.split(G.key(seed + 2), B), O.split(O.key(seed + 2), B)
    rg, wr, _, _ = Regenerate(sel).edit(k3, new, Diff.no_change((0.7, args[1])))
    org, owr, _ = om.regenerate(ok3, onew, O.selection(*pick), (np.float32(0.7), b))
    assert np.array_equal(_np(wr), owr, equal_nan=True), "regenerate weight"
    same(rg, org, "regenerate")
    # a StaticRequest of Rejuvenate moves on a few normal sites
    mv = cont[::5]
    req = StaticRequest({nm: G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5)) for nm in mv})
    oreq = {nm: O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.5))) for nm in mv}
    k4, ok4 = G.split(G.key(seed + 3), B), O.split(O.key(seed + 3), B)
    rj, wj, _, _ = req.edit(k4, rg, Diff.no_change((0.7, args[1])))
    orj, owj = om.edit_static(ok4, org, oreq, (np.float32(0.7), b))
    assert np.array_equal(_np(wj), owj, equal_nan=True), "static request weight"
    same(rj, orj, "static request")


def check_three_nested_plates(dims=(20, 20, 20), B=5, seed=1):
    """A15 three combinator levels deep (ref vmap.py:180-218 nests freely): `vmap(vmap(vmap(elem)))` over
    dims[0] x dims[1] x dims[2] elements — three counted loops in the site program (GMX_F_FLAT: the row-major index
    over all three) — simulate / importance (launch-uniform and per-particle observations) / assess / update under a
    changed argument, against the oracle bit for bit."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, numpy as jnp

    def mk(g, lit):
        @g.gen
        def elem(m, x):
            return g.normal(m + x, lit(1.5)) @ "y"
        return elem
    e, oe = mk(G, float), mk(O, np.float32)
    A_, B_, C_ = dims
    rng = np.random.default_rng(seed)
    xs = np.linspace(0, 1, A_ * B_ * C_).astype(np.float32).reshape(dims)
    ax = (None, 0)
    v3 = G.Vmap(G.Vmap(G.Vmap(e, in_axes=ax), in_axes=ax), in_axes=ax)
    ov3 = O.Vmap(O.Vmap(O.Vmap(oe, in_axes=ax), in_axes=ax), in_axes=ax)
    dev = G._lib.get().device
    args, oargs = (0.3, jnp.array(xs)), (np.float32(0.3), xs)
    tr, otr = v3.simulate(G.split(G.key(seed), B), args), ov3.simulate(O.split(O.key(seed), B), oargs)
    assert tuple(tr.get_choices()["y"].shape) == (B,) + tuple(dims)
    assert np.array_equal(_np(tr.get_choices()["y"]), otr.get_choices()["y"]), "simulate values"
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), "simulate score"
    ys = rng.normal(size=dims).astype(np.float32)
    tri, w = v3.importance(G.split(G.key(seed + 1), B), C["y"].set(jnp.array(ys)), args)
    otri, ow = ov3.importance(O.split(O.key(seed + 1), B), O.C.d({"y": ys}), oargs)
    assert np.array_equal(_np(w), ow) and np.array_equal(_np(tri.get_score()), otri.get_score()), "importance"
    ypp = rng.normal(size=(B,) + tuple(dims)).astype(np.float32)
    trp, wp = v3.importance(G.split(G.key(seed + 2), B), C["y"].set(torch.from_numpy(ypp).to(dev)), args)
    otrp, owp = ov3.importance(O.split(O.key(seed + 2), B), O.C.d({"y": ypp}), oargs)
    assert np.array_equal(_np(wp), owp), "importance, one observation table per particle"
    sc, _ = v3.assess(tr.get_choices(), args)
    osc, _ = ov3.assess(otr.get_choices(), oargs, batch_shape=(B,))
    assert np.array_equal(_np(sc), osc) and np.array_equal(_np(sc), _np(tr.get_score())), "assess"
    # update: the shared argument changes, every element re-scored; then new observations everywhere
    new, wu, _, _ = Update(C.n()).edit(G.split(G.key(seed + 3), B), tr, (Diff(0.5, G.UnknownChange), Diff.no_change(args[1])))
    sc2, _ = v3.assess(tr.get_choices(), (0.5, args[1]))
    assert np.array_equal(_np(new.get_score()), _np(sc2)), "update: score under the changed argument"
    lp = lambda x, loc: -0.5 * ((x - loc) / 1.5) ** 2 - np.log(1.5) - 0.5 * np.log(2 * np.pi)
    y64 = _np(tr.get_choices()["y"]).astype(np.float64)
    want = (lp(y64, 0.5 + xs) - lp(y64, 0.3 + xs)).reshape(B, -1).sum(-1)
    assert np.allclose(_np(wu), want, rtol=0, atol=2e-2 * max(1.0, A_ * B_ * C_ / 8000)), "update weight vs f64"
    assert np.array_equal(_np(new.get_choices()["y"]), _np(tr.get_choices()["y"]))


def check_long_vector_sites(n=500, K=64, seed=1):
    """A2 / A3 / A8: a vector-valued site of MANY elements under a particle batch (TFP batch semantics,
    tensorflow_probability/__init__.py:52-62; the score summed by distribution.py:383-396) — the regression
    `y ~ normal(a * xs + b, sigma)` with n observations under ImportanceK — lowered to ONE counted loop per particle
    (static._vector_site_loop, tracer.LazyVec) instead of n unrolled copies: element j draws with counter j from the one
    site key, the score is the element-order sum.  simulate / importance (launch-uniform and per-particle observations)
    / ImportanceK / assess / update (a new value upstream: every element re-scored; new observations) against the
    oracle bit for bit, a flip-valued and a uniform-valued long site included."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, numpy as jnp
    from genjax_amd.inference.smc import ImportanceK

    def mk(g, lit):
        prod = g is G

        @g.gen
        def reg(xs):
            a = g.normal(lit(0.0), lit(2.0)) @ "a"
            b = g.normal(lit(0.0), lit(2.0)) @ "b"
            mean = (a * xs + b) if prod else (a[..., None] * xs + b[..., None]).astype(np.float32)
            g.normal(mean, lit(0.5)) @ "y"
            p = (jnp.sigmoid(mean * 0.25) if prod else O.sigmoid((mean * np.float32(0.25)).astype(np.float32)))
            g.flip(p) @ "f"
            g.uniform(mean - lit(4.0), (mean * mean if prod else (mean * mean).astype(np.float32)) + lit(5.0)) @ "u"
            return a
        return reg
    m, om = mk(G, float), mk(O, np.float32)
    dev = G._lib.get().device
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1, 1, n).astype(np.float32)
    ys = (0.7 * xs - 0.2 + 0.5 * rng.normal(size=n)).astype(np.float32)
    args, oargs = (jnp.array(xs),), (xs,)
    keys, okeys = G.split(G.key(seed), K), O.split(O.key(seed), K)
    tr, otr = m.simulate(keys, args), om.simulate(okeys, oargs)
    for nm in ("a", "b", "y", "f", "u"):
        assert np.array_equal(_np(tr.get_choices()[nm]), otr.get_choices()[nm]), ("simulate", nm)
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), "simulate score"
    coll = ImportanceK(G.Target(m, args, C["y"].set(jnp.array(ys))), k_particles=K).run_smc(G.key(seed + 1))
    ocoll = O.ImportanceK(O.Target(om, oargs, O.C.d({"y": ys})), K).run_smc(O.key(seed + 1))
    assert np.array_equal(_np(coll.get_log_weights()), ocoll.get_log_weights()), "ImportanceK log weights"
    assert np.array_equal(_np(coll.get_particles().get_choices()["u"]), ocoll.get_particles().get_choices()["u"])
    ypp = (ys[None, :] + 0.1 * rng.normal(size=(K, n))).astype(np.float32)
    tri, w = m.importance(keys, C["y"].set(torch.from_numpy(ypp).to(dev)), args)
    otri, ow = om.importance(okeys, O.C.d({"y": ypp}), oargs)
    assert np.array_equal(_np(w), ow) and np.array_equal(_np(tri.get_score()), otri.get_score()), "importance, per particle"
    sc, _ = m.assess(tr.get_choices(), args)
    osc, _ = om.assess(otr.get_choices(), oargs, batch_shape=(K,))
    assert np.array_equal(_np(sc), osc) and np.array_equal(_np(sc), _np(tr.get_score())), "assess"
    # update: a new slope (one per particle) — every element of the three long sites re-scored — and new observations
    a_new = rng.normal(size=K).astype(np.float32)
    k2, ok2 = G.split(G.key(seed + 2), K), O.split(O.key(seed + 2), K)
    new, wu, _, bwd = Update(C["a"].set(torch.from_numpy(a_new).to(dev)) | C["y"].set(jnp.array(ys))).edit(
        k2, tr, Diff.no_change(args))
    onew, owu, odisc = om.update(ok2, otr, O.C.d({"a": a_new, "y": ys}), oargs)
    assert np.array_equal(_np(wu), owu, equal_nan=True), "update weight"
    assert np.array_equal(_np(new.get_score()), onew.get_score(), equal_nan=True), "update score"
    assert np.array_equal(_np(bwd.constraint["y"]), odisc["y"]), "update discard"


def check_hierarchical_vector_latent(J=40, K=16, seed=2):
    """A2 / A5 / A8: the 8-schools shape at J schools (BASELINE config 4 has J = 8) — a LATENT vector site whose values
    are the parameters of the next vector site: `theta ~ normal(mu 1_J, tau 1_J); y ~ normal(theta, sigma_J)`.  The model
    computes with the values of a long vector site, which the counted-loop lowering keeps in memory only: the call is
    traced again with such sites unrolled (static._unrolled_when_values_are_used) and, past the slots of one launch,
    cut into a chain of launches (program.split_graph: J = 40 found a cut that put 70 values in flight into one
    segment).  simulate / ImportanceK / importance with per-particle theta / assess / update of mu and of half of theta /
    regenerate of theta against the oracle bit for bit."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    from genjax_amd.inference.smc import ImportanceK
    sig = np.linspace(9, 18, J).astype(np.float32)
    ys = np.linspace(-3, 28, J).astype(np.float32)

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        log_tau = G.normal(0.0, 1.0) @ "log_tau"
        theta = G.normal(mu * jnp.ones(J), jnp.exp(log_tau) * jnp.ones(J)) @ "theta"
        G.normal(theta, jnp.array(sig)) @ "y"
        return mu

    @O.gen
    def oschools():
        mu = O.normal(0.0, 5.0) @ "mu"
        log_tau = O.normal(0.0, 1.0) @ "log_tau"
        theta = O.normal(mu[..., None] * np.ones(J, np.float32), O.exp(log_tau)[..., None] * np.ones(J, np.float32)) @ "theta"
        O.normal(theta, sig) @ "y"
        return mu
    dev = G._lib.get().device
    rng = np.random.default_rng(seed)
    coll = ImportanceK(G.Target(schools, (), C["y"].set(ys)), k_particles=K).run_smc(G.key(seed))
    ocoll = O.ImportanceK(O.Target(oschools, (), O.C.d({"y": ys})), K).run_smc(O.key(seed))
    assert np.array_equal(_np(coll.get_log_weights()), ocoll.get_log_weights()), "ImportanceK log weights"
    assert np.array_equal(_np(coll.get_particles().get_choices()["theta"]), ocoll.get_particles().get_choices()["theta"])
    keys, okeys = G.split(G.key(seed + 1), K), O.split(O.key(seed + 1), K)
    tr, otr = schools.simulate(keys, ()), oschools.simulate(okeys, ())
    for nm in ("mu", "log_tau", "theta", "y"):
        assert np.array_equal(_np(tr.get_choices()[nm]), otr.get_choices()[nm]), ("simulate", nm)
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), "simulate score"
    th = rng.normal(size=(K, J)).astype(np.float32)
    tri, w = schools.importance(keys, C["theta"].set(torch.from_numpy(th).to(dev)) | C["y"].set(ys), ())
    otri, ow = oschools.importance(okeys, O.C.d({"theta": th, "y": ys}), ())
    assert np.array_equal(_np(w), ow) and np.array_equal(_np(tri.get_score()), otri.get_score()), "importance, theta given"
    sc, _ = schools.assess(tr.get_choices(), ())
    osc, _ = oschools.assess(otr.get_choices(), (), batch_shape=(K,))
    assert np.array_equal(_np(sc), osc), "assess"
    mu_new = rng.normal(size=K).astype(np.float32)
    k2, ok2 = G.split(G.key(seed + 2), K), O.split(O.key(seed + 2), K)
    new, wu, _, bwd = Update(C["mu"].set(torch.from_numpy(mu_new).to(dev))).edit(k2, tr, Diff.no_change(()))
    onew, owu, odisc = oschools.update(ok2, otr, O.C.d({"mu": mu_new}), ())
    assert np.array_equal(_np(wu), owu, equal_nan=True) and np.array_equal(_np(new.get_score()), onew.get_score(), equal_nan=True), "update mu"
    new, wr, _, _ = Regenerate(S["theta"]).edit(k2, tr, Diff.no_change(()))
    onew, owr = oschools.regenerate(ok2, otr, O.selection("theta"), ())[:2]
    assert np.array_equal(_np(wr), owr, equal_nan=True), "regenerate theta: weight"
    assert np.array_equal(_np(new.get_choices()["theta"]), onew.get_choices()["theta"]), "regenerate theta: values"


def time_vector_site_vs_plate(n=500, K=100_000, reps=5):
    """seconds per ImportanceK.run_smc of `y ~ normal(a * xs + b, 0.5)` over n observations and K particles: as ONE
    vector-valued site (a counted loop per particle) and as a `vmap` plate (different key tree, same work)"""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    from genjax_amd.inference.smc import ImportanceK
    dev = G._lib.get().device
    rng = np.random.default_rng(0)
    xs = np.linspace(-1, 1, n).astype(np.float32)
    ys = (0.7 * xs - 0.2 + 0.5 * rng.normal(size=n)).astype(np.float32)
    args = (jnp.array(xs),)

    @G.gen
    def point(a, b, x):
        return G.normal(a * x + b, 0.5) @ "y"

    @G.gen
    def reg_plate(xs_):
        a = G.normal(0.0, 2.0) @ "a"
        b = G.normal(0.0, 2.0) @ "b"
        point.vmap(in_axes=(None, None, 0))(a, b, xs_) @ "ys"
        return a

    @G.gen
    def reg_vec(xs_):
        a = G.normal(0.0, 2.0) @ "a"
        b = G.normal(0.0, 2.0) @ "b"
        G.normal(a * xs_ + b, 0.5) @ "y"
        return a
    out = {}
    for name, model, con in (("vector_site", reg_vec, C["y"].set(jnp.array(ys))), ("vmap_plate", reg_plate, C["ys", :, "y"].set(jnp.array(ys)))):
        alg = ImportanceK(G.Target(model, args, con), k_particles=K)
        lw = alg.run_smc(G.key(3)).get_log_weights()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            lw = alg.run_smc(G.key(4 + r)).get_log_weights()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps
    return out


def check_scan_index_request_o1(n=64, T=40, seed=14, edits=7, timing=False):
    """F1 / VERDICT r4 item 9: `IndexRequest(t, sub)` on a LONG scan held per particle (scan.py:325-416 `edit_index`) as
    the reference states it — slice t edited, slice t + 1 visited by an empty Update against the changed carry, nothing
    else touched — in O(1) steps (combinators._scan_edit_index_o1): chains of edits (Update of the observation, Update
    of the state, Regenerate; first, middle and last step) against the oracle's slice / edit / write-back — values,
    weights, the IN-ORDER score, the return value — and against the counted-loop form of the same edit.  A kernel whose
    carry runs through arithmetic (a running sum) is refused by the static check and takes the loop form."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    from genjax_amd import combinators as cmb
    from genjax_amd.engine import Patched
    dev = G._lib.get().device

    def mk(g, lit):
        @g.gen
        def step(c, x):
            z = g.normal(c * lit(0.5) + x, lit(1.25)) @ "z"
            g.normal(z, lit(0.75)) @ "y"
            return z, z * lit(2.0)
        return step
    step, ostep = mk(G, float), mk(O, np.float32)
    xs = np.linspace(-0.5, 0.5, T).astype(np.float32)
    sc, osc = G.Scan(step, T), O.Scan(ostep, T)
    c0 = np.random.default_rng(seed).normal(size=n).astype(np.float32)
    args, oargs = (torch.from_numpy(c0).to(dev), jnp.array(xs)), (c0, xs)
    tr, otr = sc.simulate(G.split(G.key(seed), n), args), osc.simulate(O.split(O.key(seed), n), oargs)
    rng = np.random.default_rng(seed + 1)
    saw_lazy = False
    for e in range(edits):
        t = [0, T - 1, int(rng.integers(1, T - 1)), int(rng.integers(1, T - 1))][e % 4]
        k, ok = G.split(G.key(seed + 50 + e), n), O.split(O.key(seed + 50 + e), n)
        kind = e % 3
        if kind == 0:
            v = np.float32(rng.normal())
            sub, osub = Update(C["y"].set(float(v))), (lambda kk, sl, a: ostep.update(kk, sl, O.C.d({("y",): v}), a)[:2])
        elif kind == 1:
            vv = rng.normal(size=n).astype(np.float32)
            sub = Update(C["z"].set(torch.from_numpy(vv).to(dev)))
            osub = (lambda kk, sl, a, vv=vv: ostep.update(kk, sl, O.C.d({("z",): vv}), a)[:2])
        else:
            sub, osub = Regenerate(S["z"]), (lambda kk, sl, a: ostep.regenerate(kk, sl, O.selection("z"), a)[:2])
        per = e % 5 == 4                  # ONE STEP INDEX PER PARTICLE (a traced idx under the particle vmap; round 6)
        if per:
            t_host = rng.integers(0, T, n).astype(np.int32)
            t_host[:3] = (0, T - 1, T - 2)
            t = torch.from_numpy(t_host).to(dev)
        new, w, _, bwd = IndexRequest(t, sub).edit(k, tr, Diff.no_change(args))
        if per:
            onew, ow = O.scan_edit_index_per_particle(osc, ok, otr, oargs, t_host, osub)
        else:
            onew, ow = O.scan_edit_index(osc, ok, otr, oargs, t, osub)
        saw_lazy = saw_lazy or isinstance(new.inner.subtraces["z"].value, Patched)
        assert isinstance(new.inner.subtraces["z"].value, Patched) or not per, "per-particle index: the O(1) form was not taken"
        assert isinstance(bwd, IndexRequest) and (per or bwd.idx == t)
        assert np.array_equal(_np(w), ow), (e, "weight")
        for nm in ("z", "y"):
            assert np.array_equal(_np(new.get_choices()[nm]), np.broadcast_to(onew.get_choices()[nm], (n, T))), (e, nm)
        assert np.array_equal(_np(new.get_score()), onew.get_score()), (e, "score")
        rc, ry = new.get_retval()
        orc, ory = onew.get_retval()
        assert np.array_equal(_np(rc), orc) and np.array_equal(_np(ry), ory), (e, "retval")
        # the counted-loop form of the same edit (what this replaces): same trace, same weight
        old_refused = sc.__dict__.get("_o1_refused")
        sc.__dict__["_o1_refused"] = True
        try:
            loop, wl, _, _ = IndexRequest(t, sub).edit(k, tr, Diff.no_change(args))
        finally:
            if old_refused is None:
                sc.__dict__.pop("_o1_refused")
        assert np.array_equal(_np(wl), _np(w)) and np.array_equal(_np(loop.get_score()), _np(new.get_score())), (e, "loop form")
        assert np.array_equal(_np(loop.get_choices()["z"]), _np(new.get_choices()["z"]))
        tr, otr = new, onew
    assert saw_lazy, "the O(1) form was not taken"

    # a kernel whose carry runs through arithmetic: refused statically, the loop form answers
    @G.gen
    def running(c, x):
        z = G.normal(x, 1.0) @ "z"
        return c + z, z
    sc2 = G.Scan(running, T)
    tr2 = sc2.simulate(G.split(G.key(seed + 9), n), (torch.zeros(n, device=dev), jnp.array(xs)))
    new2, w2, _, _ = IndexRequest(3, Update(C["z"].set(0.5))).edit(G.split(G.key(seed + 10), n), tr2,
                                                                   Diff.no_change((torch.zeros(n, device=dev), jnp.array(xs))))
    assert sc2.__dict__.get("_o1_refused") and not isinstance(new2.inner.subtraces["z"].value, Patched)
    z_old, z_new = _np(tr2.get_choices()["z"]), _np(new2.get_choices()["z"])
    assert np.all(z_new[:, 3] == np.float32(0.5)) and np.array_equal(np.delete(z_old, 3, 1), np.delete(z_new, 3, 1))
    if not timing:
        return None
    out = {}
    for name, refuse in (("o1", False), ("loop", True)):
        if refuse:
            sc.__dict__["_o1_refused"] = True
        else:
            sc.__dict__.pop("_o1_refused", None)
        k = G.split(G.key(seed + 99), n)
        req = IndexRequest(T // 2, Update(C["y"].set(0.25)))
        req.edit(k, tr, Diff.no_change(args))
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
            _ = _np(w[:1])
        if dev.type == "cuda":
            torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps
    sc.__dict__.pop("_o1_refused", None)
    return out


def check_plate_of_scans_index_request_o1(n=33, J=20, T=40, seed=23, edits=8, timing=False):
    """`IndexRequest(j, IndexRequest(t, sub))` on a plate of LONG scans written directly
    (`kernel.scan(n=T).vmap()`: leaves [n, J, T]) edits ONE step of ONE element in O(1) steps — the plate's slice /
    write-back (vmap.py:277-332) around the scan's (scan.py:325-416) — with Python-int and per-particle indices at either
    level, chains of edits; bit-exact against the oracle (every element edited with the caller's key, kept at j only:
    `vmap_edit_index_batched` around `scan_edit_index[_per_particle]`) and against the counted-loop form of the same edit.
    Before round 6 a directly nested `scan.vmap()` could not be edited at all (its trace is held flat)."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    from genjax_amd.engine import Patched
    dev = G._lib.get().device

    def mk(g, lit):
        @g.gen
        def step(c, x):
            z = g.normal(c * lit(0.5) + x, lit(1.25)) @ "z"
            g.normal(z, lit(0.75)) @ "y"
            return z, z * lit(2.0)
        return step
    step, ostep = mk(G, float), mk(O, np.float32)
    xs = np.linspace(-0.5, 0.5, T).astype(np.float32)
    c0 = np.linspace(0.0, 1.0, J).astype(np.float32)
    sc, osc = G.Scan(step, T), O.Scan(ostep, T)
    pl, opl = sc.vmap(in_axes=(0, None)), O.Vmap(osc, in_axes=(0, None))
    args, oargs = (jnp.array(c0), jnp.array(xs)), (c0, xs)
    tr, otr = pl.simulate(G.split(G.key(seed), n), args), opl.simulate(O.split(O.key(seed), n), oargs)
    assert np.array_equal(_np(tr.get_choices()["z"]), otr.get_choices()["z"])
    # the whole-trace edits of the flat nest (Update of one address, of a whole row)
    upd, w_u, _, _ = Update(C[1, 2, "y"].set(0.25)).edit(G.split(G.key(seed + 1), n), tr, Diff.no_change(args))
    y_new = _np(upd.get_choices()["y"])
    assert np.all(y_new[:, 1, 2] == np.float32(0.25))
    from scipy import stats
    lp = stats.norm.logpdf
    z_ = _np(tr.get_choices()["z"]).astype(np.float64)
    want = lp(0.25, z_[:, 1, 2], 0.75) - lp(_np(tr.get_choices()["y"]).astype(np.float64)[:, 1, 2], z_[:, 1, 2], 0.75)
    assert np.allclose(_np(w_u), want, rtol=2e-5, atol=2e-4)
    rng = np.random.default_rng(seed + 2)
    saw_lazy = 0
    for e in range(edits):
        k, ok = G.split(G.key(seed + 50 + e), n), O.split(O.key(seed + 50 + e), n)
        kind = e % 3
        if kind == 0:
            v = np.float32(rng.normal())
            sub, osub = Update(C["y"].set(float(v))), (lambda kk, sl, a, v=v: ostep.update(kk, sl, O.C.d({("y",): v}), a)[:2])
        elif kind == 1:
            vv = rng.normal(size=n).astype(np.float32)
            sub = Update(C["z"].set(torch.from_numpy(vv).to(dev)))
            osub = (lambda kk, sl, a, vv=vv: ostep.update(kk, sl, O.C.d({("z",): np.broadcast_to(vv[:, None], (n, J))}), a)[:2])
        else:
            sub, osub = Regenerate(S["z"]), (lambda kk, sl, a: ostep.regenerate(kk, sl, O.selection("z"), a)[:2])
        j_per, t_per = e % 4 == 2, e % 4 in (1, 2)
        j_host = rng.integers(0, J, n).astype(np.int32) if j_per else int(rng.integers(0, J))
        t_host = rng.integers(0, T, n).astype(np.int32) if t_per else [0, T - 1, int(rng.integers(1, T - 1))][e % 3]
        if t_per:
            t_host[:3] = (0, T - 1, T - 2)
        gj = torch.from_numpy(j_host).to(dev) if j_per else j_host
        gt = torch.from_numpy(t_host).to(dev) if t_per else t_host
        req = IndexRequest(gj, IndexRequest(gt, sub))
        new, w, _, bwd = req.edit(k, tr, Diff.no_change(args))
        if t_per:
            t_all = np.broadcast_to(t_host[:, None], (n, J))
            edit_all = lambda kb, inner, a: O.scan_edit_index_per_particle(osc, kb, inner, a, t_all, osub)
        else:
            edit_all = lambda kb, inner, a: O.scan_edit_index(osc, kb, inner, a, t_host, osub)
        onew, ow = O.vmap_edit_index_batched(opl, ok, otr, j_host, edit_all, oargs)
        lazy = isinstance(new.inner.subtraces["z"].value, Patched)
        saw_lazy += lazy
        assert lazy, (e, "the O(1) form was not taken")
        assert isinstance(bwd, IndexRequest) and isinstance(bwd.request, IndexRequest)
        assert np.array_equal(_np(w), ow), (e, "weight", np.abs(_np(w) - ow).max())
        for nm in ("z", "y"):
            assert np.array_equal(_np(new.get_choices()[nm]), onew.get_choices()[nm]), (e, nm)
        assert np.array_equal(_np(new.get_score()), onew.get_score()), (e, "score")
        rc, ry = new.get_retval()
        orc, ory = onew.get_retval()
        assert np.array_equal(_np(rc), orc) and np.array_equal(_np(ry), ory), (e, "retval")
        if e < 4:        # the counted-loop form of the same edit: same trace, same weight
            sc.__dict__["_o1_refused"] = True
            try:
                loop, wl, _, _ = req.edit(k, tr, Diff.no_change(args))
            finally:
                sc.__dict__.pop("_o1_refused")
            assert not isinstance(loop.inner.subtraces["z"].value, Patched)
            assert np.array_equal(_np(wl), _np(w)) and np.array_equal(_np(loop.get_score()), _np(new.get_score())), (e, "loop form")
            assert np.array_equal(_np(loop.get_choices()["z"]), _np(new.get_choices()["z"]))
        tr, otr = new, onew
    if not timing:
        return None
    out = {}
    for name, refuse in (("o1", False), ("loop", True)):
        if refuse:
            sc.__dict__["_o1_refused"] = True
        else:
            sc.__dict__.pop("_o1_refused", None)
        k = G.split(G.key(seed + 99), n)
        req = IndexRequest(J // 2, IndexRequest(T // 2, Update(C["y"].set(0.25))))
        req.edit(k, tr, Diff.no_change(args))
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            new, w, _, _ = req.edit(k, tr, Diff.no_change(args))
            _ = _np(w[:1])
        if dev.type == "cuda":
            torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps
    sc.__dict__.pop("_o1_refused", None)
    return out


def check_scan_of_plates_index_request_o1(n=9, T=40, P=30, seed=3, edits=9):
    """`IndexRequest(t, sub)` on a LONG scan whose step holds a PLATE (`leaf.vmap()(x, sigmas) @ "obs"` inside the kernel:
    leaves [n, T, P]) edits step t — and visits step t + 1 — in O(1) steps as well (round 6: the slices of nested traces are
    taken and patched back whole, [n, P] rows of the [n, T, P] leaves): chains of edits, each applied to the O(1) form's
    trace AND to the counted-loop form's, which must stay equal bit for bit — Update of the state, of one plate element
    (as an address and as a nested IndexRequest through a StaticRequest), Regenerate of the state; Python-int and
    per-particle step indices.  (The oracle's scan.edit_index slices trailing axes only: the loop form, which the oracle
    and scipy hold elsewhere — check_nested_index_edits — is the reference here.)"""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, StaticRequest, Update, numpy as jnp
    from genjax_amd.engine import Patched
    dev = G._lib.get().device

    @G.gen
    def leaf(m, s):
        return G.normal(m, s) @ "z"

    @G.gen
    def step(x, sx):
        leaf.vmap(in_axes=(None, 0))(x, jnp.array(np.linspace(1, 2, P).astype(np.float32))) @ "obs"
        xn = G.normal(0.9 * x + sx, 0.5) @ "x"
        return xn, xn
    sc = G.Scan(step, T)
    args = (torch.linspace(0, 1, n).to(dev), jnp.array(np.linspace(-0.5, 0.5, T).astype(np.float32)))
    tr_a = tr_b = sc.simulate(G.split(G.key(seed), n), args)
    rng = np.random.default_rng(seed)
    lazy = 0
    for e in range(edits):
        sub = [Update(C["x"].set(float(rng.normal()))),
               StaticRequest({"obs": IndexRequest(int(rng.integers(P)), Update(C["z"].set(float(rng.normal()))))}),
               Update(C["obs", int(rng.integers(P)), "z"].set(torch.from_numpy(rng.normal(size=n).astype(np.float32)).to(dev))),
               StaticRequest({"x": Regenerate(S.all())})][e % 4]
        if e % 3 == 2:
            t_host = rng.integers(0, T, n).astype(np.int32)
            t_host[:2] = (0, T - 1)
            t = torch.from_numpy(t_host).to(dev)
        else:
            t = [0, T - 1, int(rng.integers(1, T - 1))][e % 3]
        k = G.split(G.key(seed + 10 + e), n)
        sc.__dict__.pop("_o1_refused", None)
        new_a, w_a, _, _ = IndexRequest(t, sub).edit(k, tr_a, Diff.no_change(args))
        sc.__dict__["_o1_refused"] = True
        try:
            new_b, w_b, _, _ = IndexRequest(t, sub).edit(k, tr_b, Diff.no_change(args))
        finally:
            sc.__dict__.pop("_o1_refused")
        lazy += isinstance(new_a.inner.subtraces["x"].value, Patched)
        assert not isinstance(new_b.inner.subtraces["x"].value, Patched)
        assert np.array_equal(_np(w_a), _np(w_b), equal_nan=True), (e, "weight")
        assert np.array_equal(_np(new_a.get_score()), _np(new_b.get_score())), (e, "score")
        for ad in (("x",), ("obs", "z")):
            assert np.array_equal(_np(new_a.get_choices()[ad]), _np(new_b.get_choices()[ad])), (e, ad)
        assert np.array_equal(_np(new_a.get_retval()[0]), _np(new_b.get_retval()[0])), (e, "carry")
        assert np.array_equal(_np(new_a.get_retval()[1]), _np(new_b.get_retval()[1])), (e, "ys")
        tr_a, tr_b = new_a, new_b
    assert lazy == edits, "the O(1) form was not taken"


def check_direct_plate_of_scans_random(seed):
    """`kernel.scan(n=T).vmap()` written DIRECTLY (its trace is held flat: the kernel's sites with the [J, T] axes), at random
    sizes around the unroll limits (J in 2 / 5 / 17 / 20, T in 3 / 8 / 17 / 40; 1 / 4 / 9 particles): simulate, importance
    under a full observation, assess, Update of a whole site, an empty Update under a CHANGED mapped argument, Update of
    one (j, t) address — against the oracle's Vmap(Scan), bit for bit.  (Before round 6 `Scan._trace_edit` refused the flat
    previous trace: every edit of such a nest raised.)"""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, numpy as jnp
    f32 = np.float32
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 4, 9])); J = int(rng.choice([2, 5, 17, 20])); T = int(rng.choice([3, 8, 17, 40]))
    sd = f32(rng.uniform(0.5, 2.0))
    def mk(g, lit):
        @g.gen
        def step(c, x):
            z = g.normal(c * lit(0.5) + x, lit(sd)) @ "z"
            g.normal(z, lit(0.75)) @ "y"
            return z, z * lit(2.0)
        return step
    step, ostep = mk(G, float), mk(O, f32)
    xs = rng.normal(size=T).astype(f32); c0 = rng.normal(size=J).astype(f32)
    pl, opl = G.Scan(step, T).vmap(in_axes=(0, None)), O.Vmap(O.Scan(ostep, T), in_axes=(0, None))
    args, oargs = (jnp.array(c0), jnp.array(xs)), (c0, xs)
    k, ok = G.split(G.key(seed), n), O.split(O.key(seed), n)
    tr, otr = pl.simulate(k, args), opl.simulate(ok, oargs)
    for a in ("z", "y"):
        assert np.array_equal(_np(tr.get_choices()[a]), otr.get_choices()[a]), (seed, "simulate", a)
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), (seed, "simulate score")
    yobs = rng.normal(size=(J, T)).astype(f32)
    tr2, w2 = pl.importance(k, C["y"].set(jnp.array(yobs)), args)
    otr2, ow2 = opl.importance(ok, O.C.d({("y",): yobs}), oargs)
    assert np.array_equal(_np(w2), ow2), (seed, "importance w")
    assert np.array_equal(_np(tr2.get_choices()["z"]), otr2.get_choices()["z"]), (seed, "importance z")
    s, _ = pl.assess(tr2.get_choices(), args)
    assert np.array_equal(_np(s), _np(tr2.get_score())), (seed, "assess")
    # update: the whole of z (launch-uniform), then under a changed first argument
    znew = rng.normal(size=(J, T)).astype(f32)
    k2, ok2 = G.split(G.key(seed + 1), n), O.split(O.key(seed + 1), n)
    tr3, w3, _, _ = Update(C["z"].set(jnp.array(znew))).edit(k2, tr2, Diff.no_change(args))
    otr3, ow3, _ = opl.update(ok2, otr2, O.C.d({("z",): znew}), oargs)
    assert np.array_equal(_np(w3), np.broadcast_to(ow3, (n,))), (seed, "update w", J, T)
    assert np.array_equal(_np(tr3.get_score()), np.broadcast_to(otr3.get_score(), (n,))), (seed, "update score")
    c1 = rng.normal(size=J).astype(f32)
    a2, oa2 = (jnp.array(c1), jnp.array(xs)), (c1, xs)
    tr4, w4, _, _ = Update(C.n()).edit(k2, tr3, (Diff.unknown_change(a2[0]), Diff.no_change(a2[1])))
    otr4, ow4, _ = opl.update(ok2, otr3, O.ChoiceMap(), oa2)
    assert np.array_equal(_np(w4), np.broadcast_to(ow4, (n,))), (seed, "changed args w", J, T)
    # one address
    j, t = int(rng.integers(J)), int(rng.integers(T))
    tr5, w5, _, _ = Update(C[j, t, "y"].set(0.25)).edit(k2, tr4, Diff.no_change(a2))
    y5 = _np(tr5.get_choices()["y"]); y4 = _np(tr4.get_choices()["y"])
    assert np.all(y5[:, j, t] == f32(0.25)); y4[:, j, t] = 0.25; assert np.array_equal(y5, y4)
    ofull = np.array(otr4.get_choices()["y"]); ofull = np.broadcast_to(ofull, (n, J, T)).copy(); ofull[:, j, t] = 0.25
    otr5, ow5, _ = opl.update(ok2, otr4, O.C.d({("y",): ofull}), oa2)
    assert np.array_equal(_np(w5), np.broadcast_to(ow5, (n,))), (seed, "one address w", J, T)
    return J, T, n


def check_gather_at_group_indices(J=40, N=300, K=9, seed=3):
    """A hierarchical model's `normal(theta[group], s) @ "y"` — `theta` a latent vector of J group effects (a long
    vector-valued site: its values live in memory), `group` a table of N group labels: the gather is a RECIPE (element i =
    one load of `group`, one register-indexed load of theta: engine.StepInput.__getitem__) evaluated inside y's counted
    loop, not N unrolled reads — simulate / importance / assess / update of theta against the oracle, and against the
    unrolled form of the same program (engine.GATHER_LAZY_OFF), bit for bit."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, engine, numpy as jnp
    f32 = np.float32
    rng = np.random.default_rng(seed)
    grp = rng.integers(0, J, N).astype(np.int32)
    xs = rng.normal(size=N).astype(f32)

    @G.gen
    def m(a):
        theta = G.normal(a * jnp.ones(J), 2.0 * jnp.ones(J)) @ "theta"
        G.normal(theta[jnp.array(grp)] + jnp.array(xs), 0.5) @ "y"
        return jnp.sum(theta)

    @O.gen
    def om(a):
        theta = O.normal(np.asarray(a, f32)[..., None] * np.ones(J, f32), f32(2.0) * np.ones(J, f32)) @ "theta"
        O.normal((theta[..., grp] + xs).astype(f32), f32(0.5)) @ "y"
        return O.sum_vector(theta)
    dev = G._lib.get().device
    a_h = rng.normal(size=K).astype(f32)
    args, oargs = (torch.from_numpy(a_h).to(dev),), (a_h,)
    k, ok = G.split(G.key(seed), K), O.split(O.key(seed), K)
    tr, otr = m.simulate(k, args), om.simulate(ok, oargs)
    for ad in ("theta", "y"):
        assert np.array_equal(_np(tr.get_choices()[ad]), otr.get_choices()[ad]), ("simulate", ad)
    assert np.array_equal(_np(tr.get_score()), otr.get_score()) and np.array_equal(_np(tr.get_retval()), otr.get_retval())
    yobs = rng.normal(size=N).astype(f32)
    tr2, w2 = m.importance(k, C["y"].set(jnp.array(yobs)), args)
    otr2, ow2 = om.importance(ok, O.C.d({"y": yobs}), oargs)
    assert np.array_equal(_np(w2), ow2), "importance weight"
    s, _ = m.assess(tr2.get_choices(), args)
    assert np.array_equal(_np(s), _np(tr2.get_score()))
    th_new = rng.normal(size=(K, J)).astype(f32)
    k2, ok2 = G.split(G.key(seed + 1), K), O.split(O.key(seed + 1), K)
    tr3, w3, _, _ = Update(C["theta"].set(torch.from_numpy(th_new).to(dev))).edit(k2, tr2, Diff.no_change(args))
    otr3, ow3, _ = om.update(ok2, otr2, O.C.d({"theta": th_new}), oargs)
    assert np.array_equal(_np(w3), ow3) and np.array_equal(_np(tr3.get_score()), otr3.get_score()), "update of the group effects"
    G.clear_caches()
    engine.GATHER_LAZY_OFF[0] = 1
    try:
        tr_u = m.simulate(k, args)
    finally:
        engine.GATHER_LAZY_OFF[0] = 0
        G.clear_caches()
    assert np.array_equal(_np(tr_u.get_choices()["y"]), _np(tr.get_choices()["y"])) and np.array_equal(_np(tr_u.get_score()), _np(tr.get_score()))


def check_nested_marginal(k=129, seed=5):
    """A12 / F4: ChangeTarget.run_csmc_for_normalizing_constant (ref smc.py:432-465),
    estimate_reciprocal_normalizing_constant (:214-225) and Marginal.random_weighted with an inner algorithm
    (sp.py:229-238) against the oracle's restatement; and ChangeTarget.run_smc (:370-396) directly."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, SelectionBuilder as S, Target
    from genjax_amd.inference.smc import ChangeTarget, ImportanceK
    from genjax_amd.inference.sp import Marginal

    def mk(g):
        @g.gen
        def model():
            mu = g.normal(0.0, 1.0) @ "mu"
            s = g.normal(mu, 2.0) @ "s"
            x = g.normal(mu + 0.5 * s, 0.5) @ "x"
            return x
        return model
    m, om = mk(G), mk(O)
    prior, oprior = Target(m, (), G.ChoiceMap.empty()), O.Target(om, (), O.ChoiceMap.empty())
    alg, oalg = ImportanceK(prior, k_particles=k), O.ImportanceK(oprior, k)
    tgt, otgt = Target(m, (), C["x"].set(0.7)), O.Target(om, (), O.C.d({"x": np.float32(0.7)}))
    # ChangeTarget.run_smc: every particle re-weighted under the new target
    pc, opc = ChangeTarget(alg, tgt).run_smc(G.key(seed)), O.ChangeTarget(oalg, otgt).run_smc(O.key(seed))
    assert np.array_equal(pc.get_log_weights().cpu().numpy(), opc.get_log_weights())
    for a in ("mu", "s"):
        assert np.array_equal(pc.get_particles().get_choices()[a].cpu().numpy(), opc.get_particles().get_choices()[a])
    assert abs(float(pc.get_log_marginal_likelihood_estimate()) - float(opc.get_log_marginal_likelihood_estimate())) <= 1e-5
    # run_csmc_for_normalizing_constant / estimate_reciprocal_normalizing_constant
    ret, oret = C.d({"mu": 0.2, "s": -0.4}), O.C.d({"mu": np.float32(0.2), "s": np.float32(-0.4)})
    z = alg.estimate_reciprocal_normalizing_constant(G.key(seed + 1), tgt, ret, -1.25)
    oz = O.estimate_reciprocal_normalizing_constant(oalg, O.key(seed + 1), otgt, oret, np.float32(-1.25))
    assert abs(float(z) - float(oz)) <= 1e-5, (float(z), float(oz))
    z1 = ImportanceK(prior, k_particles=1).estimate_reciprocal_normalizing_constant(G.key(seed + 2), tgt, ret, 0.5)
    oz1 = O.estimate_reciprocal_normalizing_constant(O.ImportanceK(oprior, 1), O.key(seed + 2), otgt, oret, np.float32(0.5))
    assert abs(float(z1) - float(oz1)) <= 1e-6
    # Marginal over "x" whose weight comes from the inner algorithm: algorithms nest
    mar = Marginal(m, S["x"], algorithm=alg)
    w, chm = mar.random_weighted(G.key(seed + 3))
    ow, ochm = O.marginal_random_weighted(om, [("x",)], oalg, O.key(seed + 3), ())
    assert np.float32(chm["x"].cpu().numpy() if hasattr(chm["x"], "cpu") else chm["x"]) == np.float32(ochm["x"])
    assert abs(float(w) - float(ow)) <= 1e-5, (float(w), float(ow))
    # ... and as the proposal q of an outer ImportanceK (smc.py:301-305): q.random_weighted(key, target)
    @G.gen
    def outer():
        x = G.normal(0.0, 3.0) @ "x"
        G.normal(x, 1.0) @ "y"

    @G.gen
    def prop(target):
        mu = G.normal(0.0, 1.0) @ "mu"
        return G.normal(mu, 1.0) @ "x"
    otgt2 = Target(outer, (), C["y"].set(1.0))
    inner = ImportanceK(Target(prop, (otgt2,), G.ChoiceMap.empty()), k_particles=5)
    pcq = ImportanceK(otgt2, q=Marginal(prop, S["x"], algorithm=inner), k_particles=1).run_smc(G.key(seed + 4))
    assert bool(torch.isfinite(pcq.get_log_weights()).all())
    # estimate_logpdf of a Marginal with an algorithm = the algorithm's normalising-constant estimate
    lp = mar.estimate_logpdf(G.key(seed + 5), C["x"].set(0.7))
    olp = O.log_marginal_likelihood_estimate(oalg, O.key(seed + 5), otgt)
    assert abs(float(lp) - float(olp)) <= 1e-5


def check_masked_constraints(n=257, seed=9):
    """Indexed choice maps and Mask constraints (ref choice_map.py:1453-1531, distribution.py:129-142, 189-224):
    a plate constrained on a SUBSET of its indices (`C["ys", idx_array, "y"].set(v)`), and a runtime-conditional
    constraint Mask(value, flag) with one flag per particle — the select between the importance and the simulate
    branch runs in the site program (OP_SEL)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Mask, Update, Diff
    dev = G._lib.get().device
    sig = np.array([15, 10, 16, 11, 9, 11, 10, 18], np.float32)

    from genjax_amd import numpy as jnp

    def mk(g, plate):
        @g.gen
        def school(mu, s):
            th = g.normal(mu, 2.0) @ "theta"
            return g.normal(th, s) @ "y"

        @g.gen
        def model():
            mu = g.normal(0.0, 5.0) @ "mu"
            plate(school)(mu) @ "ys"
            return mu
        return model
    m = mk(G, lambda sch: (lambda mu: sch.vmap(in_axes=(None, 0))(mu, jnp.array(sig))))
    om = mk(O, lambda sch: (lambda mu: O.Vmap(sch, in_axes=(None, 0))(mu, sig)))
    idx = np.array([1, 4, 6])
    vals = np.array([8.0, -1.0, 18.0], np.float32)
    tr, w = m.importance(G.split(G.key(seed), n), C["ys", idx, "y"].set(vals), ())
    otr, ow = om.importance(O.split(O.key(seed), n), O.C.d({("ys", "y"): O.indexed(vals, idx, 8)}), ())
    y, oy = tr.get_choices()["ys", "y"].cpu().numpy(), otr.get_choices()["ys", "y"]
    assert np.array_equal(y, oy) and np.array_equal(y[:, idx], np.broadcast_to(vals, (n, 3)))
    assert not np.array_equal(y[:, 0], y[:, 2])                              # the other elements were sampled
    assert np.array_equal(w.cpu().numpy(), ow)                                # weight = the 3 constrained densities only
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
    # a list of indices, and a value with the index axis last on a per-particle tensor
    per = np.random.default_rng(seed).normal(size=(n, 3)).astype(np.float32)
    tr2, w2 = m.importance(G.split(G.key(seed + 1), n), C["ys", [1, 4, 6], "y"].set(torch.from_numpy(per).to(dev)), ())
    otr2, ow2 = om.importance(O.split(O.key(seed + 1), n), O.C.d({("ys", "y"): O.indexed(per, idx, 8)}), ())
    assert np.array_equal(tr2.get_choices()["ys", "y"].cpu().numpy(), otr2.get_choices()["ys", "y"])
    assert np.array_equal(w2.cpu().numpy(), ow2)

    # Mask(value, flag) with one flag per particle at a plain site
    def mk1(g):
        @g.gen
        def pair():
            x = g.normal(0.0, 1.0) @ "x"
            return g.normal(x, 0.5) @ "z"
        return pair
    p1, op1 = mk1(G), mk1(O)
    flags = np.random.default_rng(seed + 2).random(n) < 0.4
    xv = np.random.default_rng(seed + 3).normal(size=n).astype(np.float32)
    mk_t = lambda a: torch.from_numpy(a).to(dev)
    tr3, w3 = p1.importance(G.split(G.key(seed + 2), n), C["x"].set(Mask(mk_t(xv), mk_t(flags))), ())
    otr3, ow3 = op1.importance(O.split(O.key(seed + 2), n), O.C.d({"x": O.Mask(xv, flags)}), ())
    x3 = tr3.get_choices()["x"].cpu().numpy()
    assert np.array_equal(x3, otr3.get_choices()["x"]) and np.array_equal(x3[flags], xv[flags])
    assert np.array_equal(w3.cpu().numpy(), ow3) and np.all(w3.cpu().numpy()[~flags] == 0.0)
    assert np.array_equal(tr3.get_score().cpu().numpy(), otr3.get_score())
    # flags decided on the host: plain constrained / unconstrained generate
    tr4, w4 = p1.importance(G.split(G.key(seed + 2), n), C["x"].set(Mask(mk_t(xv), True)), ())
    tr5, w5 = p1.importance(G.split(G.key(seed + 2), n), C["x"].set(mk_t(xv)), ())
    assert np.array_equal(w4.cpu().numpy(), w5.cpu().numpy())
    tr6, w6 = p1.importance(G.split(G.key(seed + 2), n), C["x"].set(Mask(mk_t(xv), False)), ())
    assert np.all(w6.cpu().numpy() == 0.0)
    # Update with a masked constraint (distribution.py:189-224): new value where the flag holds, old one elsewhere
    u, wu, _, bwd = Update(C["x"].set(Mask(mk_t(xv), mk_t(flags)))).edit(G.split(G.key(seed + 5), n), tr6, Diff.no_change(()))
    ou, owu, _ = op1.update(O.split(O.key(seed + 5), n), op1.importance(O.split(O.key(seed + 2), n), O.ChoiceMap.empty(), ())[0],
                            O.C.d({"x": O.Mask(xv, flags)}), ())
    assert np.array_equal(u.get_choices()["x"].cpu().numpy(), ou.get_choices()["x"])
    assert np.array_equal(wu.cpu().numpy(), owu)
    assert np.array_equal(u.get_score().cpu().numpy(), ou.get_score())


def mk2(g):
    @g.gen
    def model():
        p = g.beta(2.0, 2.0) @ "p"
        return g.flip(p) @ "v"
    return model


# ---------------------------------------------------------------------------
# BASELINE config 5: mixture-model cluster assignments (integer gate)
# ---------------------------------------------------------------------------
def check_mixture_assignments(n=3000, K=64, seed=11, specialize=False):
    """gibbs_categorical == the oracle's materialised [n, K] categorical draw, bit for bit."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, workloads
    from genjax_amd.inference import gibbs

    class _ONP:
        log = staticmethod(O.log)
    be = __import__("genjax_amd")._lib.get()
    x, guess, probs, z = workloads.mixture_data(n, K)
    gd, ogd = workloads.make_mixture(G), workloads.make_mixture(O, jnp=_ONP)
    dev = be.device
    args = (torch.from_numpy(probs).to(dev), torch.from_numpy(guess).to(dev))
    chm = C["obs"].set(torch.from_numpy(x).to(dev))
    if specialize:
        gibbs.gibbs_categorical(G.key(seed), gd, args, C["obs"].set(torch.from_numpy(x[:100]).to(dev)), "idx", K)   # 100 != K: shapes decide what is per-datapoint
        for comp, _ in gibbs._CACHE.values():
            assert comp.specialize(), "hiprtc specialisation failed"
    idx = gibbs.gibbs_categorical(G.key(seed), gd, args, chm, "idx", K)
    oidx, _ = O.gibbs_categorical(O.key(seed), ogd, (probs, guess), O.C.d({"obs": x}), "idx", K, n)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (n,)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert (oidx == z).mean() > 0.85            # well-separated clusters: mostly the generating component
    # shards of the dataset with their global offsets reproduce the whole (how ranks split config 5)
    cut = n // 3
    a = gibbs.gibbs_categorical(G.key(seed), gd, args, C["obs"].set(torch.from_numpy(x[:cut]).to(dev)), "idx", K)
    b = gibbs.gibbs_categorical(G.key(seed), gd, args, C["obs"].set(torch.from_numpy(x[cut:]).to(dev)), "idx", K,
                                index_offset=cut)
    assert np.array_equal(np.concatenate([a.cpu().numpy(), b.cpu().numpy()]), oidx)
    return idx


def check_mixture_gibbs_through_the_plate(n=5000, K=64, seed=11, timing=False):
    """BASELINE config 5 THROUGH THE GFI (SURVEY 8f item 2's parenthetical): the datapoints are a
    `generate_datapoint.repeat(n=N)` plate called directly (its elements on the launch axis), the initial trace comes
    from `importance` with the observations constrained, and one assignment sweep is `gibbs.enumerative_gibbs` on that
    trace — the notebook's `update_datapoint_assignment`: the new assignments equal the oracle's materialised [n, K]
    categorical draw bit for bit, the updated trace holds them, its score is the oracle's re-assessment (fixed-tree
    plate sum), and the weight is new score - old score."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, workloads
    from genjax_amd.inference import gibbs

    class _ONP:
        log = staticmethod(O.log)
    dev = G._lib.get().device
    x, guess, probs, z = workloads.mixture_data(n, K)
    gd, ogd = workloads.make_mixture(G), workloads.make_mixture(O, jnp=_ONP)
    plate = gd.repeat(n=n)
    args = (torch.from_numpy(probs).to(dev), torch.from_numpy(guess).to(dev))
    tr, w0 = plate.importance(G.key(seed), C["obs"].set(torch.from_numpy(x).to(dev)), args)
    assert tuple(tr.batch_shape) == () and tuple(tr.get_choices()["idx"].shape) == (n,)
    oplate = O.Repeat(ogd, n)
    otr, ow0 = oplate.importance(O.key(seed), O.C.d({"obs": x}), (probs, guess))
    assert np.array_equal(tr.get_choices()["idx"].cpu().numpy(), otr.get_choices()["idx"])        # the prior draws
    assert float(w0) == float(ow0) and float(tr.get_score()) == float(otr.get_score())
    new_tr, idx, w = gibbs.enumerative_gibbs(G.key(seed + 1), tr, "idx", K)
    k1, _k2 = O.split(O.key(seed + 1))
    oidx, _ = O.gibbs_categorical(k1, ogd, (probs, guess), O.C.d({"obs": x}), "idx", K, n)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(new_tr.get_choices()["idx"].cpu().numpy(), oidx)
    assert np.array_equal(new_tr.get_choices()["obs"].cpu().numpy(), x)
    so, _ = oplate.assess(O.C.d({"obs": x, "idx": oidx}), (probs, guess), ())
    assert float(new_tr.get_score()) == float(so)
    # (the weight is the fixed-tree sum of the per-element differences; against the difference of the two sums)
    assert abs(float(w) - (float(so) - float(otr.get_score()))) <= 4e-6 * max(1.0, abs(float(so)), abs(float(otr.get_score())))
    assert (oidx == z).mean() > 0.85
    if not timing:
        return None
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)

    def timed(fn, reps=10, rounds=3):
        fn(); sync()
        best = float("inf")
        for _ in range(rounds):          # (the best of three rounds: one round was seen at twice its usual time on a shared box)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            best = min(best, (time.perf_counter() - t0) / reps)
        return best
    chm = C["obs"].set(torch.from_numpy(x).to(dev))
    t_plate = timed(lambda: gibbs.enumerative_gibbs(G.key(seed + 1), tr, "idx", K))
    t_bare = timed(lambda: gibbs.gibbs_categorical(G.key(seed + 1), gd, args, chm, "idx", K))
    return t_plate, t_bare


# ---------------------------------------------------------------------------
# Scan combinator: a short state-space model as ONE generative function
# ---------------------------------------------------------------------------
def check_scan(n=257, T=6, seed=5):
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    ys = np.array([0.3, -0.2, 0.5, 1.0, 0.7, -0.4, 0.1, 0.9][:T], np.float32)

    def mk(g):
        @g.gen
        def step(x, t):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, xn * 2.0
        return step
    step, ostep = mk(G), mk(O)

    @G.gen
    def ssm():
        x0 = G.normal(0.0, 1.0) @ "x0"
        xT, doubled = step.scan(n=T)(x0, jnp.arange(T).astype("float32")) @ "steps"
        return xT

    @O.gen
    def o_ssm():
        x0 = O.normal(0.0, 1.0) @ "x0"
        xT, doubled = O.Scan(ostep, T)(x0, np.arange(T, dtype=np.float32)) @ "steps"
        return xT
    tr, w = ssm.importance(G.split(G.key(seed), n), C["steps", :, "y"].set(ys), ())
    tro, wo = o_ssm.importance(O.split(O.key(seed), n), O.C.d({("steps", "y"): ys}), ())
    x = tr.get_choices()["steps", "x"]
    assert tuple(x.shape) == (n, T)
    assert np.array_equal(x.cpu().numpy(), tro.get_choices()["steps", "x"])          # chained keys fold_in(key, t)
    assert np.array_equal(w.cpu().numpy(), wo)
    assert np.array_equal(tr.get_score().cpu().numpy(), tro.get_score())
    assert np.array_equal(tr.get_retval().cpu().numpy(), tro.get_retval())
    tr2, tro2 = ssm.simulate(G.split(G.key(seed + 1), n), ()), o_ssm.simulate(O.split(O.key(seed + 1), n), ())
    assert np.array_equal(tr2.get_choices()["steps", "y"].cpu().numpy(), tro2.get_choices()["steps", "y"])
    s, _ = ssm.assess(tr2.get_choices(), ())
    so, _ = o_ssm.assess(tro2.get_choices(), (), (n,))
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(s.cpu().numpy(), tr2.get_score().cpu().numpy())
    # a constraint at ONE step through its integer address (constraint.get_submap(idx), scan.py:262):
    # the weight is that step's observation density alone
    tr3, w3 = ssm.importance(G.split(G.key(seed), n), C["steps", 2, "y"].set(0.25), ())
    y3 = tr3.get_choices()["steps", "y"].cpu().numpy()
    assert np.all(y3[:, 2] == np.float32(0.25)) and not np.all(y3[:, 1] == np.float32(0.25))
    x3 = tr3.get_choices()["steps", "x"].cpu().numpy()[:, 2]
    assert np.array_equal(w3.cpu().numpy(), O.normal.assess(O.C.choice(np.full(n, 0.25, np.float32)), (x3, np.float32(1.0)), (n,))[0])
    # edits through a scan (scan.py:417-594): Update of every step's observation, then Regenerate of the
    # latent path — chained keys, carries threaded through the edited predecessors
    from genjax_amd import Diff, Regenerate, SelectionBuilder as S, Update
    dev = G._lib.get().device
    sc5, osc5 = step.scan(n=T), O.Scan(ostep, T)
    a5 = (torch.zeros(n, device=dev), jnp.zeros(T))
    oa5 = (np.zeros(n, np.float32), np.zeros(T, np.float32))
    t5, ot5 = sc5.simulate(G.split(G.key(seed + 3), n), a5), osc5.simulate(O.split(O.key(seed + 3), n), oa5)
    u5, w5, _, bwd5 = Update(C["y"].set(ys)).edit(G.split(G.key(seed + 4), n), t5, Diff.no_change(a5))
    ou5, ow5 = O.scan_edit(osc5, O.split(O.key(seed + 4), n), ot5, oa5, update=O.C.d({"y": ys}))
    assert np.array_equal(w5.cpu().numpy(), ow5)
    assert np.array_equal(u5.get_score().cpu().numpy(), ou5.get_score())
    assert np.array_equal(bwd5.constraint["y"].cpu().numpy(), ot5.get_choices()["y"])          # discard = old values
    r5, wr5, _, _ = Regenerate(S["x"]).edit(G.split(G.key(seed + 5), n), u5, Diff.no_change(a5))
    or5, owr5 = O.scan_edit(osc5, O.split(O.key(seed + 5), n), ou5, oa5, regenerate=O.selection("x"))
    assert np.array_equal(r5.get_choices()["x"].cpu().numpy(), or5.get_choices()["x"])
    assert np.array_equal(wr5.cpu().numpy(), owr5)
    assert np.array_equal(r5.get_score().cpu().numpy(), or5.get_score())
    # IndexRequest on a scan (scan.py:325-416): edit one step, re-score the next against the new carry
    from genjax_amd import IndexRequest
    for idx in (2, T - 1):
        e5, we5, _, be5 = IndexRequest(idx, Regenerate(S["x"])).edit(G.split(G.key(seed + 6), n), r5, Diff.no_change(a5))
        oe5, owe5 = O.scan_edit_index(osc5, O.split(O.key(seed + 6), n), or5, oa5, idx,
                                      lambda k, sl, a: ostep.regenerate(k, sl, O.selection("x"), a)[:2])
        xe, xo_ = e5.get_choices()["x"].cpu().numpy(), oe5.get_choices()["x"]
        assert np.array_equal(xe, xo_) and np.array_equal(we5.cpu().numpy(), owe5)
        assert np.array_equal(e5.get_score().cpu().numpy(), oe5.get_score())
        old = r5.get_choices()["x"].cpu().numpy()
        assert np.array_equal(np.delete(old, idx, axis=1), np.delete(xe, idx, axis=1))      # only step idx moved
        assert isinstance(be5, IndexRequest) and be5.idx == idx
    # one index PER PARTICLE: every step is edited in the program and selected where idx == t; the oracle runs
    # the static-idx edit for every t and takes particle i's result from the run with t = idx_i
    idx = np.random.default_rng(seed + 9).integers(0, T, n).astype(np.int32)
    e6, we6, _, be6 = IndexRequest(torch.from_numpy(idx).to(dev), Regenerate(S["x"])).edit(
        G.split(G.key(seed + 7), n), r5, Diff.no_change(a5))
    xo6 = or5.get_choices()["x"].copy()
    wo6, so6 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t_ in range(T):
        cand, wt = O.scan_edit_index(osc5, O.split(O.key(seed + 7), n), or5, oa5, t_,
                                     lambda k, sl, a: ostep.regenerate(k, sl, O.selection("x"), a)[:2])
        m_ = idx == t_
        xo6[m_] = cand.get_choices()["x"][m_]
        wo6[m_], so6[m_] = np.asarray(wt, np.float32)[m_], np.asarray(cand.get_score(), np.float32)[m_]
    assert np.array_equal(e6.get_choices()["x"].cpu().numpy(), xo6)
    assert np.array_equal(we6.cpu().numpy(), wo6)
    assert np.array_equal(e6.get_score().cpu().numpy(), so6)
    moved = e6.get_choices()["x"].cpu().numpy() != r5.get_choices()["x"].cpu().numpy()
    assert np.array_equal(moved, np.arange(T)[None, :] == idx[:, None])                  # exactly step idx_i moved
    # a Scan used directly: the chain starts at the caller's key
    sc = step.scan(n=3)
    t4 = sc.simulate(G.split(G.key(seed + 2), 16), (torch.zeros(16, device=G._lib.get().device), jnp.zeros(3)))
    t4o = O.Scan(ostep, 3).simulate(O.split(O.key(seed + 2), 16), (np.zeros(16, np.float32), np.zeros(3, np.float32)))
    assert np.array_equal(t4.get_choices()["x"].cpu().numpy(), t4o.get_choices()["x"])


def check_scan_long(n=257, T=100, seed=5):
    """A LONG scan (T > 16) runs as a counted loop in the site program (ref scan.py:200-294 `lax.scan`, :638-664):
    BASELINE config 2's model as ONE generative function — `step.scan(n=T)` inside a @gen model — importance /
    simulate / assess for all T steps in one launch; chained keys fold_in(key, t); addresses ["steps", t, "x"]."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp, workloads
    ys = workloads.lgssm_data(T + 1)

    def mk(g):
        @g.gen
        def step(x, t):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, xn * 2.0
        return step
    step, ostep = mk(G), mk(O)

    @G.gen
    def ssm():
        x0 = G.normal(0.0, 1.0) @ "x0"
        G.normal(x0, 1.0) @ "y0"
        xT, doubled = step.scan(n=T)(x0, jnp.arange(T).astype("float32")) @ "steps"
        return xT

    @O.gen
    def o_ssm():
        x0 = O.normal(0.0, 1.0) @ "x0"
        O.normal(x0, 1.0) @ "y0"
        xT, doubled = O.Scan(ostep, T)(x0, np.arange(T, dtype=np.float32)) @ "steps"
        return xT
    con = C["steps", :, "y"].set(ys[1:]).set(("y0",), float(ys[0]))
    ocon = O.C.d({("steps", "y"): ys[1:], ("y0",): np.float32(ys[0])})
    tr, w = ssm.importance(G.split(G.key(seed), n), con, ())
    tro, wo = o_ssm.importance(O.split(O.key(seed), n), ocon, ())
    x = tr.get_choices()["steps", "x"]
    assert tuple(x.shape) == (n, T)
    assert np.array_equal(x.cpu().numpy(), tro.get_choices()["steps", "x"])          # chained keys fold_in(key, t)
    assert np.array_equal(w.cpu().numpy(), wo)
    assert np.array_equal(tr.get_score().cpu().numpy(), tro.get_score())
    assert np.array_equal(tr.get_retval().cpu().numpy(), tro.get_retval())
    assert np.array_equal(tr.get_choices()["steps", "y"].cpu().numpy(), np.broadcast_to(ys[1:], (n, T)))
    assert np.array_equal(tr.get_subtrace("steps").get_subtrace("x").get_score().cpu().numpy(), tro.subtraces["steps"].subtraces["x"].score)
    # simulate, then assess the simulated choices (per-particle [n, T] leaves read step by step)
    tr2, tro2 = ssm.simulate(G.split(G.key(seed + 1), n), ()), o_ssm.simulate(O.split(O.key(seed + 1), n), ())
    assert np.array_equal(tr2.get_choices()["steps", "y"].cpu().numpy(), tro2.get_choices()["steps", "y"])
    s, _ = ssm.assess(tr2.get_choices(), ())
    so, _ = o_ssm.assess(tro2.get_choices(), (), (n,))
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(s.cpu().numpy(), tr2.get_score().cpu().numpy())
    # the scan used directly: the chain starts at the caller's key; stacked outputs come back as [n, T]
    dev = G._lib.get().device
    sc = step.scan(n=T)
    t4 = sc.simulate(G.split(G.key(seed + 2), n), (torch.zeros(n, device=dev), jnp.zeros(T)))
    t4o = O.Scan(ostep, T).simulate(O.split(O.key(seed + 2), n), (np.zeros(n, np.float32), np.zeros(T, np.float32)))
    assert np.array_equal(t4.get_choices()["x"].cpu().numpy(), t4o.get_choices()["x"])
    carry, doubled = t4.get_retval()
    ocarry, odoubled = t4o.get_retval()
    assert np.array_equal(carry.cpu().numpy(), ocarry) and np.array_equal(doubled.cpu().numpy(), odoubled)
    # constraints at single steps through integer addresses (constraint.get_submap(idx), scan.py:262): in the loop
    # they are masked constraints `Mask(v, t == idx)`; the weight is those steps' observation densities alone
    tr5, w5 = sc.importance(G.split(G.key(seed + 3), n), C[2, "y"].set(0.25).set((T - 3, "y"), -1.5),
                            (torch.zeros(n, device=dev), jnp.zeros(T)))
    y5, x5 = tr5.get_choices()["y"].cpu().numpy(), tr5.get_choices()["x"].cpu().numpy()
    assert np.all(y5[:, 2] == np.float32(0.25)) and np.all(y5[:, T - 3] == np.float32(-1.5))
    assert not np.all(y5[:, 3] == np.float32(0.25))
    w_ref = (O.normal.assess(O.C.choice(np.full(n, 0.25, np.float32)), (x5[:, 2], np.float32(1.0)), (n,))[0]
             + O.normal.assess(O.C.choice(np.full(n, -1.5, np.float32)), (x5[:, T - 3], np.float32(1.0)), (n,))[0])
    assert np.array_equal(w5.cpu().numpy(), w_ref)
    return dict(log_ml_is=float(torch.logsumexp(w.double(), 0) - math.log(n)), kalman=workloads.kalman_log_ml(ys))


# ---------------------------------------------------------------------------
# edits of plates: Vmap.edit = Update | IndexRequest (vmap.py:236-362)
# ---------------------------------------------------------------------------
def check_plate_edits(n=257, seed=1):
    import genjax_amd as G
    from genjax_amd import (ChoiceMapBuilder as C, Diff, IndexRequest, NotSupportedEditRequest, Regenerate,
                            SelectionBuilder as S, StaticRequest, Update, numpy as jnp)
    from genjax_amd.static import run_mh
    sig, ys = SCHOOL_SIGMA, SCHOOL_Y
    school, oschool = _school(G), _school(O)
    v, ov = school.vmap(in_axes=(None, None, 0)), O.Vmap(oschool, in_axes=(None, None, 0))
    args = (1.0, 2.0, jnp.array(sig))
    oargs = (np.float32(1.0), np.float32(2.0), np.array(sig, np.float32))
    tr, otr = v.simulate(G.split(G.key(seed), n), args), ov.simulate(O.split(O.key(seed), n), oargs)
    # Update of every element's "y": keys split(key, n), w = sum over the plate, discard = old values
    new_tr, w, _, bwd = Update(C["y"].set(ys)).edit(G.split(G.key(seed + 1), n), tr, Diff.no_change(args))
    onew, ow, odisc = O.vmap_update(ov, O.split(O.key(seed + 1), n), otr, O.C.d({"y": ys}), oargs)
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(new_tr.get_score().cpu().numpy(), onew.get_score())
    assert np.array_equal(bwd.constraint["y"].cpu().numpy(), odisc["y"])
    assert np.array_equal(new_tr.get_choices()["theta"].cpu().numpy(), otr.get_choices()["theta"])    # untouched
    # IndexRequest(3, Regenerate(theta)): element 3 only, with the caller's key
    tr2, w2, _, bwd2 = IndexRequest(3, Regenerate(S["theta"])).edit(G.split(G.key(seed + 2), n), new_tr, Diff.no_change(args))
    a3 = (np.float32(1.0), np.float32(2.0), np.float32(sig[3]))
    otr2, ow2 = O.vmap_edit_index(ov, O.split(O.key(seed + 2), n), onew, 3,
                                  lambda k, sl, a: oschool.regenerate(k, sl, O.selection("theta"), a)[:2], a3)
    th2 = tr2.get_choices()["theta"].cpu().numpy()
    assert np.array_equal(th2, otr2.get_choices()["theta"])
    assert np.array_equal(w2.cpu().numpy(), ow2)
    assert np.array_equal(tr2.get_score().cpu().numpy(), otr2.get_score())
    th1 = new_tr.get_choices()["theta"].cpu().numpy()
    assert np.array_equal(np.delete(th1, 3, axis=1), np.delete(th2, 3, axis=1)) and not np.array_equal(th1[:, 3], th2[:, 3])
    assert isinstance(bwd2, IndexRequest) and bwd2.idx == 3
    # IndexRequest(5, StaticRequest({theta: Rejuvenate})): the MH proposal on one element
    rej = G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 1.0))
    orej = {"theta": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(1.0)))}
    tr3, w3, _, _ = IndexRequest(5, StaticRequest({"theta": rej})).edit(G.split(G.key(seed + 3), n), tr2, Diff.no_change(args))
    a5 = (np.float32(1.0), np.float32(2.0), np.float32(sig[5]))
    otr3, ow3 = O.vmap_edit_index(ov, O.split(O.key(seed + 3), n), otr2, 5,
                                  lambda k, sl, a: oschool.edit_static(k, sl, orej, a), a5)
    assert np.array_equal(tr3.get_choices()["theta"].cpu().numpy(), otr3.get_choices()["theta"])
    assert np.array_equal(w3.cpu().numpy(), ow3)
    # nested in a model + the fused MH accept: only school 2's theta may move, and only where accepted
    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        return school.vmap(in_axes=(None, None, 0))(mu, 2.0, jnp.array(sig)) @ "schools"
    trm, _ = schools.importance(G.split(G.key(seed + 4), n), C["schools", :, "y"].set(ys), ())
    req = StaticRequest({"schools": IndexRequest(2, StaticRequest({"theta": rej}))})
    sel, acc, wmh = run_mh(schools, G.split(G.key(seed + 5), n), trm, req, Diff.no_change(()))
    t0, t1 = trm.get_choices()["schools", "theta"].cpu().numpy(), sel.get_choices()["schools", "theta"].cpu().numpy()
    a = acc.cpu().numpy().astype(bool)
    assert np.array_equal(np.delete(t0, 2, axis=1), np.delete(t1, 2, axis=1))
    assert np.array_equal(t0[~a, 2], t1[~a, 2]) and np.all(t0[a, 2] != t1[a, 2]) and 0 < a.mean() < 1
    # one index PER PARTICLE (IntArray idx): every element is edited in the program and selected where idx == j
    idx = np.random.default_rng(seed).integers(0, 8, n).astype(np.int32)
    dev = G._lib.get().device
    tr4, w4, _, _ = IndexRequest(torch.from_numpy(idx).to(dev), Regenerate(S["theta"])).edit(
        G.split(G.key(seed + 6), n), tr, Diff.no_change(args))
    th, wo = otr.get_choices()["theta"].copy(), np.zeros(n, np.float32)
    for j in range(8):
        aj = (np.float32(1.0), np.float32(2.0), np.float32(sig[j]))
        cand, wj = O.vmap_edit_index(ov, O.split(O.key(seed + 6), n), otr, j,
                                     lambda k, sl, a: oschool.regenerate(k, sl, O.selection("theta"), a)[:2], aj)
        m = idx == j
        th[m, j], wo[m] = cand.get_choices()["theta"][m, j], np.asarray(wj, np.float32)[m]
    assert np.array_equal(tr4.get_choices()["theta"].cpu().numpy(), th) and np.array_equal(w4.cpu().numpy(), wo)
    # anything else is refused, as in the reference (vmap.py:361-362)
    try:
        Regenerate(S["theta"]).edit(G.split(G.key(seed), n), tr, Diff.no_change(args))
        raise AssertionError("Regenerate on a plate should be refused")
    except NotSupportedEditRequest:
        pass


# ---------------------------------------------------------------------------
# HMC: one launch = L leapfrog steps + L+1 reverse-mode gradients
# ---------------------------------------------------------------------------
def check_hmc(n=257, seed=3):
    import genjax_amd as G
    from genjax_amd import ChoiceMap, Diff, SelectionBuilder as S
    from genjax_amd.inference.requests import HMC, SafeHMC

    def mk(g):
        @g.gen
        def model():
            x = g.normal(0.0, 1.0) @ "x"
            y = g.normal(x, 0.01) @ "y"
            return y

        @g.gen
        def chain():
            mu = g.normal(0.0, 2.0) @ "mu"
            x = g.normal(mu * 0.5 + 1.0, 1.5) @ "x"
            g.normal(x * x * 0.1 + mu, 0.3) @ "y"
            return x
        return model, chain
    (model, chain), (omodel, ochain) = mk(G), mk(O)
    # --- tests/inference/test_requests.py:197-235 (test_simple_normal_hmc), one particle, no accept step ---
    key = G.key(0)
    key, sub_key = G.split(key)
    tr, _ = model.importance(sub_key, ChoiceMap.kw(y=3.0), ())
    request = HMC(S["x"], 1e-2)
    new_tr, fwd_w, _, bwd = request.edit(key, tr, Diff.no_change(()))
    lp = lambda t: float(G.normal.logpdf(t.get_choices()["x"], 0.0, 1.0)) + \
        float(G.normal.logpdf(t.get_choices()["y"], t.get_choices()["x"], 0.01))
    assert float(fwd_w) != 0.0
    assert float(new_tr.get_score() - tr.get_score()) == pytest_approx(lp(new_tr) - lp(tr), 1e-6)
    assert float(fwd_w) - float(new_tr.get_score() - tr.get_score()) != 0.0
    assert isinstance(bwd, HMC)
    cur = tr
    for _ in range(20):
        key, sub_key = G.split(key)
        cur, *_ = request.edit(sub_key, cur, Diff.no_change(()))
    assert abs(float(cur.get_choices()["x"]) - 3.0) <= 3.0 * 5e-3            # pytest.approx(3.0, 5e-3)
    # --- against the oracle (forward-mode duals there, reverse-mode IR here), batched ---
    for gm, om, sel, osel, obs, tol in ((model, omodel, S["x"], ["x"], 3.0, 0.0),
                                        (chain, ochain, S["x"] | S["mu"], ["x", "mu"], 0.7, 2e-5)):
        trb, _ = gm.importance(G.split(G.key(seed), n), ChoiceMap.kw(y=obs), ())
        otrb, _ = om.importance(O.split(O.key(seed), n), O.C.kw(y=np.float32(obs)), ())
        ntr, w, _, _ = HMC(sel, 1e-2, L=10).edit(G.split(G.key(seed + 1), n), trb, Diff.no_change(()))
        ontr, ow = O.hmc_edit(O.split(O.key(seed + 1), n), otrb, osel, 1e-2, 10, ())
        for a in osel:
            x, ox = ntr.get_choices()[a].cpu().numpy(), np.asarray(ontr.get_choices()[a], np.float32)
            assert np.allclose(x, ox, rtol=tol, atol=tol) if tol else np.array_equal(x, ox), a
        wa = w.cpu().numpy()
        assert np.allclose(wa, ow, rtol=max(tol, 0) * 50, atol=max(tol, 0) * 50) if tol else np.array_equal(wa, ow)
        assert np.allclose(ntr.get_score().cpu().numpy(), np.asarray(ontr.get_score(), np.float32), rtol=1e-5, atol=1e-4)
    # --- tests/inference/test_requests.py:237-255 (test_simple_scan_hmc): HMC over all 10 steps of a scan;
    # > 32 live values per chain (values, momenta, initial gradient: 3 x 10), i.e. a specialised kernel on the GPU
    from genjax_amd import Selection, numpy as jnp

    @G.gen
    def kernel(z, scanned_in):
        z = G.normal(z, 1.0) @ "x"
        _ = G.normal(z, 0.01) @ "y"
        return z, None
    key = G.key(0)
    key, sub_key = G.split(key)
    smodel = kernel.scan(n=10)
    str_, _ = smodel.importance(sub_key, ChoiceMap.empty().at["y"].set(3.0 * jnp.ones(10)), (0.0, None))
    srequest = HMC(Selection.at["x"], jnp.array(1e-2))
    cur = str_
    for _ in range(50):
        key, sub_key = G.split(key)
        cur, *_ = srequest.edit(sub_key, cur, Diff.no_change((0.0, None)))
    xs = cur.get_choices()["x"].cpu().numpy()
    assert xs.shape == (10,) and np.all(np.abs(xs - 3.0) <= 3.0 * 8e-3)       # pytest.approx(3.0, 8e-3)
    # SafeHMC = HMC + an assertion on the retdiff (hmc.py:217-227): fine when the return value (the
    # constrained y) cannot move, trips when it depends on a selected choice
    SafeHMC(S["x"], 1e-2).edit(G.key(1), tr, Diff.no_change(()))
    try:
        SafeHMC(S["x"], 1e-2).edit(G.key(1), chain.importance(G.key(2), ChoiceMap.kw(y=0.7), ())[0], Diff.no_change(()))
        raise RuntimeError("SafeHMC should assert: the return value depends on x")
    except AssertionError:
        pass


def pytest_approx(x, rel):
    import pytest
    return pytest.approx(x, rel)


def oracle_nlssm_mh_sweep(n, T, seed):
    """The oracle run of BASELINE config 3 with the sweeps' key schedule: per step t the key fold_in(key, t)
    splits into (k_prop, k_res, k_mh); the resampling that follows step t-1 uses ITS k_res; the MH move
    before step t uses k_mh of step t.  Returns the last step's particles / log-weights / accept bits,
    the evidence terms and the particles after the final resampling."""
    from genjax_amd import workloads
    ys = workloads.nlssm_data(T)
    oi, ost = workloads.make_nlssm(O)
    oreq = {"x": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.5)))}
    okey = O.key(seed)
    otr = olw = oacc = None
    terms = []
    for t in range(T):
        oks = O.split(O.fold_in(okey, t), 3)
        oobs = O.C.kw(y=np.float32(ys[t]))
        if t == 0:
            otr, olw = oi.importance(O.split(oks[0], n), oobs, ())
        else:
            okr = O.split(O.fold_in(okey, t - 1), 3)[1]
            cdf, total, M, shift = O.weight_cdf(olw)
            terms.append(O.log_ml_increment(M, total, shift, n))
            otr = O.gather_trace(otr, (O.ancestors_c if n > 20_000 else O.ancestors)(O.SYSTEMATIC, okr, cdf))
            gf = otr.get_gen_fn()
            otr, oacc, _ = O.rejuvenate(oks[2], otr, lambda k, tr_: gf.edit_static(k, tr_, oreq, tr_.get_args()))
            otr, olw = ost.importance(O.split(oks[0], n), oobs, (np.asarray(otr.get_retval(), np.float32), np.float32(t)))
    x = np.asarray(otr.get_retval(), np.float32)
    olw = np.asarray(olw, np.float32)
    cdf, total, M, shift = O.weight_cdf(olw)
    terms.append(O.log_ml_increment(M, total, shift, n))
    anc = (O.ancestors_c if n > 20_000 else O.ancestors)(O.SYSTEMATIC, O.split(O.fold_in(okey, T - 1), 3)[1], cdf)
    return {"x": x, "lw": olw, "acc": oacc, "terms": terms, "resampled": x[anc]}


def load_golden(name):
    """a record of tests/golden/full_size.json (tests/golden/make_full_size.py wrote it from the oracle)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size.json")) as fh:
        return json.load(fh)[name]


def equals_golden(a, rec, index) -> bool:
    """the device array against a committed oracle array: sha256 of the raw bytes AND the sampled entries"""
    import hashlib
    a = np.ascontiguousarray(a)
    if list(a.shape) != rec["shape"] or str(a.dtype) != rec["dtype"]:
        return False
    flat = a.reshape(-1)[np.asarray(index)]
    got = [float(np.float32(x)).hex() for x in flat] if a.dtype.kind == "f" else [int(x) for x in flat.astype(np.uint8 if a.dtype == np.bool_ else flat.dtype)]
    return got == rec["sample"] and hashlib.sha256(a.tobytes()).hexdigest() == rec["sha256"]


def check_nlssm_mh_sweep(n=2000, T=5, seed=7, capture=False, specialize=False, want_chained=None, noise_ahead=None,
                         chain_mh=True, noise_roots=None, fuse_resample=None, golden=None):
    """BASELINE config 3 as ONE captured sweep: BootstrapSweep(rejuvenate=...) (k_vm -> resample ->
    fused MH -> k_vm ...) against the oracle run step by step with the sweep's key schedule
    (step key fold_in(key, t) -> (k_prop, k_res, k_mh); resampling of step t-1 uses ITS k_res)."""
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference import smc
    ys = workloads.nlssm_data(T)
    init, step = workloads.make_nlssm(G)
    oi, ost = workloads.make_nlssm(O)
    req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
    oreq = {"x": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.5)))}
    sw = smc.BootstrapSweep(init, step, n, T, step_extra=lambda t: (float(t),), rejuvenate=req, specialize=specialize,
                            noise_ahead=noise_ahead, chain_mh=chain_mh, noise_roots=noise_roots,
                            fuse_resample=fuse_resample).prepare(G.key(seed), torch.from_numpy(ys))
    if fuse_resample is not None:
        assert sw.fuse == fuse_resample
    if noise_ahead is not None:
        assert sw.noise_ahead == noise_ahead
        if noise_ahead:     # the move's proposal and accept draws (launch key) and / or the extension's draw (its own key)
            roots = noise_roots or smc.BootstrapSweep.NOISE_ROOTS_MH
            want = [("KSPLITU", "normal"), ("LDKEY", "normal"), ("LDKEY", "uniform")]
            if roots != "all":
                want = [w for w in want if w[0] in roots.split(",")]
            assert sorted((d[0], d[3]) for d in sw.p_mhvm_step.noise) == want
    if want_chained is not None:
        assert sw.fuse_mh == want_chained, "the sweep did not take the requested (chained / two-launch) MH form"
    if capture:
        sw.capture()
    sw.launch()
    x, lw, anc = sw.state()
    if golden is not None:           # the oracle's run of this very sweep, committed (tests/golden/make_full_size.py)
        rec = load_golden(golden)
        assert (rec["n"], rec["T"], rec["seed"]) == (n, T, seed)
        assert equals_golden(x.cpu().numpy(), rec["x"], rec["index"]), "particles differ from the committed oracle run"
        assert equals_golden(lw.cpu().numpy(), rec["lw"], rec["index"]), "log-weights differ from the committed oracle run"
        assert equals_golden(sw.accept.cpu().numpy().astype(np.bool_), rec["acc"], rec["index"]), "accept bits differ"
        olml = float.fromhex(rec["log_ml"])
        assert abs(sw.log_ml() - olml) < 1e-9 * max(1.0, abs(olml))
        return {"accept_rate": rec["accept_rate"], "log_ml": sw.log_ml()}
    ref = oracle_nlssm_mh_sweep(n, T, seed)
    otr_x, olw, oacc, terms = ref["x"], ref["lw"], ref["acc"], ref["terms"]
    assert np.array_equal(x.cpu().numpy(), otr_x)
    assert np.array_equal(lw.cpu().numpy(), olw)
    assert np.array_equal(sw.accept.cpu().numpy(), oacc)
    assert abs(sw.log_ml() - sum(terms)) < 1e-9 * max(1.0, abs(sum(terms)))
    return {"accept_rate": float(oacc.mean()), "log_ml": sw.log_ml()}


# ---------------------------------------------------------------------------
# dirichlet (tfp/__init__.py:125; used by the mixture model's weights)
# ---------------------------------------------------------------------------
def check_dirichlet(n=2000, seed=1):
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    al = np.array([0.5, 2.0, 3.5, 1.0], np.float32)

    def mk(g, arr):
        @g.gen
        def m(scale):
            return g.dirichlet(arr(al) * scale) @ "probs"
        return m
    m, om = mk(G, jnp.array), mk(O, lambda a: np.asarray(a, np.float32))
    tr, otr = m.simulate(G.split(G.key(seed), n), (1.0,)), om.simulate(O.split(O.key(seed), n), (np.float32(1.0),))
    p = tr.get_choices()["probs"].cpu().numpy()
    assert p.shape == (n, 4) and np.array_equal(p, otr.get_choices()["probs"])
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
    assert np.abs(p.sum(1) - 1).max() < 1e-6
    assert np.abs(p.mean(0) - al / al.sum()).max() < 4 * np.sqrt(0.25 / n)            # moments of Dir(a)
    from scipy.stats import dirichlet as sd
    ref = np.array([sd.logpdf(p[i].astype(np.float64) / p[i].astype(np.float64).sum(), al) for i in range(64)])
    assert np.abs(tr.get_score().cpu().numpy()[:64] - ref).max() < 2e-5
    s, _ = m.assess(tr.get_choices(), (1.0,), batch_shape=(n,))
    assert np.array_equal(s.cpu().numpy(), tr.get_score().cpu().numpy())
    # importance with the value constrained: w = log density
    tr2, w2 = m.importance(G.split(G.key(seed + 1), n), G.ChoiceMap.kw(probs=tr.get_choices()["probs"]), (1.0,))
    assert np.array_equal(w2.cpu().numpy(), s.cpu().numpy())


# ---------------------------------------------------------------------------
# a vector-valued state through BootstrapSweep: 2-D constant-velocity tracker
# ---------------------------------------------------------------------------
def make_tracker(g, stack):
    @g.gen
    def init():
        p = g.normal(0.0, 1.0) @ "p"
        v = g.normal(0.0, 0.5) @ "v"
        g.normal(p, 0.3) @ "y"
        return stack(p, v)

    @g.gen
    def step(s):
        p = g.normal(s[..., 0] + 0.1 * s[..., 1], 0.05) @ "p"
        v = g.normal(s[..., 1], 0.1) @ "v"
        g.normal(p, 0.3) @ "y"
        return stack(p, v)
    return init, step


def tracker_data(T):
    return (0.1 * np.arange(T) + np.random.default_rng(0).normal(0, 0.3, T)).astype(np.float32)


def oracle_tracker_sweep(n, T, seed):
    oi, ost = make_tracker(O, lambda a, b: np.stack([a, b], axis=-1))
    return oracle_bootstrap_sweep(oi, ost, n, T, tracker_data(T), O.key(seed))


def check_vector_state_sweep(n=3000, T=6, seed=5, capture=False, specialize=False):
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from genjax_amd.inference import smc
    init, step = make_tracker(G, lambda a, b: jnp.stack([a, b]))
    oi, ost = make_tracker(O, lambda a, b: np.stack([a, b], axis=-1))
    ys = tracker_data(T)
    sw = smc.BootstrapSweep(init, step, n, T, specialize=specialize).prepare(G.key(seed), torch.from_numpy(ys))
    if capture:
        sw.capture()
    sw.launch()
    x, lw, anc = sw.state()
    ref = oracle_bootstrap_sweep(oi, ost, n, T, ys, O.key(seed))
    assert tuple(x.shape) == (n, 2)
    assert np.array_equal(x.cpu().numpy(), ref["x"])
    assert np.array_equal(anc.cpu().numpy(), ref["anc"])
    assert sw.log_ml() == ref["log_ml"]


# ---------------------------------------------------------------------------
# a VECTOR state through the fused MH sweep: 2-D state as one vector-valued site, Rejuvenate on it
def make_vec_mh(g, stack, ones2):
    """a D-vector state (D = len(ones2); `stack(a, b, ...)` joins D components) in ONE vector-valued site"""
    D = int(ones2.shape[-1])

    @g.gen
    def init():
        x = g.normal(0.0 * ones2, ones2) @ "x"
        g.normal(x[..., 0], 0.5) @ "y"
        return x

    @g.gen
    def step(xp, t):
        loc = stack(*[0.9 * xp[..., i] + 0.1 * xp[..., i + 1] for i in range(D - 1)], 0.8 * xp[..., D - 1])
        x = g.normal(loc, 0.3 * ones2) @ "x"
        g.normal(x[..., 0], 0.5) @ "y"
        return x
    return init, step


def oracle_mh_sweep(oi, ost, oreq, ys, n, T, seed, extra=lambda t: ()):
    """oracle_nlssm_mh_sweep for any (init, step, request): same key schedule"""
    okey = O.key(seed)
    otr = olw = oacc = None
    terms = []
    for t in range(T):
        oks = O.split(O.fold_in(okey, t), 3)
        oobs = O.C.kw(y=np.float32(ys[t]))
        if t == 0:
            otr, olw = oi.importance(O.split(oks[0], n), oobs, ())
        else:
            okr = O.split(O.fold_in(okey, t - 1), 3)[1]
            cdf, total, M, shift = O.weight_cdf(olw)
            terms.append(O.log_ml_increment(M, total, shift, n))
            otr = O.gather_trace(otr, O.ancestors(O.SYSTEMATIC, okr, cdf))
            gf = otr.get_gen_fn()
            otr, oacc, _ = O.rejuvenate(oks[2], otr, lambda k, tr_: gf.edit_static(k, tr_, oreq, tr_.get_args()))
            otr, olw = ost.importance(O.split(oks[0], n), oobs, (np.asarray(otr.get_retval(), np.float32),) + tuple(extra(t)))
    x = np.asarray(otr.get_retval(), np.float32)
    olw = np.asarray(olw, np.float32)
    cdf, total, M, shift = O.weight_cdf(olw)
    terms.append(O.log_ml_increment(M, total, shift, n))
    anc = O.ancestors(O.SYSTEMATIC, O.split(O.fold_in(okey, T - 1), 3)[1], cdf)
    return {"x": x, "lw": olw, "acc": oacc, "terms": terms, "anc": anc}


def check_vector_mh_sweep(n=1500, T=5, seed=11, capture=False, specialize=False, chain_mh=True):
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from genjax_amd.inference import smc
    init, step = make_vec_mh(G, lambda *v: jnp.stack(list(v)), jnp.ones(2))
    oi, ost = make_vec_mh(O, lambda *v: np.stack(v, axis=-1), np.ones(2, np.float32))
    ys = tracker_data(T)
    req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.2))})
    oreq = {"x": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.2)))}
    sw = smc.BootstrapSweep(init, step, n, T, step_extra=lambda t: (float(t),), rejuvenate=req,
                            specialize=specialize, chain_mh=chain_mh).prepare(G.key(seed), torch.from_numpy(ys))
    if capture:
        sw.capture()
    sw.launch()
    x, lw, anc = sw.state()
    ref = oracle_mh_sweep(oi, ost, oreq, ys, n, T, seed, extra=lambda t: (np.float32(t),))
    assert tuple(x.shape) == (n, 2)
    assert np.array_equal(x.cpu().numpy(), ref["x"])
    assert np.array_equal(lw.cpu().numpy(), ref["lw"])
    assert np.array_equal(anc.cpu().numpy(), ref["anc"])
    assert np.array_equal(sw.accept.cpu().numpy(), ref["acc"])
    assert abs(sw.log_ml() - sum(ref["terms"])) < 1e-9 * max(1.0, abs(sum(ref["terms"])))
    return {"accept_rate": float(ref["acc"].mean())}


def check_scan_long_vector_site(n=130, T=40, seed=3):
    """A long scan (counted loop) whose kernel has a VECTOR-valued site: a 2-D latent state x_t ~ N(A x_{t-1}, 0.5) held
    in one site (one site key, element counters — App. A.3), scalar observations.  Values come back as [n, T, 2];
    simulate / importance / assess equal the oracle's step-by-step statement."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    ys = np.random.default_rng(seed).normal(size=T).astype(np.float32)

    def mk(g, np_):
        @g.gen
        def step(x, t):
            m = np_.stack([0.9 * x[..., 0] + 0.1 * x[..., 1], 0.8 * x[..., 1]], axis=-1)
            xn = g.normal(m, np_.array([0.5, 0.3], dtype=np.float32) if np_ is np else jnp.array([0.5, 0.3])) @ "x"
            g.normal(xn[..., 0] + xn[..., 1], 1.0) @ "y"
            return xn, xn[..., 0]
        return step
    step, ostep = mk(G, jnp), mk(O, np)
    dev = G._lib.get().device
    x0 = torch.zeros((n, 2), device=dev)
    sc, osc = step.scan(n=T), O.Scan(ostep, T)
    ts = jnp.zeros(T)
    tr = sc.simulate(G.split(G.key(seed), n), (x0, ts))
    otr = osc.simulate(O.split(O.key(seed), n), (np.zeros((n, 2), np.float32), np.zeros(T, np.float32)))
    xs = tr.get_choices()["x"]
    assert tuple(xs.shape) == (n, T, 2)
    assert np.array_equal(xs.cpu().numpy(), otr.get_choices()["x"])
    assert np.array_equal(tr.get_choices()["y"].cpu().numpy(), otr.get_choices()["y"])
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
    carry, first = tr.get_retval()
    ocarry, ofirst = otr.get_retval()
    assert np.array_equal(carry.cpu().numpy(), ocarry) and np.array_equal(first.cpu().numpy(), ofirst)
    con = C[:, "y"].set(ys)
    ocon = O.C.d({("y",): ys})
    tr2, w = sc.importance(G.split(G.key(seed + 1), n), con, (x0, ts))
    otr2, ow = osc.importance(O.split(O.key(seed + 1), n), ocon, (np.zeros((n, 2), np.float32), np.zeros(T, np.float32)))
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(tr2.get_choices()["x"].cpu().numpy(), otr2.get_choices()["x"])


def check_scan_long_edits(n=150, T=40, seed=9):
    """Update (new observations at every step) and Regenerate (the latent path) of a LONG scan — the counted-loop form of
    Scan.edit (scan.py:417-594): chained keys, carries threaded through the edited predecessors, weights / scores /
    discards equal to the oracle's step-by-step statement."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    ys = np.random.default_rng(seed).normal(size=T).astype(np.float32)

    def mk(g):
        @g.gen
        def step(x, t):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, xn * 2.0
        return step
    step, ostep = mk(G), mk(O)
    dev = G._lib.get().device
    sc, osc = step.scan(n=T), O.Scan(ostep, T)
    a = (torch.zeros(n, device=dev), jnp.zeros(T))
    oa = (np.zeros(n, np.float32), np.zeros(T, np.float32))
    t0, ot0 = sc.simulate(G.split(G.key(seed), n), a), osc.simulate(O.split(O.key(seed), n), oa)
    u, w, _, bwd = Update(C[:, "y"].set(ys)).edit(G.split(G.key(seed + 1), n), t0, Diff.no_change(a))
    ou, ow = O.scan_edit(osc, O.split(O.key(seed + 1), n), ot0, oa, update=O.C.d({"y": ys}))
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(u.get_score().cpu().numpy(), ou.get_score())
    assert np.array_equal(u.get_choices()["y"].cpu().numpy(), np.broadcast_to(ys, (n, T)))
    assert np.array_equal(u.get_choices()["x"].cpu().numpy(), ot0.get_choices()["x"])           # latents untouched
    assert np.array_equal(bwd.constraint["y"].cpu().numpy(), ot0.get_choices()["y"])            # discard = old values
    r, wr, _, _ = Regenerate(S["x"]).edit(G.split(G.key(seed + 2), n), u, Diff.no_change(a))
    orr, owr = O.scan_edit(osc, O.split(O.key(seed + 2), n), ou, oa, regenerate=O.selection("x"))
    assert np.array_equal(r.get_choices()["x"].cpu().numpy(), orr.get_choices()["x"])
    assert np.array_equal(wr.cpu().numpy(), owr)
    assert np.array_equal(r.get_score().cpu().numpy(), orr.get_score())
    carry, doubled = r.get_retval()
    ocarry, odoubled = orr.get_retval()
    assert np.array_equal(carry.cpu().numpy(), ocarry) and np.array_equal(doubled.cpu().numpy(), odoubled)
    # IndexRequest (scan.py:325-416) in the loop: one step edited with the caller's key, the next re-scored against the
    # new carry; a Python-int idx and one idx per particle
    from genjax_amd import IndexRequest
    regen = lambda k, sl, a_: ostep.regenerate(k, sl, O.selection("x"), a_)[:2]
    for idx in (2, T - 1):
        e, we, _, be = IndexRequest(idx, Regenerate(S["x"])).edit(G.split(G.key(seed + 3), n), r, Diff.no_change(a))
        oe, owe = O.scan_edit_index(osc, O.split(O.key(seed + 3), n), orr, oa, idx, regen)
        xe = e.get_choices()["x"].cpu().numpy()
        assert np.array_equal(xe, oe.get_choices()["x"]) and np.array_equal(we.cpu().numpy(), owe)
        assert np.array_equal(e.get_score().cpu().numpy(), oe.get_score())
        old = r.get_choices()["x"].cpu().numpy()
        assert np.array_equal(np.delete(old, idx, axis=1), np.delete(xe, idx, axis=1))      # only step idx moved
    idx = np.random.default_rng(seed + 9).integers(0, T, n).astype(np.int32)
    e6, we6, _, _ = IndexRequest(torch.from_numpy(idx).to(dev), Regenerate(S["x"])).edit(
        G.split(G.key(seed + 4), n), r, Diff.no_change(a))
    xo6 = orr.get_choices()["x"].copy()
    wo6, so6 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t_ in range(T):
        m_ = idx == t_
        if not m_.any():
            continue
        cand, wt = O.scan_edit_index(osc, O.split(O.key(seed + 4), n), orr, oa, t_, regen)
        xo6[m_] = cand.get_choices()["x"][m_]
        wo6[m_], so6[m_] = np.asarray(wt, np.float32)[m_], np.asarray(cand.get_score(), np.float32)[m_]
    assert np.array_equal(e6.get_choices()["x"].cpu().numpy(), xo6)
    assert np.array_equal(we6.cpu().numpy(), wo6)
    assert np.array_equal(e6.get_score().cpu().numpy(), so6)


def check_scan_long_vector_constraints(n=140, T=30, seed=6):
    """Long scan, 2-D latent state in one site AND 2-D observations in one site: per-step constraints on a vector-valued
    site from a launch-uniform [T, 2] table (importance), from per-particle [n, T, 2] choices (assess), and the previous
    values of vector-valued sites under Update / Regenerate."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    ys = np.random.default_rng(seed).normal(size=(T, 2)).astype(np.float32)

    def mk(g, np_):
        @g.gen
        def step(x, t):
            m = np_.stack([0.9 * x[..., 0] + 0.1 * x[..., 1], 0.8 * x[..., 1]], axis=-1)
            xn = g.normal(m, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, xn[..., 1]
        return step
    step, ostep = mk(G, jnp), mk(O, np)
    dev = G._lib.get().device
    sc, osc = step.scan(n=T), O.Scan(ostep, T)
    a = (torch.zeros((n, 2), device=dev), jnp.zeros(T))
    oa = (np.zeros((n, 2), np.float32), np.zeros(T, np.float32))
    tr, w = sc.importance(G.split(G.key(seed), n), C[:, "y"].set(jnp.array(ys)), a)
    otr, ow = osc.importance(O.split(O.key(seed), n), O.C.d({("y",): ys}), oa)
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(tr.get_choices()["x"].cpu().numpy(), otr.get_choices()["x"])
    assert np.array_equal(tr.get_choices()["y"].cpu().numpy(), np.broadcast_to(ys, (n, T, 2)))
    s, _ = sc.assess(tr.get_choices(), a)                              # per-particle [n, T, 2] choices read row by row
    so, _ = osc.assess(otr.get_choices(), oa, (n,))
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(s.cpu().numpy(), tr.get_score().cpu().numpy())
    t0, ot0 = sc.simulate(G.split(G.key(seed + 1), n), a), osc.simulate(O.split(O.key(seed + 1), n), oa)
    u, wu, _, bwd = Update(C[:, "y"].set(jnp.array(ys))).edit(G.split(G.key(seed + 2), n), t0, Diff.no_change(a))
    ou, owu = O.scan_edit(osc, O.split(O.key(seed + 2), n), ot0, oa, update=O.C.d({"y": ys}))
    assert np.array_equal(wu.cpu().numpy(), owu) and np.array_equal(u.get_score().cpu().numpy(), ou.get_score())
    assert np.array_equal(bwd.constraint["y"].cpu().numpy(), ot0.get_choices()["y"])
    r, wr, _, _ = Regenerate(S["x"]).edit(G.split(G.key(seed + 3), n), u, Diff.no_change(a))
    orr, owr = O.scan_edit(osc, O.split(O.key(seed + 3), n), ou, oa, regenerate=O.selection("x"))
    assert np.array_equal(r.get_choices()["x"].cpu().numpy(), orr.get_choices()["x"])
    assert np.array_equal(wr.cpu().numpy(), owr) and np.array_equal(r.get_score().cpu().numpy(), orr.get_score())


def check_scan_carry_forms(n=130, seed=21, Ts=(8, 17, 40)):
    """Loop-carried values that FORWARD one another: a shift register `(xn, a)` from `(a, b)`
    (AR(2)), and a swap `(b, a)`.  In the counted-loop form (T > 16) the carry update must be a parallel copy —
    an in-order MOV sequence reads a register it has already overwritten.  simulate / generate / Update / Regenerate
    against the oracle for an unrolled T (8) and two looped ones (17, 40)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    dev = G._lib.get().device

    def mk(g):
        @g.gen
        def ar2(carry, t):
            a, b = carry
            xn = g.normal(0.6 * a + 0.3 * b, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return (xn, a), xn - b

        @g.gen
        def swap(carry, t):
            a, b = carry
            z = g.normal(a - b, 1.0) @ "x"
            g.normal(z + b, 1.0) @ "y"
            return (b, a + 1.0), z
        return ar2, swap
    kernels, okernels = mk(G), mk(O)
    for T in Ts:
        ys = np.linspace(-1.0, 1.0, T).astype(np.float32)
        for kern, okern in zip(kernels, okernels):
            sc, osc = kern.scan(n=T), O.Scan(okern, T)
            a = ((torch.full((n,), 0.5, device=dev), torch.full((n,), -0.25, device=dev)), jnp.zeros(T))
            oa = ((np.full(n, 0.5, np.float32), np.full(n, -0.25, np.float32)), np.zeros(T, np.float32))
            tr, otr = sc.simulate(G.split(G.key(seed), n), a), osc.simulate(O.split(O.key(seed), n), oa)
            assert np.array_equal(tr.get_choices()["x"].cpu().numpy(), otr.get_choices()["x"]), (T, kern)
            (c0, c1), out = tr.get_retval()
            (oc0, oc1), oout = otr.get_retval()
            assert np.array_equal(c0.cpu().numpy(), oc0) and np.array_equal(c1.cpu().numpy(), oc1)
            assert np.array_equal(out.cpu().numpy(), oout)
            assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
            tg, wg = sc.importance(G.split(G.key(seed + 1), n), C[:, "y"].set(ys), a)
            otg, owg = osc.importance(O.split(O.key(seed + 1), n), O.C.d({"y": ys}), oa)
            assert np.array_equal(tg.get_choices()["x"].cpu().numpy(), otg.get_choices()["x"])
            assert np.array_equal(wg.cpu().numpy(), owg)
            u, wu, _, _ = Update(C["y"].set(-ys)).edit(G.split(G.key(seed + 2), n), tg, Diff.no_change(a))
            ou, owu = O.scan_edit(osc, O.split(O.key(seed + 2), n), otg, oa, update=O.C.d({"y": -ys}))
            assert np.array_equal(wu.cpu().numpy(), owu)
            assert np.array_equal(u.get_score().cpu().numpy(), ou.get_score())
            r, wr, _, _ = Regenerate(S["x"]).edit(G.split(G.key(seed + 3), n), u, Diff.no_change(a))
            orr, owr = O.scan_edit(osc, O.split(O.key(seed + 3), n), ou, oa, regenerate=O.selection("x"))
            assert np.array_equal(r.get_choices()["x"].cpu().numpy(), orr.get_choices()["x"])
            assert np.array_equal(wr.cpu().numpy(), owr)
            (r0, r1), _ = r.get_retval()
            (or0, or1), _ = orr.get_retval()
            assert np.array_equal(r0.cpu().numpy(), or0) and np.array_equal(r1.cpu().numpy(), or1)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def check_plate_on_the_launch_axis(n=10_000, seed=21, compare_batch_form=False):
    """A LARGE plate under ONE key runs with its ELEMENTS on the launch axis (ref vmap.py:180-218 is jax.vmap: a plate
    is as parallel as a particle batch): `model.vmap()` over n datapoints called directly —
    simulate / importance / assess / Update.  Every choice and every per-element score equals the oracle's Vmap bit for
    bit (element j's key is split(key, n)[j]); the plate's score / weight is the fixed-tree sum (gmx_sum_rows), which
    the oracle restates (plate_sum_tree): equal bit for bit too.  A bare distribution under vmap likewise.
    compare_batch_form: returns (seconds of the plate form, seconds of the same model run with the datapoints as the
    particle batch) for the GPU test's 1.5x bound."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, combinators
    dev = G._lib.get().device
    assert n >= combinators.VMAP_LAUNCH_MIN == O.Vmap.LAUNCH_MIN
    rng = np.random.default_rng(seed)
    mus = rng.normal(0, 2, n).astype(np.float32)
    ys = rng.normal(0, 3, n).astype(np.float32)

    def mk(g):
        @g.gen
        def pt(mu, s):
            z = g.normal(mu, s) @ "z"
            y = g.normal(z, 1.0) @ "y"
            return z + y
        return pt
    pt, opt = mk(G), mk(O)
    v, ov = pt.vmap(in_axes=(0, None)), O.Vmap(opt, in_axes=(0, None))
    args, oargs = (_t(mus, dev), 2.0), (mus, np.float32(2.0))
    tr, otr = v.simulate(G.key(seed), args), ov.simulate(O.key(seed), oargs)
    assert tuple(tr.batch_shape) == () and tuple(tr.get_choices()["z"].shape) == (n,)
    for a in ("z", "y"):
        assert np.array_equal(tr.get_choices()[a].cpu().numpy(), otr.get_choices()[a]), a
    assert np.array_equal(tr.get_retval().cpu().numpy(), otr.get_retval())
    assert float(tr.get_score()) == float(otr.get_score())
    assert abs(float(tr.get_score()) - float(np.sum(np.asarray(otr.inner.get_score(), np.float64)))) <= 2e-6 * abs(float(otr.get_score()))
    tri, w = v.importance(G.key(seed + 1), C["y"].set(_t(ys, dev)), args)
    otri, ow = ov.importance(O.key(seed + 1), O.C.d({"y": ys}), oargs)
    assert np.array_equal(tri.get_choices()["z"].cpu().numpy(), otri.get_choices()["z"])
    assert float(w) == float(ow) and float(tri.get_score()) == float(otri.get_score())
    s, r = v.assess(tri.get_choices(), args)
    so, ro = ov.assess(otri.get_choices(), oargs, ())
    assert float(s) == float(so) == float(tri.get_score()) and np.array_equal(r.cpu().numpy(), ro)
    zs = rng.normal(0, 1, n).astype(np.float32)
    new_tr, wu, _, bwd = Update(C["z"].set(_t(zs, dev))).edit(G.key(seed + 2), tri, Diff.no_change(args))
    onew, owu, odisc = O.vmap_update(ov, O.key(seed + 2), otri, O.C.d({"z": zs}), oargs)
    assert float(wu) == float(owu) and float(new_tr.get_score()) == float(onew.get_score())
    assert np.array_equal(bwd.constraint["z"].cpu().numpy(), odisc["z"])
    assert np.array_equal(new_tr.get_choices()["z"].cpu().numpy(), zs)
    # a bare distribution under vmap: one vector-valued site, keys split(key, n)[j]
    b, ob = G.normal.vmap(in_axes=(0, None)), O.Vmap(O.normal, in_axes=(0, None))
    tb, otb = b.simulate(G.key(seed + 3), (_t(mus, dev), 0.5)), ob.simulate(O.key(seed + 3), (mus, np.float32(0.5)))
    assert np.array_equal(tb.get_choices().get_value().cpu().numpy(), np.asarray(otb.get_choices().get_value()))
    assert float(tb.get_score()) == float(otb.get_score())
    if not compare_batch_form:
        return None
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)

    def timed(fn, reps=5):
        fn(); sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        return (time.perf_counter() - t0) / reps
    con = C["y"].set(_t(ys, dev))
    t_plate = timed(lambda: v.importance(G.key(seed + 1), con, args))
    keys = G.split(G.key(seed + 1), n)
    t_batch = timed(lambda: pt.importance(keys, con, args))
    return t_plate, t_batch


def check_plates_long(n=130, P=40, seed=8, light=False):
    """A LARGE plate (more than 16 elements) runs as a counted loop in the site program (ref vmap.py:180-218: `jax.vmap`
    over any n): the P-schools model written with `Vmap` — simulate / importance with a per-element
    constraint (`C["schools", :, "y"]`) / assess, the plate used directly, `Update` of every element, `IndexRequest`
    with a Python-int index and with one index per particle, a bare distribution under vmap, `repeat(n=P)` —
    bit-exact against the oracle's per-element restatement."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    dev = G._lib.get().device
    sig = np.linspace(1.0, 3.0, P).astype(np.float32)
    ys = np.linspace(-2.0, 2.0, P).astype(np.float32)
    school, oschool = _school(G), _school(O)

    @G.gen
    def schools():
        mu = G.normal(0.0, 5.0) @ "mu"
        return school.vmap(in_axes=(None, None, 0))(mu, 2.0, jnp.array(sig)) @ "schools"

    @O.gen
    def o_schools():
        mu = O.normal(0.0, 5.0) @ "mu"
        return O.Vmap(oschool, in_axes=(None, None, 0))(mu, 2.0, sig) @ "schools"
    tr, otr = schools.simulate(G.split(G.key(seed), n), ()), o_schools.simulate(O.split(O.key(seed), n), ())
    th = tr.get_choices()["schools", "theta"]
    assert tuple(th.shape) == (n, P)
    assert np.array_equal(th.cpu().numpy(), otr.get_choices()["schools", "theta"])          # keys split(key, P)[j]
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
    assert np.array_equal(tr.get_retval().cpu().numpy(), otr.get_retval())
    tri, w = schools.importance(G.split(G.key(seed + 1), n), C["schools", :, "y"].set(ys), ())
    otri, ow = o_schools.importance(O.split(O.key(seed + 1), n), O.C.d({("schools", "y"): ys}), ())
    assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(tri.get_score().cpu().numpy(), otri.get_score())
    assert np.array_equal(tri.get_choices()["schools", "y"].cpu().numpy(), np.broadcast_to(ys, (n, P)))
    s, _ = schools.assess(tri.get_choices(), ())
    so, _ = o_schools.assess(otri.get_choices(), (), (n,))
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(s.cpu().numpy(), tri.get_score().cpu().numpy())
    # constraints on single elements through integer addresses: masked constraints inside the loop
    trj, wj = schools.importance(G.split(G.key(seed + 2), n), C["schools", 3, "y"].set(0.5).set(("schools", P - 2, "y"), -1.0), ())
    yj, tj = trj.get_choices()["schools", "y"].cpu().numpy(), trj.get_choices()["schools", "theta"].cpu().numpy()
    assert np.all(yj[:, 3] == np.float32(0.5)) and np.all(yj[:, P - 2] == np.float32(-1.0)) and not np.all(yj[:, 4] == np.float32(0.5))
    w_ref = (O.normal.assess(O.C.choice(np.full(n, 0.5, np.float32)), (tj[:, 3], sig[3]), (n,))[0]
             + O.normal.assess(O.C.choice(np.full(n, -1.0, np.float32)), (tj[:, P - 2], sig[P - 2]), (n,))[0])
    assert np.array_equal(wj.cpu().numpy(), w_ref)
    if light:            # (full-size plates: the oracle's per-element loops below would take minutes)
        return
    # the plate used directly; edits of it as a loop
    v, ov = school.vmap(in_axes=(None, None, 0)), O.Vmap(oschool, in_axes=(None, None, 0))
    args = (1.0, 2.0, jnp.array(sig))
    oargs = (np.float32(1.0), np.float32(2.0), sig)
    t2, ot2 = v.simulate(G.split(G.key(seed + 3), n), args), ov.simulate(O.split(O.key(seed + 3), n), oargs)
    assert np.array_equal(t2.get_choices()["theta"].cpu().numpy(), ot2.get_choices()["theta"])
    new_tr, wu, _, bwd = Update(C["y"].set(ys)).edit(G.split(G.key(seed + 4), n), t2, Diff.no_change(args))
    onew, owu, odisc = O.vmap_update(ov, O.split(O.key(seed + 4), n), ot2, O.C.d({"y": ys}), oargs)
    assert np.array_equal(wu.cpu().numpy(), owu) and np.array_equal(new_tr.get_score().cpu().numpy(), onew.get_score())
    assert np.array_equal(bwd.constraint["y"].cpu().numpy(), odisc["y"])
    assert np.array_equal(new_tr.get_choices()["theta"].cpu().numpy(), ot2.get_choices()["theta"])
    j0 = P - 5
    t3, w3, _, bwd3 = IndexRequest(j0, Regenerate(S["theta"])).edit(G.split(G.key(seed + 5), n), new_tr, Diff.no_change(args))
    aj = (np.float32(1.0), np.float32(2.0), np.float32(sig[j0]))
    ot3, ow3 = O.vmap_edit_index(ov, O.split(O.key(seed + 5), n), onew, j0,
                                 lambda k, sl, a: oschool.regenerate(k, sl, O.selection("theta"), a)[:2], aj)
    th3 = t3.get_choices()["theta"].cpu().numpy()
    assert np.array_equal(th3, ot3.get_choices()["theta"]) and np.array_equal(w3.cpu().numpy(), ow3)
    assert np.array_equal(t3.get_score().cpu().numpy(), ot3.get_score())
    th1 = new_tr.get_choices()["theta"].cpu().numpy()
    assert np.array_equal(np.delete(th1, j0, axis=1), np.delete(th3, j0, axis=1)) and not np.array_equal(th1[:, j0], th3[:, j0])
    assert isinstance(bwd3, IndexRequest) and bwd3.idx == j0
    idx = np.random.default_rng(seed).integers(0, P, n).astype(np.int32)
    t4, w4, _, _ = IndexRequest(torch.from_numpy(idx).to(dev), Regenerate(S["theta"])).edit(
        G.split(G.key(seed + 6), n), t2, Diff.no_change(args))
    tho, wo = ot2.get_choices()["theta"].copy(), np.zeros(n, np.float32)
    for j in np.unique(idx):
        a_ = (np.float32(1.0), np.float32(2.0), np.float32(sig[j]))
        cand, wj_ = O.vmap_edit_index(ov, O.split(O.key(seed + 6), n), ot2, int(j),
                                      lambda k, sl, a: oschool.regenerate(k, sl, O.selection("theta"), a)[:2], a_)
        m_ = idx == j
        tho[m_, j], wo[m_] = cand.get_choices()["theta"][m_, j], np.asarray(wj_, np.float32)[m_]
    assert np.array_equal(t4.get_choices()["theta"].cpu().numpy(), tho) and np.array_equal(w4.cpu().numpy(), wo)
    # a bare distribution under vmap (one vector-valued site with split keys) and repeat(n=P)
    locs = np.linspace(-1.0, 1.0, P).astype(np.float32)

    @G.gen
    def bare():
        xs = G.normal.vmap(in_axes=(0, None))(jnp.array(locs), 1.0) @ "xs"
        zs = school.repeat(n=P)(0.5, 1.0, 2.0) @ "zs"
        return xs

    @O.gen
    def o_bare():
        xs = O.Vmap(O.normal, in_axes=(0, None))(np.broadcast_to(locs, (n, P)), 1.0) @ "xs"      # [batch, plate]
        zs = O.Repeat(oschool, P)(0.5, 1.0, 2.0) @ "zs"
        return xs
    tb, otb = bare.simulate(G.split(G.key(seed + 7), n), ()), o_bare.simulate(O.split(O.key(seed + 7), n), ())
    assert np.array_equal(tb.get_choices()["xs"].cpu().numpy(), otb.get_choices()["xs"])
    assert np.array_equal(tb.get_choices()["zs", "theta"].cpu().numpy(), otb.get_choices()["zs", "theta"])
    assert np.array_equal(tb.get_score().cpu().numpy(), otb.get_score())
    assert np.array_equal(tb.get_retval().cpu().numpy(), otb.get_retval())


def check_batched_csmc(k=33, B=1000, seed=11):
    """VERDICT r2 item 7 (ref smc.py:317-351, 398-465; sp.py:217-240): conditional SMC under a BATCH of keys is one
    launch set over [keys, K] with the retained particle in slot K-1 of every row — `vmap(alg.estimate_logpdf)` over B
    keys, `run_csmc`, `estimate_reciprocal_normalizing_constant` and a nested `Marginal(algorithm=...)` — and equals
    the per-key runs (the host walk of round 2) bit for bit in weights, choices and sampled particles (the estimates up
    to the tree-vs-row order of the log-sum-exp: 1e-5)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Target
    from genjax_amd.inference.smc import ChangeTarget, ImportanceK
    from genjax_amd.inference.sp import Marginal
    dev = G._lib.get().device

    @G.gen
    def model():
        p = G.beta(2.0, 2.0) @ "p"
        v = G.flip(p) @ "v"
        G.normal(p, 1.0) @ "z"
        return v
    tgt = Target(model, (), C["v"].set(True))
    alg = ImportanceK(tgt, k_particles=k)
    keys = G.split(G.key(seed), B)
    rng = np.random.default_rng(seed)
    ps = torch.from_numpy(rng.uniform(0.1, 0.9, B).astype(np.float32)).to(dev)
    zs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32)).to(dev)
    ret = C.d({"p": ps, "z": zs})
    pc = alg.run_csmc(keys, ret)
    lw = pc.get_log_weights()
    assert tuple(lw.shape) == (B, k)
    ch = pc.get_particles().get_choices()
    assert torch.equal(ch["p"][:, -1], ps) and torch.equal(ch["z"][:, -1], zs)        # slot K-1 of every row
    est = alg.estimate_logpdf(keys, ret, tgt)
    ct = ChangeTarget(alg, tgt)
    lw2 = ct.run_csmc(keys, ret).get_log_weights()
    wz = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32)).to(dev)
    rz = alg.estimate_reciprocal_normalizing_constant(keys, tgt, ret, wz)
    assert tuple(est.shape) == (B,) and tuple(rz.shape) == (B,)
    for i in list(range(0, B, max(1, B // 7))) + [B - 1]:
        ri = C.d({"p": float(ps[i]), "z": float(zs[i])})
        pci = alg.run_csmc(keys[i], ri)
        assert torch.equal(pci.get_log_weights(), lw[i]), i
        assert torch.equal(pci.get_particles().get_choices()["p"], ch["p"][i])
        assert torch.equal(ct.run_csmc(keys[i], ri).get_log_weights(), lw2[i])
        assert abs(float(alg.estimate_logpdf(keys[i], ri, tgt)) - float(est[i])) <= 1e-5
        assert abs(float(alg.estimate_reciprocal_normalizing_constant(keys[i], tgt, ri, float(wz[i]))) - float(rz[i])) <= 1e-5
    # a Marginal with an inner algorithm under a batch of keys: on the device, equal to the per-key host walk
    marg = Marginal(model, G.SelectionBuilder["p"], ImportanceK(Target(model, (), C.n()), k_particles=k))
    nb = min(B, 64)
    w_dev, chm_dev = marg.random_weighted(keys[:nb])
    for i in range(nb):                      # the per-key walk: every key its own conditional SMC
        w_i, chm_i = marg.random_weighted(keys[i])
        assert torch.equal(chm_dev["p"][i].cpu(), torch.as_tensor(chm_i["p"]).cpu()), i
        assert abs(float(w_dev[i]) - float(w_i)) <= 1e-5, i


def check_runtime_indexed(n=257, seed=13):
    """Run-time `Indexed` addresses (ref choice_map.py:1453-1531 with a traced index): one plate
    index PER PARTICLE — `C["schools", dynamic_index(idx), "y"].set(v)`, and the reference's plain spelling
    `C["schools", idx, "y"].set(v)` under `genjax.vmap`.  Inside the plate it is `Mask(v, idx == j)` at element j:
    equal to writing those masks out by hand (small plate), and — for a plate run as a counted loop — the weight is
    the density of v at the indexed element, the other elements are sampled as without the constraint."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Mask, numpy as jnp
    dev = G._lib.get().device
    school = _school(G)
    rng = np.random.default_rng(seed)
    for P in (8, 40):
        sig = np.linspace(1.0, 3.0, P).astype(np.float32)

        @G.gen
        def m():
            mu = G.normal(0.0, 5.0) @ "mu"
            return school.vmap(in_axes=(None, None, 0))(mu, 2.0, jnp.array(sig)) @ "schools"
        idx_h = rng.integers(0, P, n).astype(np.int32)
        vals_h = rng.normal(0, 1, n).astype(np.float32)
        idx, vals = torch.from_numpy(idx_h).to(dev), torch.from_numpy(vals_h).to(dev)
        keys = G.split(G.key(seed), n)
        tr, w = m.importance(keys, C["schools", G.dynamic_index(idx), "y"].set(vals), ())
        y = tr.get_choices()["schools", "y"].cpu().numpy()
        th = tr.get_choices()["schools", "theta"].cpu().numpy()
        rows = np.arange(n)
        assert np.array_equal(y[rows, idx_h], vals_h)                              # the indexed element took the value
        w_ref = O.normal.assess(O.C.choice(vals_h), (th[rows, idx_h], sig[idx_h]), (n,))[0]
        assert np.array_equal(w.cpu().numpy(), w_ref)                              # and the weight is its density alone
        free = m.simulate(keys, ())                                                # same keys, no constraint
        yf = free.get_choices()["schools", "y"].cpu().numpy()
        mask = np.ones((n, P), bool)
        mask[rows, idx_h] = False
        assert np.array_equal(y[mask], yf[mask])                                   # the other elements: as sampled
        w3 = G.vmap(lambda k, i, v: m.importance(k, C["schools", i, "y"].set(v), ())[1])(keys, idx, vals)
        assert torch.equal(w3, w)                                                  # the reference's spelling, vmapped
        if P <= 16:
            con = C.n()
            for j in range(P):
                con = con.set(("schools", j, "y"), Mask(vals, idx == j))
            tr2, w2 = m.importance(keys, con, ())
            assert torch.equal(w, w2) and torch.equal(tr.get_choices()["schools", "y"], tr2.get_choices()["schools", "y"])


def check_plate_of_scans(n=130, no=5, T=40, seed=17):
    """A plate of time series — `series.vmap()` where every element runs a LONG scan (ref: vmap.py:180-218 over
    scan.py:200-294; the reference nests combinators freely) — simulate / importance with one observation table per
    series (`C["series", :, "steps", :, "y"]`-shaped constraints: plate axis first, then the steps) / assess, bit-exact
    against the oracle: element j's scan runs under key split(key, no)[j], its steps chain fold_in(., t)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    rng = np.random.default_rng(seed)
    x0s = rng.normal(0, 1, no).astype(np.float32)
    sig = np.linspace(0.5, 1.5, no).astype(np.float32)
    ys = rng.normal(0, 1, (no, T)).astype(np.float32)

    def mk(g, scan_of):
        @g.gen
        def step(carry, _):
            x, s = carry
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, s) @ "y"
            return (xn, s), xn * 2.0

        @g.gen
        def series(x0, s, mu):
            (xT, _), doubled = scan_of(step)((x0 + mu, s), None) @ "steps"
            return xT, doubled
        return series
    series = mk(G, lambda st: st.scan(n=T))
    oseries = mk(O, lambda st: O.Scan(st, T))

    @G.gen
    def model():
        mu = G.normal(0.0, 1.0) @ "mu"
        return series.vmap(in_axes=(0, 0, None))(jnp.array(x0s), jnp.array(sig), mu) @ "series"

    @O.gen
    def o_model():
        mu = O.normal(0.0, 1.0) @ "mu"
        return O.Vmap(oseries, in_axes=(0, 0, None))(x0s, sig, mu) @ "series"
    tr, otr = model.simulate(G.split(G.key(seed), n), ()), o_model.simulate(O.split(O.key(seed), n), ())
    x = tr.get_choices()["series", "steps", "x"]
    assert tuple(x.shape) == (n, no, T)
    assert np.array_equal(x.cpu().numpy(), otr.get_choices()["series", "steps", "x"])
    assert np.array_equal(tr.get_choices()["series", "steps", "y"].cpu().numpy(), otr.get_choices()["series", "steps", "y"])
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())
    xT, dbl = tr.get_retval()
    oxT, odbl = otr.get_retval()
    assert np.array_equal(xT.cpu().numpy(), oxT) and np.array_equal(dbl.cpu().numpy(), odbl) and tuple(dbl.shape) == (n, no, T)
    tri, w = model.importance(G.split(G.key(seed + 1), n), C["series", "steps", "y"].set(ys), ())
    otri, ow = o_model.importance(O.split(O.key(seed + 1), n), O.C.d({("series", "steps", "y"): ys}), ())
    assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(tri.get_score().cpu().numpy(), otri.get_score())
    assert np.array_equal(tri.get_choices()["series", "steps", "x"].cpu().numpy(), otri.get_choices()["series", "steps", "x"])
    assert np.array_equal(tri.get_choices()["series", "steps", "y"].cpu().numpy(), np.broadcast_to(ys, (n, no, T)))
    s, _ = model.assess(tri.get_choices(), ())
    so, _ = o_model.assess(otri.get_choices(), (), (n,))
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(s.cpu().numpy(), tri.get_score().cpu().numpy())
    # ... and against scipy, which knows nothing of either implementation: the score is the sum of the sites' densities
    from scipy import stats
    ch = tr.get_choices()
    mu = ch["mu"].cpu().numpy().astype(np.float64)
    x = ch["series", "steps", "x"].cpu().numpy().astype(np.float64)
    y = ch["series", "steps", "y"].cpu().numpy().astype(np.float64)
    prev = np.concatenate([(x0s[None, :] + mu[:, None])[:, :, None], x[:, :, :-1]], axis=2)
    ref = (stats.norm.logpdf(mu, 0.0, 1.0) + stats.norm.logpdf(x, 0.9 * prev, 0.5).sum((1, 2))
           + stats.norm.logpdf(y, x, sig[None, :, None]).sum((1, 2)))
    assert np.allclose(tr.get_score().cpu().numpy(), ref, rtol=2e-5, atol=1e-3)
    return {"score_mean": float(tr.get_score().mean())}


# ---- nested combinators: two counted loops, one inside the other -------------------------------------------------
def check_nested_combinators(n=21):
    """Both levels long, in every combination the reference allows (its combinators nest freely): a plate of plates
    (20 x 30), a scan (40 steps) whose step runs a 30-element plate, a scan (30 steps) whose step runs a 40-step scan —
    simulate, bit-exact against the oracle: values shaped [n, outer, inner], keys derived level by level."""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    g_ = {"G": G, "jnp": jnp}
    _nest_plate_of_plates.__globals__.update(g_)
    _nest_plate_of_plates(n)
    _nest_scan_of_plate(n)
    _nest_scan_of_scan(n)


def _nest_cmp(tr, otr, addrs):
    for a in addrs:
        v = tr.get_choices()[a]; ov = otr.get_choices()[a]
        assert tuple(v.shape) == tuple(np.shape(ov)), (a, v.shape, np.shape(ov))
        assert np.array_equal(v.cpu().numpy(), ov), a
    assert np.array_equal(tr.get_score().cpu().numpy(), otr.get_score())

def _nest_plate_of_plates(n):
    A, B = 20, 30
    def mk(g, inner_v, outer_v):
        @g.gen
        def leaf(m):
            return g.normal(m, 1.0) @ "z"
        @g.gen
        def row(m):
            return inner_v(leaf)(m) @ "cols"
        return outer_v(row)
    mus = np.linspace(-1, 1, A * B).reshape(A, B).astype(np.float32)
    model = mk(G, lambda f: f.vmap(in_axes=(0,)), lambda f: f.vmap(in_axes=(0,)))
    omodel = mk(O, lambda f: O.Vmap(f, in_axes=(0,)), lambda f: O.Vmap(f, in_axes=(0,)))
    tr = model.simulate(G.split(G.key(3), n), (jnp.array(mus),))
    otr = omodel.simulate(O.split(O.key(3), n), (mus,))
    _nest_cmp(tr, otr, [("cols", "z")])

def _nest_scan_of_plate(n):
    T, B = 40, 30
    def mk(g, inner_v, scan_of):
        @g.gen
        def leaf(m):
            return g.normal(m, 1.0) @ "z"
        @g.gen
        def step(x, _):
            zs = inner_v(leaf)(x) @ "obs"
            xn = g.normal(0.9 * x, 0.5) @ "x"
            return xn, xn
        return scan_of(step)
    model = mk(G, lambda f: f.repeat(n=B), lambda f: f.scan(n=T))
    omodel = mk(O, lambda f: O.Repeat(f, B), lambda f: O.Scan(f, T))
    tr = model.simulate(G.split(G.key(4), n), (0.5, None))
    otr = omodel.simulate(O.split(O.key(4), n), (np.float32(0.5), None))
    _nest_cmp(tr, otr, ["x", ("obs", "z")])

def _nest_scan_of_scan(n):
    T, U = 30, 40
    def mk(g, scan_in, scan_out):
        @g.gen
        def inner(c, _):
            cn = g.normal(c, 0.1) @ "w"
            return cn, None
        @g.gen
        def step(x, _):
            xe, _ = scan_in(inner)(x, None) @ "fine"
            xn = g.normal(0.9 * xe, 0.5) @ "x"
            return xn, xn
        return scan_out(step)
    model = mk(G, lambda f: f.scan(n=U), lambda f: f.scan(n=T))
    omodel = mk(O, lambda f: O.Scan(f, U), lambda f: O.Scan(f, T))
    tr = model.simulate(G.split(G.key(5), n), (0.5, None))
    otr = omodel.simulate(O.split(O.key(5), n), (np.float32(0.5), None))
    _nest_cmp(tr, otr, ["x", ("fine", "w")])


def check_nested_constraint_forms(n=9, A=12, T=20):
    """Constraints addressed INTO two nested loops (a plate of long scans): one cell through two integer addresses
    (`C[3, "steps", 5, "y"]`: masked at (t_outer == 3) & (t_inner == 5)), one whole series through its integer address
    (`C[7, "steps", :, "y"]`: a masked [T] table read step by step), per-particle observations [n, A, T] (one
    step-indexed leaf, GMX_F_FLAT) — weights against the oracle's leaf densities, and assess of the resulting choices."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    @G.gen
    def step(x, _):
        xn = G.normal(0.9 * x, 0.5) @ "x"
        G.normal(xn, 1.0) @ "y"
        return xn, None
    @G.gen
    def series(x0):
        xT, _ = step.scan(n=T)(x0, None) @ "steps"
        return xT
    model = series.vmap(in_axes=(0,))
    x0s = np.linspace(-1, 1, A).astype(np.float32)
    tr, w = model.importance(G.split(G.key(2), n), C[3, "steps", 5, "y"].set(0.25), (jnp.array(x0s),))
    y = tr.get_choices()["steps", "y"].cpu().numpy(); x = tr.get_choices()["steps", "x"].cpu().numpy()
    assert y.shape == (n, A, T)
    assert np.all(y[:, 3, 5] == np.float32(0.25))
    assert (y == np.float32(0.25)).sum() == n
    ref = O.normal.assess(O.C.choice(np.full(n, 0.25, np.float32)), (x[:, 3, 5], np.float32(1.0)), (n,))[0]
    assert np.array_equal(w.cpu().numpy(), ref)
    # a whole series' observations through its integer address
    ys = np.linspace(-1, 1, T).astype(np.float32)
    tr2, w2 = model.importance(G.split(G.key(3), n), C[7, "steps", :, "y"].set(ys), (jnp.array(x0s),))
    y2 = tr2.get_choices()["steps", "y"].cpu().numpy(); x2 = tr2.get_choices()["steps", "x"].cpu().numpy()
    assert np.array_equal(y2[:, 7, :], np.broadcast_to(ys, (n, T)))
    ref2 = np.zeros(n, np.float32)
    for t in range(T):
        ref2 = (ref2 + O.normal.assess(O.C.choice(np.full(n, ys[t], np.float32)), (x2[:, 7, t], np.float32(1.0)), (n,))[0]).astype(np.float32)
    assert np.array_equal(w2.cpu().numpy(), ref2)
    # per-particle observations [n, A, T]
    yp = np.random.default_rng(0).normal(size=(n, A, T)).astype(np.float32)
    tr3, w3 = model.importance(G.split(G.key(4), n), C["steps", "y"].set(torch.from_numpy(yp)), (jnp.array(x0s),))
    assert np.array_equal(tr3.get_choices()["steps", "y"].cpu().numpy(), yp)
    s3, _ = model.assess(tr3.get_choices(), (jnp.array(x0s),))
    assert np.array_equal(s3.cpu().numpy(), tr3.get_score().cpu().numpy())


def check_multinomial_tiled(n=5000, seed=5, sigma=2.0, dead=False, spike=0.0):
    """gmx_multinomial_tiled (two-stage multinomial: tile by LDS histogram, then inside the tile) against the oracle's
    definition, directly through the C-ABI: ragged n, skewed weights (one particle with most of the mass: its tile owns
    almost every slot), no mass at all; with and without the stage-1 uniforms handed over (gmx_slot_uniforms)."""
    from ctypes import c_uint32
    from genjax_amd import _lib
    be = _lib.get()
    dev = be.device
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, sigma, n).astype(np.float32)
    if spike:
        lw[n // 3] += np.float32(spike)
    if dead:
        lw[:] = -np.inf
    k = O.key(seed + 1)
    cdf, total, M, shift = O.weight_cdf(lw)
    want = O.ancestors_multinomial_tiled(k, cdf)
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lw_d = T_(lw)
    tiles = (n + 1023) // 1024
    tmax = torch.zeros((tiles,), dtype=torch.float32, device=dev)
    agg = torch.zeros((tiles,), dtype=torch.int64, device=dev)
    be.check(be.c.gmx_tile_stats(be.ptr(lw_d), n, shift, be.ptr(tmax), be.ptr(agg), be.stream()), "gmx_tile_stats")
    kk = (c_uint32 * 2)(int(k[0]), int(k[1]))
    ws = torch.zeros(((int(be.c.gmx_multinomial_tiled_workspace(n)) + 3) // 4,), dtype=torch.int32, device=dev)
    k1 = O.split(k, 2)[0]
    for with_u in (False, True):
        u_d = None
        if with_u:
            keys = T_(np.asarray(k1, np.uint32).reshape(1, 2).view(np.int32))
            u_d = torch.zeros((n,), dtype=torch.int32, device=dev)
            be.check(be.c.gmx_slot_uniforms(be.ptr(keys), 1, n, be.ptr(u_d), 0, be.stream()), "gmx_slot_uniforms")
        mx = torch.zeros((1,), dtype=torch.float32, device=dev)
        tot = torch.zeros((1,), dtype=torch.int64, device=dev)
        anc = torch.full((n,), -1, dtype=torch.int32, device=dev)
        be.check(be.c.gmx_multinomial_tiled(kk, be.ptr(lw_d), n, shift, be.ptr(tmax), be.ptr(agg),
                                            be.ptr(u_d) if u_d is not None else None, be.ptr(mx), be.ptr(tot), be.ptr(anc),
                                            be.ptr(ws), -1 if not with_u else 1, be.stream()), "gmx_multinomial_tiled")
        got = anc.cpu().numpy()
        assert np.array_equal(got, want), (with_u, int((got != want).sum()))
        assert int(tot.item()) & 0xFFFFFFFFFFFFFFFF == total
        if not dead:
            assert float(mx.item()) == M
    return {"distinct": int(np.unique(want).size)}


def check_resample_beyond_2048_tiles(n=2049 * 1024 + 7, seed=3, sigma=2.0):
    """smc.resample_fused past the fused resampler's 2048 tiles (n > 2^21; BASELINE config 4's 1e7 is 9766 tiles): tile
    statistics -> gmx_tile_prefix (one workgroup, chunks of 2048 tiles with a running carry) -> gmx_resample_tiles_p,
    no CDF array — the same ancestors, total and max as gmx_weight_cdf + gmx_ancestors and as the oracle (C statement),
    systematic and stratified"""
    import genjax_amd as G
    from genjax_amd.inference import smc
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, sigma, n).astype(np.float32)
    cdf, total, M, shift = O.weight_cdf_c(lw)
    from genjax_amd import _lib
    lw_d = torch.from_numpy(lw).to(_lib.get().device)
    for kind, okind in ((smc.SYSTEMATIC, O.SYSTEMATIC), (smc.STRATIFIED, O.STRATIFIED)):
        anc, tot, mx, sh = smc.resample_fused(kind, G.key(seed + 1), lw_d)
        assert sh == shift and int(tot.item()) & 0xFFFFFFFFFFFFFFFF == total and float(mx.item()) == M
        want = O.ancestors_c(okind, O.key(seed + 1), cdf)
        assert np.array_equal(anc.cpu().numpy(), want), int((anc.cpu().numpy() != want).sum())
        cdf_d, tot2, mx2, _ = smc.weight_cdf(lw_d)
        anc2 = smc.ancestors_from_cdf(kind, G.key(seed + 1), cdf_d, tot2)
        assert torch.equal(anc, anc2) and torch.equal(tot, tot2)
    return True


def check_multinomial_sorted(n=5000, seed=5, sigma=2.0, dead=False, spike=0.0, rows=1):
    """gmx_resample_sorted (multinomial resampling with sorted uniforms on the ordered resampler's kernel) against the
    oracle's definition, directly through the C-ABI: ragged n, skewed weights, one particle with almost all the mass, no
    mass at all; the order-statistics table built inside the call, and handed over from gmx_sorted_uniforms (several
    keys in one 2-D launch) — whose words are checked against the oracle's statement of the table."""
    from ctypes import c_uint32
    from genjax_amd import _lib
    be = _lib.get()
    dev = be.device
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, sigma, n).astype(np.float32)
    if spike:
        lw[n // 3] += np.float32(spike)
    if dead:
        lw[:] = -np.inf
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lw_d = T_(lw)
    cdf, total, M, shift = O.weight_cdf(lw)
    tiles = (n + 1023) // 1024
    tmax = torch.zeros((tiles,), dtype=torch.float32, device=dev)
    agg = torch.zeros((tiles,), dtype=torch.int64, device=dev)
    be.check(be.c.gmx_tile_stats(be.ptr(lw_d), n, shift, be.ptr(tmax), be.ptr(agg), be.stream()), "gmx_tile_stats")
    words = int(be.c.gmx_sorted_uniforms_words(n))
    ks = [O.key(seed + 1 + r) for r in range(rows)]
    keys = T_(np.stack([np.asarray(k, np.uint32) for k in ks]).view(np.int32))
    tables = torch.full((rows, words), -1, dtype=torch.int32, device=dev)
    be.check(be.c.gmx_sorted_uniforms(be.ptr(keys), rows, n, be.ptr(tables), 0, be.stream()), "gmx_sorted_uniforms")
    out = {}
    for r, k in enumerate(ks):
        want = O.ancestors_of_kind(O.MULTINOMIAL_SORTED, k, cdf)
        assert np.all(np.diff(want) >= 0)
        # the table, word for word where it is defined
        t = O.sorted_uniforms_table(k, n)
        got_t = tables[r].cpu().numpy().view(np.uint32)
        og = t["tiles"] * 1024
        assert np.array_equal(got_t[:n], t["slow"]) and not got_t[n:og].any()
        assert np.array_equal(got_t[og:og + t["guide"].size], t["guide"])
        ng2 = (t["ng"] + 2 + 3) & ~3
        toff = got_t[og + ng2 + 2 * t["tiles"]: og + ng2 + 2 * t["tiles"] + 2 * (t["tiles"] + 1)].view(np.uint64)
        assert np.array_equal(toff, t["toff"]) and int(got_t[og + ng2 + 4 * t["tiles"] + 2]) == t["sh"]
        kk = (c_uint32 * 2)(int(k[0]), int(k[1]))
        scratch = torch.full((words,), -1, dtype=torch.int32, device=dev)
        for table, ready in ((scratch, 0), (tables[r], 1)):
            mx = torch.zeros((1,), dtype=torch.float32, device=dev)
            tot = torch.zeros((1,), dtype=torch.int64, device=dev)
            anc = torch.full((n,), -1, dtype=torch.int32, device=dev)
            be.check(be.c.gmx_resample_sorted(kk, be.ptr(lw_d), n, shift, be.ptr(tmax), be.ptr(agg), be.ptr(table), ready,
                                              be.ptr(mx), be.ptr(tot), be.ptr(anc), be.stream()), "gmx_resample_sorted")
            got = anc.cpu().numpy()
            assert np.array_equal(got, want), (r, ready, int((got != want).sum()))
            assert int(tot.item()) & 0xFFFFFFFFFFFFFFFF == total
            if not dead:
                assert float(mx.item()) == M
        out = {"distinct": int(np.unique(want).size), "sh": t["sh"]}
    return out


def check_multinomial_sorted_big(n=(1 << 21) + 5000, seed=9, sigma=1.5):
    """the sorted multinomial past 2^21 particles (BASELINE config 4's k = 1e7): tile statistics ->
    gmx_tile_prefix -> gmx_resample_sorted_p (the resampler reads prefixes; the table's tile offsets are computed in
    chunks with a running carry), through smc.resample — ancestors bit-exact vs the oracle's definition."""
    import genjax_amd as G
    from genjax_amd.inference import smc
    dev = G._lib.get().device
    rng = np.random.default_rng(seed)
    lw = rng.normal(0, sigma, n).astype(np.float32)
    assert n > smc.FUSED_RESAMPLE_MAX
    anc, total, mx, shift = smc.resample_fused(smc.MULTINOMIAL_SORTED, G.key(seed + 1), torch.from_numpy(lw).to(dev))
    cdf, ototal, M, oshift = O.weight_cdf_c(lw) if hasattr(O, "weight_cdf_c") else O.weight_cdf(lw)
    assert shift == oshift and int(total.item()) & 0xFFFFFFFFFFFFFFFF == ototal and float(mx.item()) == M
    want = O.ancestors_of_kind(O.MULTINOMIAL_SORTED, O.key(seed + 1), cdf)
    got = anc.cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())
    return int(np.unique(want).size)


def check_nested_edge_cases():
    """two nested loops at their edges: a one-element plate and a 20-element plate of 40-step scans, `repeat(n=20)` of a
    scan, a VECTOR-valued site inside the inner loop ([n, A, T, 3] values: one [A * T, n] plane per element, written
    and — for assess — read back through GMX_F_FLAT)"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    _nested_outer_one_and_repeat.__globals__.update({"G": G, "jnp": jnp})
    _nested_outer_one_and_repeat()
    _nested_vector_site()


def _edge_eq(a, b): return np.array_equal(a.cpu().numpy() if hasattr(a, "cpu") else a, b)

def _nested_outer_one_and_repeat():
    n, T = 11, 40
    def mk(g, scan_of):
        @g.gen
        def step(x, _):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            return xn, xn
        @g.gen
        def series(x0):
            xT, xs = scan_of(step)(x0, None) @ "steps"
            return xT
        return series
    s, os_ = mk(G, lambda f: f.scan(n=T)), mk(O, lambda f: O.Scan(f, T))
    for A in (1, 20):
        tr = s.vmap(in_axes=(0,)).simulate(G.split(G.key(1), n), (jnp.array(np.linspace(0, 1, A).astype(np.float32)),))
        otr = O.Vmap(os_, in_axes=(0,)).simulate(O.split(O.key(1), n), (np.linspace(0, 1, A).astype(np.float32),))
        assert _edge_eq(tr.get_choices()["steps", "x"], otr.get_choices()["steps", "x"]), A
        assert _edge_eq(tr.get_retval(), otr.get_retval()) and _edge_eq(tr.get_score(), otr.get_score())
    tr = s.repeat(n=20).simulate(G.split(G.key(2), n), (0.5,))
    otr = O.Repeat(os_, 20).simulate(O.split(O.key(2), n), (np.float32(0.5),))
    assert _edge_eq(tr.get_choices()["steps", "x"], otr.get_choices()["steps", "x"]) and _edge_eq(tr.get_score(), otr.get_score())

def _nested_vector_site():
    n, A, T = 9, 20, 20
    def mk(g, scan_of):
        @g.gen
        def step(x, _):
            if g is G:
                v = g.normal(x * jnp.array([1.0, 0.5, 0.25]), 1.0) @ "v"
                return v[0] * 0.5 + v[2] * 0.1, None
            v = g.normal(np.asarray(x, np.float32)[..., None] * np.asarray([1.0, 0.5, 0.25], np.float32), np.float32(1.0)) @ "v"
            return (v[..., 0] * np.float32(0.5) + v[..., 2] * np.float32(0.1)).astype(np.float32), None
        @g.gen
        def series(x0):
            xT, _ = scan_of(step)(x0, None) @ "steps"
            return xT
        return series
    def jnp_or_np(g, a):
        return jnp.array(a) if g is G else np.asarray(a, np.float32)
    s, os_ = mk(G, lambda f: f.scan(n=T)), mk(O, lambda f: O.Scan(f, T))
    x0 = np.linspace(-1, 1, A).astype(np.float32)
    tr = s.vmap(in_axes=(0,)).simulate(G.split(G.key(3), n), (jnp.array(x0),))
    otr = O.Vmap(os_, in_axes=(0,)).simulate(O.split(O.key(3), n), (x0,))
    v, ov = tr.get_choices()["steps", "v"], otr.get_choices()["steps", "v"]
    assert tuple(v.shape) == (n, A, T, 3) == tuple(np.shape(ov)), (v.shape, np.shape(ov))
    assert _edge_eq(v, ov) and _edge_eq(tr.get_score(), otr.get_score())
    sc, _ = s.vmap(in_axes=(0,)).assess(tr.get_choices(), (jnp.array(x0),))
    assert _edge_eq(sc, tr.get_score().cpu().numpy())


def check_evidence_unbiased(kind, R=3000, N=32, T=8, seed0=1000, mh=False):
    """INDEPENDENT of the oracle: a bootstrap particle filter's evidence estimate is unbiased, E[exp(log_ml_hat)] = Z,
    whatever the (valid) resampling scheme and however few particles — and for the linear-Gaussian model Z is the
    Kalman filter's closed form.  R sweeps with N = 32 particles under R different keys: the mean of Z_hat / Z must be 1
    within 4 standard errors.  Holds the resampling DEFINITIONS (systematic, stratified, multinomial, the two-stage
    multinomial), the key schedule and the weight algebra against a number neither the oracle nor the product computed."""
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference.smc import BootstrapSweep
    ys = workloads.lgssm_data(T)
    kal = workloads.kalman_log_ml(ys)
    init, step = workloads.make_lgssm(G)
    kw = {}
    if mh:
        # one MH move per step after resampling (BootstrapSweep(rejuvenate=...): the fused MH + extension programs).
        # An INDEPENDENCE proposal: the reference's `Rejuvenate` scores the backward proposal at
        # argument_mapping(OLD choices) (rejuvenate.py:84-88), which is the Metropolis-Hastings ratio only when the
        # proposal does not depend on the current value — with a random-walk proposal the move is not invariant (the
        # evidence comes out 6 % low: DESIGN.md section 3), and that is the reference's behaviour, restated as it is.
        kw["rejuvenate"] = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (0.5, 1.5))})
    sw = BootstrapSweep(init, step, N, T, resample=kind, **kw)
    z = np.empty(R)
    for r in range(R):
        sw.prepare(G.key(seed0 + r), torch.from_numpy(ys))
        sw.launch()
        z[r] = np.exp(sw.log_ml() - kal)
    se = z.std(ddof=1) / np.sqrt(R)
    assert abs(z.mean() - 1.0) < 4.0 * se, (kind, z.mean(), se)
    assert se < 0.05
    return {"mean": float(z.mean()), "se": float(se)}


def check_sampler_laws(n=200_000):
    """INDEPENDENT of the oracle: the product's samplers against scipy's distributions (Kolmogorov–Smirnov on 20 000
    draws, moments on all of them; class frequencies by chi-square) — the evidence there is for the streams the
    reference cannot pin (Gumbel / categorical, Bernoulli, Beta-via-gamma, Dirichlet: DESIGN.md §3)."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    keys = G.split(G.key(77), n)
    val = lambda d, args: d.simulate(keys, args).get_retval().cpu().numpy()
    z = val(G.normal, (1.0, 2.0))
    assert stats.kstest(z[:20000], "norm", args=(1.0, 2.0)).pvalue > 1e-3 and abs(z.mean() - 1.0) < 0.02 and abs(z.std() - 2.0) < 0.02
    # tails: the erf_inv path far from the centre
    assert abs((z > 1.0 + 2.0 * 3.0).mean() - stats.norm.sf(3.0)) < 4 * np.sqrt(stats.norm.sf(3.0) / n)
    u = val(G.uniform, (-1.0, 3.0))
    assert stats.kstest(u[:20000], "uniform", args=(-1.0, 4.0)).pvalue > 1e-3 and u.min() >= -1.0 and u.max() < 3.0
    for a, b in ((2.0, 5.0), (0.5, 0.7), (1.0, 1.0), (30.0, 2.0)):
        x = val(G.beta, (a, b))
        assert stats.kstest(x[:20000], "beta", args=(a, b)).pvalue > 1e-3, (a, b)
        assert abs(x.mean() - a / (a + b)) < 5e-3
    f = val(G.flip, (0.3,))
    assert abs(f.mean() - 0.3) < 4 * np.sqrt(0.21 / n)
    bl = val(G.bernoulli, (0.8,))          # logits
    p = 1.0 / (1.0 + np.exp(-0.8))
    assert abs(bl.mean() - p) < 4 * np.sqrt(p * (1 - p) / n)
    probs = np.array([0.4, 0.2, 0.1, 0.1, 0.05, 0.05, 0.05, 0.05])
    c = val(G.categorical, (jnp.array(np.log(probs).astype(np.float32)),))
    chi = stats.chisquare(np.bincount(c, minlength=8), probs * n)
    assert chi.pvalue > 1e-4, chi
    alpha = np.array([2.0, 3.0, 0.5], np.float32)
    d = val(G.dirichlet, (jnp.array(alpha),))
    assert np.allclose(d.sum(-1), 1.0, atol=1e-5)
    for k_ in range(3):                    # a Dirichlet's marginals are Beta(alpha_k, sum - alpha_k)
        assert stats.kstest(d[:20000, k_], "beta", args=(alpha[k_], alpha.sum() - alpha[k_])).pvalue > 1e-3, k_


def check_offspring_laws(kind, n=256, R=3000):
    """INDEPENDENT of the oracle: a valid resampling scheme has E[offspring_i] = n w_i; multinomial schemes also have
    Var[offspring_i] = n w_i (1 - w_i); systematic offspring are less than 1 away from n w_i, stratified less than 2.  R
    resamplings of ONE weight vector under R keys through the product (smc.resample_fused), against those laws."""
    import genjax_amd as G
    from genjax_amd.inference import smc
    dev = G._lib.get().device
    rng = np.random.default_rng(3)
    lw = rng.normal(0, 1.5, n).astype(np.float32)
    w = np.exp(lw.astype(np.float64) - lw.max())
    w /= w.sum()
    lw_d = torch.from_numpy(lw).to(dev)
    counts = np.zeros((R, n))
    kid = smc._KINDS[kind]
    for r in range(R):
        key = G.key(900 + r)
        if kid == smc.MULTINOMIAL:
            cdf, total, mx, shift = smc.weight_cdf(lw_d)
            anc = smc.ancestors_from_cdf(kid, key, cdf, total)
        else:
            anc = smc.resample_fused(kid, key, lw_d)[0]
        counts[r] = np.bincount(anc.cpu().numpy(), minlength=n)
    mean = counts.mean(0)
    se = np.maximum(counts.std(0, ddof=1), 0.05) / np.sqrt(R)
    assert np.all(np.abs(mean - n * w) < 5.0 * se + 1e-3), (kind, np.max(np.abs(mean - n * w) / se))
    if kind.startswith("multinomial"):
        heavy = np.argsort(-w)[:20]
        var = counts[:, heavy].var(0, ddof=1)
        want = n * w[heavy] * (1 - w[heavy])
        assert np.all(np.abs(var - want) < 0.2 * want + 0.05), (kind, var, want)
    else:
        # one shared offset: less than 1 away; one uniform per stratum: an interval can gain or lose a point at each end
        assert np.all(np.abs(counts - n * w) < (1.0 if kind == "systematic" else 2.0) + 1e-6)
    return {"max_z": float(np.max(np.abs(mean - n * w) / se))}


def check_nested_edits(A, T, n=9):
    """`Update` through a plate of long scans (vmap.py:236-275 over scan.py:509-594): weights, scores, the new and the
    untouched choices and the backward request's constraint (the old observations) against the oracle — a small plate
    (its previous values are [n, A, T] step leaves: the plate runs as a loop around the scans' loops), a mid-size and a
    large one; the scan sub-trace reports one score per series."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Update, numpy as jnp
    def mk(g, scan_of):
        @g.gen
        def step(x, _):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, None
        @g.gen
        def series(x0):
            xT, _ = scan_of(step)(x0, None) @ "steps"
            return xT
        return series
    s, os_ = mk(G, lambda f: f.scan(n=T)), mk(O, lambda f: O.Scan(f, T))
    x0 = np.linspace(-1, 1, A).astype(np.float32)
    model, omodel = s.vmap(in_axes=(0,)), O.Vmap(os_, in_axes=(0,))
    tr = model.simulate(G.split(G.key(1), n), (jnp.array(x0),))
    otr = omodel.simulate(O.split(O.key(1), n), (x0,))
    ys = np.random.default_rng(0).normal(size=(A, T)).astype(np.float32)
    new, w, _, bwd = Update(C["steps", "y"].set(ys)).edit(G.split(G.key(2), n), tr, Diff.no_change((jnp.array(x0),)))
    onew, ow, _ = O.vmap_update(omodel, O.split(O.key(2), n), otr, O.C.d({("steps", "y"): ys}), (x0,))
    assert np.array_equal(w.cpu().numpy(), ow), np.abs(w.cpu().numpy() - ow).max()
    assert np.array_equal(new.get_score().cpu().numpy(), onew.get_score())
    assert np.array_equal(new.get_choices()["steps", "y"].cpu().numpy(), np.broadcast_to(ys, (n, A, T)))
    assert np.array_equal(new.get_choices()["steps", "x"].cpu().numpy(), otr.get_choices()["steps", "x"])
    assert np.array_equal(bwd.constraint["steps", "y"].cpu().numpy(), otr.get_choices()["steps", "y"])
    # ... and against scipy: an Update of observations changes the score by the observation densities' difference
    from scipy import stats
    xo = tr.get_choices()["steps", "x"].cpu().numpy().astype(np.float64)
    yo = tr.get_choices()["steps", "y"].cpu().numpy().astype(np.float64)
    want = (stats.norm.logpdf(ys[None], xo, 1.0) - stats.norm.logpdf(yo, xo, 1.0)).sum((1, 2))
    assert np.allclose(w.cpu().numpy(), want, rtol=2e-5, atol=2e-3)
    # the sub-trace's own score: one per series
    sub = tr.get_subtrace("steps") if hasattr(tr, "get_subtrace") else None
    if sub is not None:
        assert tuple(sub.get_score().shape) == (n, A)


def check_nested_index_edits(A=20, T=24, n=7):
    """`IndexRequest` through two nested counted loops (vmap.py:277-332 over scan.py:325-594), traced once under a gate
    (static._gate_site) because the element's outputs are stored inside its own loop:
      a plate of long scans — IndexRequest(a, Regenerate / Update) with a Python-int index and one index per particle,
      and IndexRequest(a, StaticRequest({"steps": IndexRequest(t, Regenerate)})): ONE step of ONE series, whose successor
      is re-scored against the new carry — weights, scores, return values and every choice against the oracle, the last
      one also against scipy (incl. the reference's own Regenerate weight: the regenerated site's density ratio counts);
      a scan whose step runs a plate — IndexRequest(t, Regenerate of the plate's latent)."""
    import genjax_amd as G
    from genjax_amd import (ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, StaticRequest,
                            Update, numpy as jnp)
    from scipy import stats
    dev = G.split(G.key(0), 1).dev.device if hasattr(G.split(G.key(0), 1), "dev") else None

    def mk(g, scan_of):
        @g.gen
        def step(x, _):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(xn, 1.0) @ "y"
            return xn, None

        @g.gen
        def series(x0):
            xT, _ = scan_of(step)(x0, None) @ "steps"
            return xT
        return series, step
    (s, _), (os_, ostep) = mk(G, lambda f: f.scan(n=T)), mk(O, lambda f: O.Scan(f, T))
    x0 = np.linspace(-1, 1, A).astype(np.float32)
    model, omodel = s.vmap(in_axes=(0,)), O.Vmap(os_, in_axes=(0,))
    tr = model.simulate(G.split(G.key(1), n), (jnp.array(x0),))
    otr = omodel.simulate(O.split(O.key(1), n), (x0,))
    ynew = (np.arange(T, dtype=np.float32) * 0.1)
    a_i, t_i = min(2, A - 1), 5

    class _ScanIdx:                      # the oracle's sub-request object for a static edit: scan.edit_index on "steps"
        def edit(self, k, subtrace, gen_fn, args):
            return O.scan_edit_index(gen_fn, k, subtrace, args, t_i,
                                     lambda kk, sl, a: ostep.regenerate(kk, sl, lambda addr: addr == ("x",), a)[:2])
    cases = [("regen", Regenerate(S["steps", "x"]),
              lambda kb, inner, a: os_.regenerate(kb, inner, lambda addr: addr == ("steps", "x"), a)[:2]),
             ("update", Update(C["steps", "y"].set(ynew)),
              lambda kb, inner, a: os_.update(kb, inner, O.C.d({("steps", "y"): ynew}), a)[:2]),
             ("one step", StaticRequest({"steps": IndexRequest(t_i, Regenerate(S["x"]))}),
              lambda kb, inner, a: os_.edit_static(kb, inner, {"steps": _ScanIdx(), ("steps",): _ScanIdx()}, a))]
    for idx in (a_i, (np.arange(n) * 3) % A):
        for name, sub, edit_all in cases:
            gi = idx if isinstance(idx, int) else torch.from_numpy(idx.astype(np.int32)).to(tr.get_score().device)
            new, w, _, bwd = IndexRequest(gi, sub).edit(G.split(G.key(2), n), tr, Diff.no_change((jnp.array(x0),)))
            onew, ow = O.vmap_edit_index_batched(omodel, O.split(O.key(2), n), otr, idx, edit_all, (x0,))
            assert np.array_equal(w.cpu().numpy(), ow), (name, np.abs(w.cpu().numpy() - ow).max())
            for ad in (("steps", "x"), ("steps", "y")):
                assert np.array_equal(new.get_choices()[ad].cpu().numpy(), onew.get_choices()[ad]), (name, ad)
            assert np.array_equal(new.get_score().cpu().numpy(), onew.get_score()), name
            assert np.array_equal(new.get_retval().cpu().numpy(), onew.get_retval()), name
            assert isinstance(bwd, IndexRequest)
            if name == "one step" and isinstance(idx, int):
                x = new.get_choices()["steps", "x"].cpu().numpy().astype(np.float64)
                xo = tr.get_choices()["steps", "x"].cpu().numpy().astype(np.float64)
                y = tr.get_choices()["steps", "y"].cpu().numpy().astype(np.float64)
                ch = x != xo
                assert ch[:, idx, t_i].all() and ch.sum() == n          # ONE value per particle changed
                lp = stats.norm.logpdf
                want = (lp(x[:, idx, t_i], 0.9 * xo[:, idx, t_i - 1], 0.5) - lp(xo[:, idx, t_i], 0.9 * xo[:, idx, t_i - 1], 0.5)
                        + lp(y[:, idx, t_i], x[:, idx, t_i], 1.0) - lp(y[:, idx, t_i], xo[:, idx, t_i], 1.0)
                        + lp(xo[:, idx, t_i + 1], 0.9 * x[:, idx, t_i], 0.5) - lp(xo[:, idx, t_i + 1], 0.9 * xo[:, idx, t_i], 0.5))
                assert np.allclose(w.cpu().numpy(), want, rtol=2e-5, atol=5e-4), np.abs(w.cpu().numpy() - want).max()
    # a VECTOR-valued site inside the inner loop ([n, A, T, 3] values), gated element-wise
    def mkv(g, scan_of, ones, lift):
        @g.gen
        def stepv(x, _):
            xn = g.normal(0.9 * x, 0.5) @ "x"
            g.normal(lift(xn) * ones, 1.0) @ "v"
            return xn, None

        @g.gen
        def seriesv(x0_):
            xT, _ = scan_of(stepv)(x0_, None) @ "steps"
            return xT
        return seriesv
    sv = mkv(G, lambda f: f.scan(n=T), jnp.ones(3), lambda v: v)
    osv = mkv(O, lambda f: O.Scan(f, T), np.ones(3, np.float32), lambda v: v[..., None])
    mv, omv = sv.vmap(in_axes=(0,)), O.Vmap(osv, in_axes=(0,))
    trv = mv.simulate(G.split(G.key(6), n), (jnp.array(x0),))
    otrv = omv.simulate(O.split(O.key(6), n), (x0,))
    newv, wv, _, _ = IndexRequest(a_i, Regenerate(S["steps", "x"])).edit(G.split(G.key(7), n), trv, Diff.no_change((jnp.array(x0),)))
    onewv, owv = O.vmap_edit_index_batched(
        omv, O.split(O.key(7), n), otrv, a_i,
        lambda kb, inner, a: osv.regenerate(kb, inner, lambda ad: ad == ("steps", "x"), a)[:2], (x0,))
    assert np.array_equal(wv.cpu().numpy(), owv) and np.array_equal(newv.get_score().cpu().numpy(), onewv.get_score())
    for ad in (("steps", "x"), ("steps", "v")):
        assert np.array_equal(newv.get_choices()[ad].cpu().numpy(), onewv.get_choices()[ad]), ad
    # a scan (40 steps) whose step runs a 30-element plate: IndexRequest(t, StaticRequest({x: Regenerate})) — a bare
    # Regenerate would reach the plate, which answers Update / IndexRequest only (vmap.py:342-362), as in the reference;
    # the step after the edited one re-scores its plate and its latent against the new carry; against scipy (the
    # oracle's scan.edit_index slices trailing axes only and does not address a plate inside a step)
    Ts, B, t_e = 40, 30, 7

    @G.gen
    def leaf(m):
        return G.normal(m, 1.0) @ "z"

    @G.gen
    def step2(x, _):
        leaf.repeat(n=B)(x) @ "obs"
        xn = G.normal(0.9 * x, 0.5) @ "x"
        return xn, xn
    sc = step2.scan(n=Ts)
    tr2 = sc.simulate(G.split(G.key(4), n), (0.5, None))
    for idx in (t_e, (np.arange(n) * 5) % (Ts - 1)):
        gi = idx if isinstance(idx, int) else torch.from_numpy(idx.astype(np.int32)).to(tr2.get_score().device)
        try:
            IndexRequest(gi, Regenerate(S["x"])).edit(G.split(G.key(5), n), tr2, Diff.no_change((0.5, None)))
            raise AssertionError("a Regenerate that reaches a plate should be refused")
        except G.NotSupportedEditRequest:
            pass
        new2, w2, _, _ = IndexRequest(gi, StaticRequest({"x": Regenerate(G.Selection.all())})).edit(
            G.split(G.key(5), n), tr2, Diff.no_change((0.5, None)))
        x = new2.get_choices()["x"].cpu().numpy().astype(np.float64)
        xo = tr2.get_choices()["x"].cpu().numpy().astype(np.float64)
        z = tr2.get_choices()["obs", "z"].cpu().numpy().astype(np.float64)            # [n, Ts, B]
        assert np.array_equal(new2.get_choices()["obs", "z"].cpu().numpy(), tr2.get_choices()["obs", "z"].cpu().numpy())
        ii = np.full(n, idx) if isinstance(idx, int) else idx
        r = np.arange(n)
        ch = x != xo
        assert ch[r, ii].all() and ch.sum() == n
        lp = stats.norm.logpdf
        xin = np.where(ii > 0, xo[r, np.maximum(ii - 1, 0)], 0.5)
        want = (lp(x[r, ii], 0.9 * xin, 0.5) - lp(xo[r, ii], 0.9 * xin, 0.5)
                + lp(xo[r, ii + 1], 0.9 * x[r, ii], 0.5) - lp(xo[r, ii + 1], 0.9 * xo[r, ii], 0.5)
                + (lp(z[r, ii + 1], x[r, ii][:, None], 1.0) - lp(z[r, ii + 1], xo[r, ii][:, None], 1.0)).sum(-1))
        assert np.allclose(w2.cpu().numpy(), want, rtol=5e-5, atol=2e-3), np.abs(w2.cpu().numpy() - want).max()
        sc_new = (lp(x[:, 0], 0.45, 0.5) + lp(x[:, 1:], 0.9 * x[:, :-1], 0.5).sum(-1)
                  + lp(z, np.concatenate([np.full((n, 1), 0.5), x[:, :-1]], 1)[..., None], 1.0).sum((1, 2)))
        assert np.allclose(new2.get_score().cpu().numpy(), sc_new, rtol=2e-5, atol=2e-2)
    return True


def check_importance_unbiased(R=3000, K=16):
    """INDEPENDENT of the oracle: `ImportanceK`'s evidence estimate is unbiased, and for a conjugate normal-normal model
    the evidence is a closed form: mu ~ N(0, 1), y ~ N(mu, 0.5), y = 1.3  =>  Z = N(1.3; 0, sqrt(1.25)).  R estimates of
    K = 16 particles each in ONE batched launch set (keys [R]): mean(Z_hat) / Z = 1 within 4 standard errors; the posterior
    mean of the resampled particle E[mu | y] = y / 1.25 within 4 standard errors too."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd.inference.smc import ImportanceK

    @G.gen
    def model():
        mu = G.normal(0.0, 1.0) @ "mu"
        G.normal(mu, 0.5) @ "y"
        return mu
    tgt = G.Target(model, (), G.ChoiceMap.kw(y=1.3))
    alg = ImportanceK(tgt, k_particles=K)
    keys = G.split(G.key(4242), R)
    logz = G.vmap(alg.estimate_normalizing_constant, in_axes=(0, None))(keys, tgt)
    z = np.exp(logz.cpu().numpy().astype(np.float64)) / stats.norm.pdf(1.3, 0.0, np.sqrt(1.25))
    se = z.std(ddof=1) / np.sqrt(R)
    assert abs(z.mean() - 1.0) < 4.0 * se and se < 0.05, (z.mean(), se)
    _, chm = G.vmap(alg.random_weighted, in_axes=(0, None))(keys, tgt)
    mu = chm["mu"].cpu().numpy().astype(np.float64)
    # SIR with K = 16 is biased at O(1 / K); the posterior sd is sqrt(0.2): a loose 0.05 band holds the algebra
    assert abs(mu.mean() - 1.3 / 1.25) < 0.05, mu.mean()
    return {"mean": float(z.mean()), "se": float(se), "post_mean": float(mu.mean())}


def check_marginal_density_unbiased(R=3000):
    """INDEPENDENT of the oracle (GenSP, sp.py:217-254): `Marginal.estimate_logpdf` is an unbiased estimate of the marginal
    density; for mu ~ N(0, 1), s ~ N(mu, 2), x ~ N(mu + s / 2, 0.5) the marginal of x is N(0, sqrt(3.5)).  R keys in one
    batched launch set: with an inner `ImportanceK` on the posterior target (its evidence estimate), and without an
    algorithm (one prior draw's likelihood)."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, SelectionBuilder as S, Target
    from genjax_amd.inference.smc import ImportanceK
    from genjax_amd.inference.sp import Marginal

    @G.gen
    def model():
        mu = G.normal(0.0, 1.0) @ "mu"
        s = G.normal(mu, 2.0) @ "s"
        return G.normal(mu + 0.5 * s, 0.5) @ "x"
    keys = G.split(G.key(99), R)
    want = stats.norm.pdf(0.7, 0.0, np.sqrt(3.5))
    out = {}
    for name, mar in (("importancek", Marginal(model, S["x"], algorithm=ImportanceK(Target(model, (), C["x"].set(0.7)), k_particles=8))),
                      ("prior", Marginal(model, S["x"]))):
        lp = G.vmap(mar.estimate_logpdf, in_axes=(0, None))(keys, C["x"].set(0.7))
        z = np.exp(lp.cpu().numpy().astype(np.float64)) / want
        se = z.std(ddof=1) / np.sqrt(R)
        assert abs(z.mean() - 1.0) < 4.0 * se and se < 0.05, (name, z.mean(), se)
        out[name] = (float(z.mean()), float(se))
    return out


def check_scan_importance_vs_kalman(n=200_000, T=20):
    """INDEPENDENT of the oracle: importance sampling from the prior is unbiased for the evidence, E[exp(w)] = Z, and the
    linear-Gaussian state-space model's Z is the Kalman filter's.  The model as ONE generative function whose latent path is
    a `step.scan(n = T - 1)` (a counted loop in the site program: T - 1 > 16): the mean of exp(w - log Z) over n particles
    is 1 within 4 standard errors (the weights are heavy-tailed at T = 20: the band is wide, the check is on the loop's
    weight algebra, key chain and table-fed observations)."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, workloads
    p = workloads.LGSSM
    ys = workloads.lgssm_data(T)
    kal = workloads.kalman_log_ml(ys)

    @G.gen
    def step(x, _):
        xn = G.normal(p["a"] * x, p["sx"]) @ "x"
        G.normal(xn, p["sy"]) @ "y"
        return xn, None

    @G.gen
    def ssm():
        x0 = G.normal(0.0, p["s0"]) @ "x0"
        G.normal(x0, p["sy"]) @ "y0"
        xT, _ = step.scan(n=T - 1)(x0, None) @ "steps"
        return xT
    _, w = ssm.importance(G.split(G.key(3), n), C["y0"].set(float(ys[0])).set(("steps", "y"), ys[1:]), ())
    z = np.exp(w.cpu().numpy().astype(np.float64) - kal)
    se = z.std(ddof=1) / np.sqrt(n)
    assert abs(z.mean() - 1.0) < 4.0 * se and se < 0.15, (z.mean(), se)
    return {"mean": float(z.mean()), "se": float(se)}


def check_hmc_invariance(n=100_000, L=1):
    """INDEPENDENT of the oracle: an HMC move with the accept test leaves the posterior invariant.  x ~ N(0, 1),
    y ~ N(x, 0.5), y = 1.3: the posterior is N(1.04, 0.2).  n chains START from exact posterior draws; after three
    `HMC(S["x"], 0.3, L)` edits, each accepted where log U < weight (the reference's idiom, tests/inference/
    test_requests.py:131-137), mean and variance are still the posterior's within 4 standard errors — with L = 1, where
    the reference's integrator IS leapfrog.  (For L > 1 the reference's kernel returns the carry's OLD gradient
    (`requests/hmc.py:183-197`: `return (new_trace, values, gradient, momenta)`), so every first half-kick uses the
    initial gradient: not leapfrog, not invariant — restated as it is, bit-exact against the oracle's literal statement:
    with L = 5 the variance grows 0.20 -> 0.30 in three moves at 50 % acceptance.)"""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, SelectionBuilder as S
    from genjax_amd.inference.requests import HMC

    @G.gen
    def model():
        x = G.normal(0.0, 1.0) @ "x"
        G.normal(x, 0.5) @ "y"
    pv = 1.0 / (1.0 + 4.0)
    pm = pv * 1.3 / 0.25
    rng = np.random.default_rng(0)
    x = torch.from_numpy((pm + np.sqrt(pv) * rng.standard_normal(n)).astype(np.float32))
    keys = G.split(G.key(1), n)
    req = HMC(S["x"], 0.3, L=L)
    acc = []
    for i in range(3):
        cur, _ = model.importance(keys, C["x"].set(x).set("y", 1.3), ())
        new, w, _, _ = req.edit(G.split(G.key(10 + i), n), cur, Diff.no_change(()))
        u = torch.from_numpy(np.random.default_rng(100 + i).random(n).astype(np.float32))
        ok = torch.log(u) < w.cpu()
        x = torch.where(ok, new.get_choices()["x"].cpu(), x)
        acc.append(float(ok.float().mean()))
    xv = x.numpy().astype(np.float64)
    out = {"mean": float(xv.mean()), "var": float(xv.var()), "accept": acc}
    if L == 1:
        assert abs(xv.mean() - pm) < 4 * np.sqrt(pv / n) * 3 and abs(xv.var() - pv) < 4 * pv * np.sqrt(2.0 / n) * 3, out
        assert min(acc) > 0.9, out
    return out


def check_edit_weights_against_scipy(n=50_000):
    """INDEPENDENT of the oracle: the weights of the three move requests on x ~ N(0, 1), y ~ N(x, 0.5), y = 1.3, against
    scipy — each as the REFERENCE defines it:
      Regenerate(S["x"])            w = log p(x', y) - log p(x, y): the full density ratio, the prior draw's own density
                                    NOT divided out (distribution.py:266-277: `incremental_w = w - trace.get_score()`) — so
                                    `accept iff log U < w` is not the Metropolis-Hastings test for that proposal either;
      Rejuvenate(q, argmap)         w = log p(x', y) - log p(x, y) + log q(x; argmap(x)) - log q(x'; argmap(x))
                                    (rejuvenate.py:70-94: the backward proposal scored at the OLD choices);
      Update(C["x"].set(v))         w = log p(v, y) - log p(x, y)."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, Regenerate, Rejuvenate, SelectionBuilder as S, StaticRequest, Update

    @G.gen
    def model():
        x = G.normal(0.0, 1.0) @ "x"
        G.normal(x, 0.5) @ "y"
    keys = G.split(G.key(1), n)
    tr, _ = model.importance(keys, C["y"].set(1.3), ())
    x = tr.get_choices()["x"].cpu().numpy().astype(np.float64)
    lj = lambda v: stats.norm.logpdf(v, 0.0, 1.0) + stats.norm.logpdf(1.3, v, 0.5)
    nd = Diff.no_change(())
    new, w, _, _ = Regenerate(S["x"]).edit(G.split(G.key(2), n), tr, nd)
    x2 = new.get_choices()["x"].cpu().numpy().astype(np.float64)
    assert np.allclose(w.cpu().numpy(), lj(x2) - lj(x), rtol=2e-5, atol=2e-5)
    new, w, _, _ = StaticRequest({"x": Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))}).edit(G.split(G.key(3), n), tr, nd)
    x3 = new.get_choices()["x"].cpu().numpy().astype(np.float64)
    want = lj(x3) - lj(x) + stats.norm.logpdf(x, x, 0.5) - stats.norm.logpdf(x3, x, 0.5)
    assert np.allclose(w.cpu().numpy(), want, rtol=2e-5, atol=2e-5)
    new, w, _, _ = Update(C["x"].set(0.25)).edit(G.split(G.key(4), n), tr, nd)
    assert np.allclose(w.cpu().numpy(), lj(0.25) - lj(x), rtol=2e-5, atol=2e-5)


def check_csmc_weights_against_scipy(B=20_000, k=6):
    """INDEPENDENT of the oracle: `ImportanceK.run_csmc` without a proposal, as the reference defines it
    (smc.py:332-346), on mu ~ N(0, 1), y ~ N(mu, 0.5), y = 1.3, for B keys at once: the K - 1 fresh particles carry
    log p(y | mu_i); the RETAINED particle (slot K - 1) carries `target.importance(key, retained)` = log p(mu_r, y) — its
    prior density is not divided out, so conditional SIR with `q = None` does not leave the posterior invariant (measured
    from exact posterior draws: mean 1.04 -> 0.96); with a proposal `q` the weights are target score - q score.  The
    restatement keeps the reference's arithmetic; this pins it against scipy."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Target
    from genjax_amd.inference.smc import ImportanceK

    @G.gen
    def model():
        mu = G.normal(0.0, 1.0) @ "mu"
        G.normal(mu, 0.5) @ "y"
    rng = np.random.default_rng(0)
    mu_r = (1.04 + np.sqrt(0.2) * rng.standard_normal(B)).astype(np.float32)
    alg = ImportanceK(Target(model, (), C["y"].set(1.3)), k_particles=k)
    pc = alg.run_csmc(G.split(G.key(3), B), C.d({"mu": torch.from_numpy(mu_r)}))
    lw = pc.get_log_weights().cpu().numpy().astype(np.float64)
    mus = pc.get_particles().get_choices()["mu"].cpu().numpy().astype(np.float64)
    assert lw.shape == (B, k) and np.array_equal(mus[:, -1].astype(np.float32), mu_r)
    assert np.allclose(lw[:, :-1], stats.norm.logpdf(1.3, mus[:, :-1], 0.5), rtol=2e-5, atol=2e-5)
    assert np.allclose(lw[:, -1], stats.norm.logpdf(mus[:, -1], 0.0, 1.0) + stats.norm.logpdf(1.3, mus[:, -1], 0.5), rtol=2e-5, atol=2e-5)
    # the fresh particles are prior draws
    assert stats.kstest(mus[:2000, 0], "norm").pvalue > 1e-3


def check_more_closed_forms(n=200_000):
    """INDEPENDENT of the oracle, four more: (1) the enumerative Gibbs step of BASELINE config 5 (`gibbs_categorical`):
    n datapoints that all observe the SAME value — their assignments are iid from the closed-form conditional
    softmax(log pi_k + log N(x; mu_k, 1)): chi-square over K = 8; (2) a `Mask`ed constraint's weight is the constrained
    density where the flag holds and 0 elsewhere (scipy); (3) `IndexRequest(j, Regenerate)` on a plate: the weight is
    the regenerated element's density ratio (the reference's Regenerate arithmetic) and only element j moves;
    (4) `ChangeTarget` from an `ImportanceK` under one observation to the target under another (the same latents): its
    evidence estimate is unbiased for the NEW target's evidence (conjugate closed form)."""
    from scipy import stats
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Mask, Regenerate, SelectionBuilder as S, Target
    from genjax_amd import numpy as jnp, workloads
    from genjax_amd.inference import gibbs
    from genjax_amd.inference.smc import ChangeTarget, ImportanceK
    dev = G._lib.get().device
    # (1)
    K = 8
    probs = np.array([0.3, 0.2, 0.15, 0.1, 0.1, 0.05, 0.05, 0.05], np.float32)
    mus = np.linspace(-3.0, 4.0, K).astype(np.float32)
    gd = workloads.make_mixture(G)
    xval = 0.8
    idx = gibbs.gibbs_categorical(G.key(5), gd, (torch.from_numpy(probs).to(dev), torch.from_numpy(mus).to(dev)),
                                  C["obs"].set(torch.full((n,), xval, dtype=torch.float32, device=dev)), "idx", K)
    post = probs.astype(np.float64) * stats.norm.pdf(xval, mus.astype(np.float64), 1.0)
    post /= post.sum()
    chi = stats.chisquare(np.bincount(idx.cpu().numpy(), minlength=K), post * n)
    assert chi.pvalue > 1e-4, chi
    # (2)
    @G.gen
    def m2():
        x = G.normal(0.0, 1.0) @ "x"
        G.normal(x, 0.5) @ "y"
    m = 20_000
    flag = torch.from_numpy(np.random.default_rng(1).random(m) < 0.4).to(dev)
    tr, w = m2.importance(G.split(G.key(6), m), C["y"].set(Mask(1.3, flag)), ())
    x = tr.get_choices()["x"].cpu().numpy().astype(np.float64)
    want = np.where(flag.cpu().numpy(), stats.norm.logpdf(1.3, x, 0.5), 0.0)
    assert np.allclose(w.cpu().numpy(), want, rtol=2e-5, atol=2e-5)
    yv = tr.get_choices()["y"].cpu().numpy()
    assert np.all(yv[flag.cpu().numpy()] == np.float32(1.3)) and not np.any(yv[~flag.cpu().numpy()] == np.float32(1.3))
    # (3)
    @G.gen
    def elem(mu):
        z = G.normal(mu, 1.0) @ "z"
        G.normal(z, 0.5) @ "o"
    P, j = 40, 17
    locs = np.linspace(-1, 1, P).astype(np.float32)
    obs = np.linspace(0.5, -0.5, P).astype(np.float32)
    plate = elem.vmap(in_axes=(0,))
    args = (jnp.array(locs),)
    tr3, _ = plate.importance(G.split(G.key(7), m), C["o"].set(obs), args)
    new, w3, _, _ = IndexRequest(j, Regenerate(S["z"])).edit(G.split(G.key(8), m), tr3, Diff.no_change(args))
    z0 = tr3.get_choices()["z"].cpu().numpy().astype(np.float64)
    z1 = new.get_choices()["z"].cpu().numpy().astype(np.float64)
    assert np.array_equal(np.delete(z0, j, 1), np.delete(z1, j, 1)) and not np.array_equal(z0[:, j], z1[:, j])
    lj = lambda z: stats.norm.logpdf(z, locs[j], 1.0) + stats.norm.logpdf(obs[j], z, 0.5)
    assert np.allclose(w3.cpu().numpy(), lj(z1[:, j]) - lj(z0[:, j]), rtol=2e-5, atol=2e-5)
    # (4)
    @G.gen
    def m4():
        mu = G.normal(0.0, 1.0) @ "mu"
        G.normal(mu, 0.5) @ "y"
    R = 4000
    prior_alg = ImportanceK(Target(m4, (), C["y"].set(0.0)), k_particles=16)     # the same latents under another observation
    tgt = Target(m4, (), C["y"].set(1.3))
    keys = G.split(G.key(9), R)
    logz = G.vmap(lambda k: ChangeTarget(prior_alg, tgt).run_smc(k).get_log_marginal_likelihood_estimate())(keys)
    zz = np.exp(logz.cpu().numpy().astype(np.float64)) / stats.norm.pdf(1.3, 0.0, np.sqrt(1.25))
    se = zz.std(ddof=1) / np.sqrt(R)
    assert abs(zz.mean() - 1.0) < 4 * se and se < 0.05, (zz.mean(), se)


def check_index_request_o1(n=96, P=40, seed=12, edits=5, nested=True):
    """`IndexRequest` on a long plate held per particle edits ONE element (vmap.py:277-332 `edit_index`; VERDICT r3 item
    6): chains of edits — Python-int index and one index per particle, Regenerate and Update sub-requests, longer than
    PATCH_DEPTH_MAX so that the lazy leaves are folded on the way — bit-exact against the oracle's slice / edit /
    write-back (values, weights, IN-ORDER score, backward request), the old trace untouched; `nested`: a plate of
    plates, IndexRequest(i, IndexRequest(j, ...))."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, Update, numpy as jnp
    from genjax_amd import combinators as cmb
    from genjax_amd.engine import Patched
    dev = G._lib.get().device
    sig = np.linspace(1.0, 3.0, P).astype(np.float32)
    school, oschool = _school(G), _school(O)
    v, ov = school.vmap(in_axes=(None, None, 0)), O.Vmap(oschool, in_axes=(None, None, 0))
    args = (1.0, 2.0, jnp.array(sig))
    oargs = (np.float32(1.0), np.float32(2.0), sig)
    t0, ot0 = v.simulate(G.split(G.key(seed), n), args), ov.simulate(O.split(O.key(seed), n), oargs)
    th0 = t0.get_choices()["theta"].cpu().numpy().copy()
    tr, otr = t0, ot0
    rng = np.random.default_rng(seed)
    saw_lazy = False
    for e in range(edits):
        kind = e % 3
        k, ok = G.split(G.key(seed + 100 + e), n), O.split(O.key(seed + 100 + e), n)
        if kind == 0:                  # Python-int index, Regenerate
            j = int(rng.integers(0, P))
            new, w, _, bwd = IndexRequest(j, Regenerate(S["theta"])).edit(k, tr, Diff.no_change(args))
            aj = (np.float32(1.0), np.float32(2.0), np.float32(sig[j]))
            onew, ow = O.vmap_edit_index(ov, ok, otr, j, lambda kk, sl, a: oschool.regenerate(kk, sl, O.selection("theta"), a)[:2], aj)
            assert isinstance(bwd, IndexRequest) and bwd.idx == j
        elif kind == 1:                # Python-int index, Update of y
            j = int(rng.integers(0, P))
            yv = np.float32(rng.normal())
            new, w, _, bwd = IndexRequest(j, Update(C["y"].set(float(yv)))).edit(k, tr, Diff.no_change(args))
            aj = (np.float32(1.0), np.float32(2.0), np.float32(sig[j]))
            old_y = otr.get_choices()["y"][:, j].copy()
            onew, ow = O.vmap_edit_index(ov, ok, otr, j, lambda kk, sl, a: oschool.update(kk, sl, O.C.d({"y": np.full(n, yv, np.float32)}), a)[:2], aj)
            assert np.array_equal(bwd.request.constraint["y"].cpu().numpy(), old_y)
        else:                          # one index per particle, Regenerate
            idx = rng.integers(0, P, n).astype(np.int32)
            new, w, _, bwd = IndexRequest(torch.from_numpy(idx).to(dev), Regenerate(S["theta"])).edit(k, tr, Diff.no_change(args))
            onew, ow = O.vmap_edit_index_per_particle(ov, ok, otr, idx, lambda kk, sl, a: oschool.regenerate(kk, sl, O.selection("theta"), a)[:2],
                                                      lambda j_: (np.float32(1.0), np.float32(2.0), np.float32(sig[j_])))
        saw_lazy = saw_lazy or isinstance(new.inner.subtraces["theta"].value, Patched)
        assert np.array_equal(w.cpu().numpy(), ow), (e, kind)
        for a in ("theta", "y"):
            assert np.array_equal(new.get_choices()[a].cpu().numpy(), onew.get_choices()[a]), (e, kind, a)
        if e % 2 == 0 or e == edits - 1:
            assert np.array_equal(new.get_score().cpu().numpy(), onew.get_score()), (e, kind)
            assert np.array_equal(new.get_retval().cpu().numpy(), onew.get_retval()), (e, kind)
        tr, otr = new, onew
    assert saw_lazy, "the O(1) path did not run"
    assert np.array_equal(t0.get_choices()["theta"].cpu().numpy(), th0)          # the first trace is untouched
    s, _ = v.assess(tr.get_choices(), args)
    assert np.array_equal(s.cpu().numpy(), tr.get_score().cpu().numpy())
    if not nested:
        return
    # a plate of plates: [n, P1, P2]; IndexRequest(i, IndexRequest(j, Regenerate))
    P1, P2 = 20, 24
    sig2 = np.linspace(1.0, 2.0, P1 * P2).astype(np.float32).reshape(P1, P2)
    vv = school.vmap(in_axes=(None, None, 0)).vmap(in_axes=(None, None, 0))
    ovv = O.Vmap(O.Vmap(oschool, in_axes=(None, None, 0)), in_axes=(None, None, 0))
    a2, oa2 = (0.5, 1.5, jnp.array(sig2)), (np.float32(0.5), np.float32(1.5), sig2)
    t2, ot2 = vv.simulate(G.split(G.key(seed + 7), n), a2), ovv.simulate(O.split(O.key(seed + 7), n), oa2)
    assert np.array_equal(t2.get_choices()["theta"].cpu().numpy(), ot2.get_choices()["theta"])
    i, j = 3, 17
    new, w, _, bwd = IndexRequest(i, IndexRequest(j, Regenerate(S["theta"]))).edit(G.split(G.key(seed + 8), n), t2, Diff.no_change(a2))
    inner_ov = O.Vmap(oschool, in_axes=(None, None, 0))
    a_j = (np.float32(0.5), np.float32(1.5), sig2[:, j])           # [P1] against the (n, P1) batch of the inner plates
    onew, ow = O.vmap_edit_index_batched(
        ovv, O.split(O.key(seed + 8), n), ot2, i,
        lambda kb, inner, a: O.vmap_edit_index(inner_ov, kb, inner, j,
                                               lambda k3, s3, a3: oschool.regenerate(k3, s3, O.selection("theta"), a3)[:2], a_j), oa2)
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(new.get_choices()["theta"].cpu().numpy(), onew.get_choices()["theta"])
    assert np.array_equal(new.get_score().cpu().numpy(), onew.get_score())
    assert isinstance(bwd.request, IndexRequest) and bwd.idx == i and bwd.request.idx == j


def check_one_trace_with_large_plates(n=5000, seed=21, timing=False):
    """ONE trace (no particle batch) of a `@gen` model that holds LARGE plates runs site by site (genjax_amd/sitewise.py;
    the reference's shape of `4_index_request.ipynb` c3-c9 and of the mixture model's `generate_data`): bare-distribution
    plates and a plate of a `@gen` element inside a NESTED `@gen` call, sums of plate values feeding a later site —
    simulate / importance / assess, `Update` of a whole plate (as a constraint and as a StaticRequest),
    `StaticRequest({"a": IndexRequest(i, Update(v))})` (one element: every other site's sub-trace is shared) and its
    backward request — bit-exact against the oracle (which runs models eagerly by construction)."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, StaticRequest, Update, numpy as jnp
    sig = np.linspace(1.0, 2.0, n).astype(np.float32)
    school, oschool = _school(G), _school(O)

    @G.gen
    def data(mu):
        return school.vmap(in_axes=(None, None, 0))(mu, 2.0, jnp.array(sig)) @ "schools"

    @G.gen
    def model():
        x = G.normal(0.0, 1.0) @ "x"
        a = G.normal.vmap()(jnp.zeros(n), jnp.ones(n)) @ "a"
        b = G.normal.vmap()(jnp.zeros(n) + 1.0, jnp.ones(n) * 2.0) @ "b"
        th = data(x) @ "data"
        obs = G.normal(jnp.sum(a) + jnp.sum(b) + jnp.sum(th) + x, 5.0) @ "obs"
        return obs

    @O.gen
    def odata(mu):
        return O.Vmap(oschool, in_axes=(None, None, 0))(mu, np.float32(2.0), sig) @ "schools"

    @O.gen
    def omodel():
        x = O.normal(0.0, 1.0) @ "x"
        a = O.Vmap(O.normal, in_axes=(0, 0))(np.zeros(n, np.float32), np.ones(n, np.float32)) @ "a"
        b = O.Vmap(O.normal, in_axes=(0, 0))(np.zeros(n, np.float32) + np.float32(1.0), np.ones(n, np.float32) * np.float32(2.0)) @ "b"
        th = odata(x) @ "data"
        s = ((O.sum_vector(a) + O.sum_vector(b)).astype(np.float32) + O.sum_vector(th)).astype(np.float32)
        obs = O.normal((s + x).astype(np.float32), 5.0) @ "obs"
        return obs
    f32 = lambda v: np.float32(v.item() if hasattr(v, "item") else v)
    tr, otr = model.simulate(G.key(seed), ()), omodel.simulate(O.key(seed), ())
    assert getattr(tr, "_site_by_site", False), "the call did not take the site-by-site form"
    for adr in ("x", "a", "b", "obs", ("data", "schools", "theta"), ("data", "schools", "y")):
        assert np.array_equal(tr.get_choices()[adr].cpu().numpy(), np.asarray(otr.get_choices()[adr])), adr
    assert f32(tr.get_score()) == f32(otr.get_score())
    tr2, w = model.importance(G.key(seed + 1), C["obs"].set(1.0), ())
    otr2, ow = omodel.importance(O.key(seed + 1), O.C.d({"obs": np.float32(1.0)}), ())
    assert f32(w) == f32(ow) and f32(tr2.get_score()) == f32(otr2.get_score())
    s, _ = model.assess(tr2.get_choices(), ())
    so, _ = omodel.assess(otr2.get_choices(), (), ())
    assert f32(s) == f32(so) == f32(tr2.get_score())
    # Update of a whole plate: as a constraint, and as a request (4_index_request.ipynb c7)
    vals = np.linspace(-1.0, 1.0, n).astype(np.float32)
    t0 = time.perf_counter()
    new1, w1, _, disc1 = tr2.update(G.key(seed + 2), C["a"].set(jnp.array(vals)), Diff.no_change(()))
    _ = f32(w1)
    t_update = time.perf_counter() - t0
    new2, w2, _, _ = StaticRequest({"a": Update(C.v(jnp.array(vals)))}).edit(G.key(seed + 2), tr2, Diff.no_change(()))
    onew, ow1, odisc = omodel.update(O.key(seed + 2), otr2, O.C.d({"a": vals}), ())
    assert f32(w1) == f32(w2) == f32(ow1)
    assert np.array_equal(new1.get_choices()["a"].cpu().numpy(), vals) and f32(new1.get_score()) == f32(onew.get_score())
    assert np.array_equal(disc1["a"].cpu().numpy(), np.asarray(otr2.get_choices()["a"]))
    # one element of a bare plate, and one element of the nested `@gen` plate (c9)
    req = StaticRequest({"a": IndexRequest(jnp.array(3), Update(C.v(42.0)))})
    req.edit(G.key(seed + 3), tr2, Diff.no_change(()))                 # (warm: programs compiled)
    t0 = time.perf_counter()
    new3, w3, _, bwd3 = req.edit(G.key(seed + 3), tr2, Diff.no_change(()))
    _ = f32(w3)
    t_index = time.perf_counter() - t0
    a3 = new3.get_choices()["a"].cpu().numpy()
    assert (a3 == 42.0).sum() == 1 and a3[3] == 42.0
    av = np.asarray(otr2.get_choices()["a"]).copy()
    av[3] = 42.0
    onew3, ow3, _ = omodel.update(O.key(seed + 3), otr2, O.C.d({"a": av}), ())
    assert f32(new3.get_score()) == f32(onew3.get_score()) and f32(w3) == f32(ow3), (f32(w3), f32(ow3))
    for adr in ("x", "b", "data"):
        assert new3.subtraces[adr] is tr2.subtraces[adr]              # untouched sites share their sub-traces
    back, wb, _, _ = bwd3.edit(G.key(seed + 4), new3, Diff.no_change(()))
    assert np.array_equal(back.get_choices()["a"].cpu().numpy(), np.asarray(otr2.get_choices()["a"]))
    assert f32(back.get_score()) == f32(tr2.get_score())
    req4 = StaticRequest({"data": StaticRequest({"schools": IndexRequest(n - 2, Update(C["y"].set(0.5)))})})
    new4, w4, _, _ = req4.edit(G.key(seed + 5), tr2, Diff.no_change(()))
    yv = np.asarray(otr2.get_choices()["data", "schools", "y"]).copy()
    yv[n - 2] = 0.5
    onew4, ow4, _ = omodel.update(O.key(seed + 5), otr2, O.C.d({("data", "schools", "y"): yv}), ())
    assert np.array_equal(new4.get_choices()["data", "schools", "y"].cpu().numpy(), yv)
    assert f32(new4.get_score()) == f32(onew4.get_score()) and f32(w4) == f32(ow4)
    assert new4.subtraces["a"] is tr2.subtraces["a"] and new4.subtraces["obs"] is tr2.subtraces["obs"]   # theta unchanged: obs too
    return (t_index, t_update) if timing else None


def check_one_trace_with_large_vector_sites(n=5000, K=8, seed=31):
    """ONE trace of a model whose DISTRIBUTION sites hold thousands of elements — `categorical(logits, sample_shape=n)`
    and `normal(clusters[idx] + mu, 1.0)` with idx of n elements: the shape of the mixture model's `generate_datapoints`
    (7_application_dirichlet_mixture_model.ipynb c6) — runs site by site with the elements of such a site on the launch
    axis (sitewise.vector_site: element i draws with counter i from the ONE site key, as `tfd.X.sample(seed=key)` of that
    shape does).  simulate / importance / assess, `Update` of an upstream value (the site is re-scored against its new
    arguments) and of the site's own value: bit-exact against the oracle; the first draws equal the unrolled form's."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, numpy as jnp
    dev = G._lib.get().device
    cl_h = np.linspace(-3.0, 3.0, K).astype(np.float32)
    lg_h = np.linspace(-0.5, 0.5, K).astype(np.float32)

    @G.gen
    def model():
        mu = G.normal(0.0, 1.0) @ "mu"
        idx = G.categorical(logits=jnp.array(lg_h), sample_shape=n) @ "idx"
        cl = torch.from_numpy(cl_h).to(dev)
        obs = G.normal(cl[idx.long()] + mu, 1.0) @ "obs"
        return obs

    @O.gen
    def omodel():
        mu = O.normal(0.0, 1.0) @ "mu"
        idx = O.categorical(logits=lg_h, sample_shape=n) @ "idx"
        obs = O.normal((cl_h[idx] + mu).astype(np.float32), 1.0) @ "obs"
        return obs
    f32 = lambda v: np.float32(v.item() if hasattr(v, "item") else v)
    tr, otr = model.simulate(G.key(seed), ()), omodel.simulate(O.key(seed), ())
    assert getattr(tr, "_site_by_site", False)
    assert np.array_equal(tr.get_choices()["idx"].cpu().numpy(), otr.get_choices()["idx"])
    assert np.array_equal(tr.get_choices()["obs"].cpu().numpy(), otr.get_choices()["obs"])
    assert f32(tr.get_score()) == f32(otr.get_score())
    m = 12                                   # the unrolled definition of the same site: the same first draws

    @G.gen
    def small():
        mu = G.normal(0.0, 1.0) @ "mu"
        return G.categorical(logits=jnp.array(lg_h), sample_shape=m) @ "idx"
    if K <= 8:                               # (an unrolled site holds its K logits in registers per draw)
        assert np.array_equal(small.simulate(G.key(seed), ()).get_choices()["idx"].cpu().numpy(), otr.get_choices()["idx"][:m])
    ys = np.linspace(-4.0, 4.0, n).astype(np.float32)
    tr2, w = model.importance(G.key(seed + 1), C["obs"].set(jnp.array(ys)), ())
    otr2, ow = omodel.importance(O.key(seed + 1), O.C.d({"obs": ys}), ())
    assert f32(w) == f32(ow) and f32(tr2.get_score()) == f32(otr2.get_score())
    s, _ = model.assess(tr2.get_choices(), ())
    so, _ = omodel.assess(otr2.get_choices(), (), ())
    assert f32(s) == f32(so) == f32(tr2.get_score())
    new, w3, _, _ = tr2.update(G.key(seed + 2), C["mu"].set(0.25), Diff.no_change(()))
    onew, ow3, _ = omodel.update(O.key(seed + 2), otr2, O.C.d({"mu": np.float32(0.25)}), ())
    assert f32(w3) == f32(ow3) and f32(new.get_score()) == f32(onew.get_score())
    assert new.subtraces["idx"] is tr2.subtraces["idx"]
    ys2 = ys[::-1].copy()
    new2, w4, _, disc = new.update(G.key(seed + 3), C["obs"].set(jnp.array(ys2)), Diff.no_change(()))
    onew2, ow4, _ = omodel.update(O.key(seed + 3), onew, O.C.d({"obs": ys2}), ())
    assert f32(w4) == f32(ow4) and f32(new2.get_score()) == f32(onew2.get_score())
    assert np.array_equal(disc["obs"].cpu().numpy(), ys)
    # Regenerate of a scalar site, a changed ARGUMENT of the model, project — around a 5000-element site that stays untouched
    from genjax_amd import Regenerate, SelectionBuilder as S
    mu_h = np.linspace(-1.0, 1.0, n).astype(np.float32)

    @G.gen
    def m2(scale):
        x = G.normal(0.0, scale) @ "x"
        a = G.normal(jnp.array(mu_h) + x, 1.0) @ "a"             # (a host table + a device value)
        return G.normal(jnp.sum(a) + x, 5.0) @ "obs"

    @O.gen
    def om2(scale):
        x = O.normal(0.0, scale) @ "x"
        a = O.normal((mu_h + x).astype(np.float32), np.float32(1.0)) @ "a"
        return O.normal((O.sum_vector(a) + x).astype(np.float32), 5.0) @ "obs"
    t, ot = m2.simulate(G.key(seed + 4), (1.0,)), om2.simulate(O.key(seed + 4), (np.float32(1.0),))
    assert np.array_equal(t.get_choices()["a"].cpu().numpy(), ot.get_choices()["a"])
    r, wr, _, bwd = Regenerate(S["x"]).edit(G.key(seed + 5), t, Diff.no_change((1.0,)))
    orr, owr, _ = om2.regenerate(O.key(seed + 5), ot, O.selection("x"), (np.float32(1.0),))
    assert f32(wr) == f32(owr) and f32(r.get_score()) == f32(orr.get_score())          # ("a" is re-scored around its new mean)
    assert np.array_equal(r.get_choices()["a"].cpu().numpy(), ot.get_choices()["a"])
    ra, wa, _, bwa = Regenerate(S["a"]).edit(G.key(seed + 9), t, Diff.no_change((1.0,)))           # the large site drawn again
    ora, owa, _ = om2.regenerate(O.key(seed + 9), ot, O.selection("a"), (np.float32(1.0),))
    assert f32(wa) == f32(owa) and np.array_equal(ra.get_choices()["a"].cpu().numpy(), ora.get_choices()["a"])
    assert np.array_equal(bwa.edit(G.key(seed + 10), ra, Diff.no_change((1.0,)))[0].get_choices()["a"].cpu().numpy(), ot.get_choices()["a"])
    back, wb, _, _ = bwd.edit(G.key(seed + 6), r, Diff.no_change((1.0,)))
    assert f32(back.get_choices()["x"]) == f32(t.get_choices()["x"]) and float(wr + wb) == 0.0
    u, wu, _, _ = t.update(G.key(seed + 7), C.n(), Diff.unknown_change((2.0,)))
    ou, owu, _ = om2.update(O.key(seed + 7), ot, O.C.d({}), (np.float32(2.0),))
    assert f32(wu) == f32(owu) and f32(u.get_score()) == f32(ou.get_score())
    assert f32(t.project(G.key(seed + 8), S["a"])) == f32(t.subtraces["a"].get_score())


def check_mixture_notebook_model(n=5000, k=12, seed=0):
    """`generate_data` of 7_application_dirichlet_mixture_model.ipynb (c6), written as the notebook writes it — a `repeat` of
    cluster means, an INLINED Dirichlet for the weights, a nested `@gen` call whose `categorical(log probs,
    sample_shape=n)` and `normal(clusters[idx], sigma)` sites hold all n datapoints — for ONE trace: simulate, and the
    notebook's `importance` under `C["datapoints", "obs"]` | `C["probs"]` (c10) and the `trace.update` calls its Gibbs moves
    end with, bit-exact against the oracle.  A Dirichlet of more than 20 components is not one program either (2 k live
    values): sitewise.dirichlet_site runs its operations as a handful of launches over the components.  (k = 13 .. 16 makes
    the Update of the un-batched, UNROLLED `clusters` plate exceed 64 input slots — the tests run 12, 40 and 64.)"""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp
    from genjax_amd.core.pytree import Const
    PRIOR_MEAN, PRIOR_VARIANCE, OBS_VARIANCE = 50.0, 10.0, 1.0
    alpha = float(n / (k * 10))

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
    def generate_data(n_clusters, n_datapoints, alpha_):
        clusters = generate_cluster.repeat(n=n_clusters.unwrap())(PRIOR_MEAN, PRIOR_VARIANCE) @ "clusters"
        probs = generate_cluster_weight.inline(alpha_ / n_clusters.unwrap() * jnp.ones(n_clusters.unwrap()))
        return generate_datapoints(probs, clusters, n_datapoints) @ "datapoints"

    @O.gen
    def o_cluster(mean, var):
        return O.normal(mean, var) @ "mean"

    @O.gen
    def o_datapoints(probs, clusters):
        idx = O.categorical(logits=O.log(np.asarray(probs, np.float32)), sample_shape=n) @ "idx"
        return O.normal(np.asarray(clusters, np.float32)[idx], np.float32(OBS_VARIANCE)) @ "obs"

    @O.gen
    def o_data():
        clusters = O.Repeat(o_cluster, k)(np.float32(PRIOR_MEAN), np.float32(PRIOR_VARIANCE)) @ "clusters"
        # (`alpha / n_clusters` inside the model: alpha arrives as an f32 scalar — the reference stages the source, a
        #  Python float argument is a weak-typed f32 tracer there — so the division is an f32 one)
        probs = O.dirichlet(((np.float32(alpha) / np.float32(k)) * np.ones(k, np.float32)).astype(np.float32)) @ "probs"
        return o_datapoints(probs, clusters) @ "datapoints"
    args = (Const(k), Const(n), alpha)
    f32 = lambda v: np.float32(v.item() if hasattr(v, "item") else v)
    tr, otr = generate_data.simulate(G.key(seed), args), o_data.simulate(O.key(seed), ())
    for adr in (("clusters", "mean"), "probs", ("datapoints", "idx"), ("datapoints", "obs")):
        assert np.array_equal(tr.get_choices()[adr].cpu().numpy(), np.asarray(otr.get_choices()[adr])), adr
    assert f32(tr.get_score()) == f32(otr.get_score())
    pts = np.linspace(10.0, 90.0, n).astype(np.float32)
    uniform = (np.ones(k, np.float32) / np.float32(k)).astype(np.float32)
    constraints = C["datapoints", "obs"].set(jnp.array(pts)) | C["probs"].set(jnp.ones(k) / k)
    tr2, w = generate_data.importance(G.key(seed + 1), constraints, args)
    otr2, ow = o_data.importance(O.key(seed + 1), O.C.d({("datapoints", "obs"): pts, "probs": uniform}), ())
    assert np.array_equal(tr2.get_choices()["datapoints", "idx"].cpu().numpy(), otr2.get_choices()["datapoints", "idx"])
    assert f32(w) == f32(ow) and f32(tr2.get_score()) == f32(otr2.get_score()), (f32(w), f32(ow))
    # the notebook's Gibbs moves write their draws back with `trace.update` (c10: update_cluster_means /
    # update_cluster_weights): new cluster means re-score the n observations, new weights the n assignments
    from genjax_amd import Diff
    new_means = np.linspace(20.0, 80.0, k).astype(np.float32)
    tr3, w3, _, _ = tr2.update(G.key(seed + 2), C["clusters", "mean"].set(jnp.array(new_means)), Diff.no_change(args))
    otr3, ow3, _ = o_data.update(O.key(seed + 2), otr2, O.C.d({("clusters", "mean"): new_means}), ())
    assert f32(w3) == f32(ow3) and f32(tr3.get_score()) == f32(otr3.get_score()), (f32(w3), f32(ow3))
    assert tr3.subtraces["datapoints"].subtraces["idx"] is tr2.subtraces["datapoints"].subtraces["idx"]
    new_probs = np.linspace(1.0, 2.0, k).astype(np.float32)
    new_probs = (new_probs / new_probs.sum(dtype=np.float32)).astype(np.float32)
    tr4, w4, _, _ = tr3.update(G.key(seed + 3), C["probs"].set(jnp.array(new_probs)), Diff.no_change(args))
    otr4, ow4, _ = o_data.update(O.key(seed + 3), otr3, O.C.d({"probs": new_probs}), ())
    assert f32(w4) == f32(ow4) and f32(tr4.get_score()) == f32(otr4.get_score()), (f32(w4), f32(ow4))
    # update_datapoint_assignment (c10): n draws from `categorical.simulate(key, (local_densities [n, k],))` under ONE key,
    # written back with `trace.update(C["datapoints", "idx"].set(...))` — the n observations are re-scored
    dens = np.random.default_rng(seed).normal(size=(n, k)).astype(np.float32)
    new_idx = G.categorical.simulate(G.key(seed + 4), (torch.from_numpy(dens).to(G._lib.get().device),)).get_choices().get_value()
    o_idx = O.categorical._sample(O.key(seed + 4), (dens,))
    assert np.array_equal(new_idx.cpu().numpy(), o_idx)
    tr5, w5, _, _ = tr4.update(G.key(seed + 5), C["datapoints", "idx"].set(new_idx), Diff.no_change(args))
    otr5, ow5, _ = o_data.update(O.key(seed + 5), otr4, O.C.d({("datapoints", "idx"): o_idx}), ())
    assert f32(w5) == f32(ow5) and f32(tr5.get_score()) == f32(otr5.get_score()), (f32(w5), f32(ow5))


def check_deferred_plate(B=64, n=4096, seed=41, timing=False):
    """A model over a SMALL batch of particles whose last site is a LARGE plate (`ImportanceK` with tens of particles over
    thousands of datapoints): the plate is lifted out of the program and run over particles x elements
    (combinators.Vmap._defer / run_deferred) — simulate, importance under per-element observations, assess, and
    `ImportanceK.run_smc` on such a target: bit-exact against the oracle AND against the loop form of the same call
    (deferral switched off), which it must replace without a trace."""
    import time
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, combinators as cmb, numpy as jnp, static
    from genjax_amd.inference.smc import ImportanceK
    sig = np.linspace(1.0, 3.0, n).astype(np.float32)
    ys = np.linspace(-2.0, 2.0, n).astype(np.float32)
    school, oschool = _school(G), _school(O)

    @G.gen
    def model(scale):
        mu = G.normal(0.0, scale) @ "mu"
        tau = G.normal(2.0, 0.1) @ "tau"
        return school.vmap(in_axes=(None, None, 0))(mu, tau, jnp.array(sig)) @ "schools"

    @O.gen
    def omodel(scale):
        mu = O.normal(0.0, scale) @ "mu"
        tau = O.normal(2.0, 0.1) @ "tau"
        return O.Vmap(oschool, in_axes=(None, None, 0))(mu, tau, sig) @ "schools"

    @G.gen
    def bare():                                   # a bare distribution under vmap: a per-particle mean, a table of scales
        mu = G.normal(0.0, 1.0) @ "mu"
        return G.normal.vmap(in_axes=(None, 0))(mu * 2.0, jnp.array(sig)) @ "xs"

    @O.gen
    def obare():
        mu = O.normal(0.0, 1.0) @ "mu"
        return O.Vmap(O.normal, in_axes=(None, 0))((mu * np.float32(2.0)).astype(np.float32), sig) @ "xs"
    k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
    runs = {}
    for deferred in (True, False):
        G.clear_caches()
        static._NO_DEFER.clear()
        keep = cmb.DEFER_MIN_WORK
        cmb.DEFER_MIN_WORK = keep if deferred else 1 << 62
        try:
            t0 = time.perf_counter()
            tr = model.simulate(k, (5.0,))
            tri, w = model.importance(G.split(G.key(seed + 1), B), C["schools", :, "y"].set(jnp.array(ys)), (5.0,))
            s, _ = model.assess(tri.get_choices(), (5.0,))
            trb = bare.simulate(k, ())
            _ = float(w.sum())
            t1 = time.perf_counter()
            tri2, w2 = model.importance(G.split(G.key(seed + 1), B), C["schools", :, "y"].set(jnp.array(ys)), (5.0,))
            _ = float(w2.sum())
            t_imp = time.perf_counter() - t1
            took = any(isinstance(e, tuple) and len(e) == 8 and e[7] for e in static._CACHE.values())
            assert took == deferred, "the call did not take the requested (deferred / loop) form"
            runs[deferred] = dict(theta=tr.get_choices()["schools", "theta"].cpu().numpy(), score=tr.get_score().cpu().numpy(),
                                  ret=tr.get_retval().cpu().numpy(), w=w.cpu().numpy(), iscore=tri.get_score().cpu().numpy(),
                                  s=s.cpu().numpy(), xs=trb.get_choices()["xs"].cpu().numpy(), bscore=trb.get_score().cpu().numpy(),
                                  t_importance=t_imp)
        finally:
            cmb.DEFER_MIN_WORK = keep
    for name in ("theta", "score", "ret", "w", "iscore", "s", "xs", "bscore"):
        assert np.array_equal(runs[True][name], runs[False][name]), name
    otr = omodel.simulate(ok, (np.float32(5.0),))
    otri, ow = omodel.importance(O.split(O.key(seed + 1), B), O.C.d({("schools", "y"): ys}), (np.float32(5.0),))
    otrb = obare.simulate(ok, ())
    d = runs[True]
    assert np.array_equal(d["theta"], otr.get_choices()["schools", "theta"]) and np.array_equal(d["score"], otr.get_score())
    assert np.array_equal(d["w"], ow) and np.array_equal(d["iscore"], otri.get_score()) and np.array_equal(d["s"], otri.get_score())
    assert np.array_equal(d["xs"], otrb.get_choices()["xs"]) and np.array_equal(d["bscore"], otrb.get_score())
    # ImportanceK over the target: B particles under ONE key
    G.clear_caches()
    tgt = G.Target(model, (5.0,), C["schools", :, "y"].set(jnp.array(ys)))
    coll = ImportanceK(tgt, k_particles=B).run_smc(G.key(seed + 2))
    ocoll = O.ImportanceK(O.Target(omodel, (np.float32(5.0),), O.C.d({("schools", "y"): ys})), B).run_smc(O.key(seed + 2))
    assert np.array_equal(coll.get_log_weights().cpu().numpy(), ocoll.get_log_weights())
    return (runs[True]["t_importance"], runs[False]["t_importance"]) if timing else None


def check_mask_combinator(B=33, T=10, n_plate=40, seed=51):
    """MaskCombinator and the masked scan sugar (ref combinators/mask.py:96-262, scan.py:1050-1150) against the
    oracle's restatement, bit for bit, under a batch of B keys:
      * `model.mask()` with ONE FLAG PER PARTICLE: simulate / importance / assess, and `update` with all four flag
        transitions (t->t, t->f, f->t, f->f: mask.py:197-210) in one launch, with and without a new constraint;
      * `model.mask().vmap()` over a plate of flags: unrolled (3) and as a counted loop (n_plate), incl. `update`;
      * `masked_iterate` / `masked_iterate_final` over T steps whose step holds a masked plate: simulate, the masked
        choices, importance under per-step observations, assess of the masked choices, and `update` under a changed
        vector of flags (the reference's "extend by unmasking")."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, _lib, numpy as jnp
    dev = _lib.get().device
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n_ = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    eq = lambda a, b: np.array_equal(n_(a), np.asarray(b))
    rng = np.random.default_rng(seed)

    # -- one flag per particle --------------------------------------------------------------------------------------
    @G.gen
    def inner(x):
        z = G.normal(x, 1.0) @ "z"
        y = G.normal(z * 0.5, 2.0) @ "y"
        return z + y

    @O.gen
    def oinner(x):
        z = O.normal(x, np.float32(1.0)) @ "z"
        y = O.normal((z * np.float32(0.5)).astype(np.float32), np.float32(2.0)) @ "y"
        return (z + y).astype(np.float32)
    m, om = inner.mask(), O.MaskCombinator(oinner)
    pre = rng.random(B) < 0.5
    post = rng.random(B) < 0.5
    xs = rng.normal(size=B).astype(np.float32)
    k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
    tr, otr = m.simulate(k, (t_(pre), t_(xs))), om.simulate(ok, (pre, xs))
    assert eq(tr.get_score(), otr.get_score()) and eq(tr.inner.get_score(), otr.inner.get_score())
    assert eq(tr.get_retval().value, otr.get_retval().value) and eq(tr.get_retval().flag, pre)
    ch = tr.get_choices()
    assert eq(ch["z"].value, otr.get_choices()[("z",)].value) and eq(ch["z"].flag, pre)
    ys = rng.normal(size=B).astype(np.float32)
    tri, w = m.importance(k, C["y"].set(t_(ys)), (t_(pre), t_(xs)))
    otri, ow = om.importance(ok, O.C.d({("y",): ys}), (pre, xs))
    assert eq(w, ow) and eq(tri.get_score(), otri.get_score())
    s_, r_ = m.assess(tri.get_choices(), (t_(pre), t_(xs)))
    assert eq(s_, otri.get_score()) and eq(r_.value, otri.get_retval().value)
    k2, ok2 = G.split(G.key(seed + 1), B), O.split(O.key(seed + 1), B)
    xs2 = rng.normal(size=B).astype(np.float32)
    for con, ocon in ((C.n(), O.ChoiceMap()), (C["z"].set(t_(ys)), O.C.d({("z",): ys}))):
        new, wu, _, bwd = m.update(k2, tri, con, (Diff.unknown_change(t_(post)), Diff.unknown_change(t_(xs2))))
        onew, owu, odis = om.update(ok2, otri, ocon, (post, xs2))
        assert eq(wu, owu), (n_(wu)[:4], owu[:4])
        assert eq(new.get_score(), onew.get_score()) and eq(new.inner.get_score(), onew.inner.get_score())
        assert eq(new.get_retval().flag, post)
        if not ocon.static_is_empty():
            assert eq(bwd["z"].value, odis[("z",)].value) and eq(bwd["z"].flag, post)

    # -- a plate of masked elements: unrolled and as a loop ------------------------------------------------------------
    for n in (3, n_plate):
        flags = rng.random(n) < 0.6
        locs = np.linspace(-1.0, 1.0, n).astype(np.float32)
        pm, opm = inner.mask().vmap(in_axes=(0, 0)), O.Vmap(O.MaskCombinator(oinner), in_axes=(0, 0))
        ptr, optr = pm.simulate(k, (jnp.array(flags), jnp.array(locs))), opm.simulate(ok, (flags, locs))
        assert eq(ptr.get_score(), optr.get_score())
        assert eq(ptr.inner.get_score(), optr.inner.get_score())              # the masked per-element scores
        assert eq(ptr.get_retval().value, optr.get_retval().value)
        assert eq(ptr.get_choices()[:, "y"].flag, np.broadcast_to(flags, (B, n)))
        flags2 = rng.random(n) < 0.6
        yv = rng.normal(size=n).astype(np.float32)
        new, wu, _, _ = pm.update(k2, ptr, C[:, "y"].set(jnp.array(yv)),
                                  (Diff.unknown_change(jnp.array(flags2)), Diff.no_change(jnp.array(locs))))
        onew, owu, _ = opm.update(ok2, optr, O.C.d({("y",): yv}), (flags2, locs))
        assert eq(wu, owu) and eq(new.get_score(), onew.get_score())
        bm, obm = G.normal.mask().vmap(in_axes=(0, None, 0)), O.Vmap(O.MaskCombinator(O.normal), in_axes=(0, None, 0))
        sc = np.linspace(0.5, 2.0, n).astype(np.float32)
        btr, obtr = bm.simulate(k, (jnp.array(flags), 0.5, jnp.array(sc))), obm.simulate(ok, (flags, np.float32(0.5), sc))
        assert eq(btr.get_score(), obtr.get_score()) and eq(btr.get_retval().value, obtr.get_retval().value)

    # -- the masked scans ------------------------------------------------------------------------------------------
    masks = np.array([True, False, True])

    @G.gen
    def step(x):
        _ = G.normal.mask().vmap(in_axes=(0, None, None))(jnp.array(masks), x, 1.0) @ "rats"
        z = G.normal(x, 0.5) @ "z"
        return z

    @O.gen
    def ostep(x):
        _ = O.Vmap(O.MaskCombinator(O.normal), in_axes=(0, None, None))(masks, x, np.float32(1.0)) @ "rats"
        z = O.normal(x, np.float32(0.5)) @ "z"
        return z
    for steps in (3, T, 40):
        flags = np.arange(steps) < (steps // 2)
        flags2 = np.arange(steps) < (steps // 2 + 1)
        zs = np.linspace(-1, 1, steps).astype(np.float32)
        for every in (False, True):
            model = step.masked_iterate() if every else step.masked_iterate_final()
            omodel = O.MaskedIterate(ostep, every)
            tr = model.simulate(k, (0.25, jnp.array(flags)))
            otr = omodel.simulate(ok, (np.float32(0.25), flags))
            assert eq(tr.get_score(), otr.get_score()) and eq(tr.get_retval(), otr.get_retval())
            ch, och = tr.get_choices(), otr.get_choices()
            assert eq(ch["z"].value, och[("z",)].value) and eq(ch["z"].flag, np.broadcast_to(och[("z",)].flag, (B, steps)))
            assert eq(ch["rats"].value, och[("rats",)].value)
            assert eq(ch["rats"].flag, np.broadcast_to(och[("rats",)].flag, (B, steps, 3)))
            new, wu, _, _ = model.update(k2, tr, C.n(), (Diff.no_change(0.25), Diff.unknown_change(jnp.array(flags2))))
            onew, owu, _ = omodel.update(ok2, otr, O.ChoiceMap(), (np.float32(0.25), flags2))
            assert eq(wu, owu) and eq(new.get_score(), onew.get_score())
            tri, wi = model.importance(k, C["z"].set(jnp.array(zs)), (0.25, jnp.array(flags)))
            otri, owi = omodel.importance(ok, O.C.d({("z",): zs}), (np.float32(0.25), flags))
            assert eq(wi, owi) and eq(tri.get_score(), otri.get_score())
            s_, _ = model.assess(tri.get_choices(), (0.25, jnp.array(flags)))
            assert eq(s_, otri.get_score())


def check_update_under_changed_table_arguments(B=9, n=24, seed=61):
    """`Update` whose ARGUMENT change is a launch-uniform TABLE (more than 16 elements: read with a run-time index in a
    plate's / scan's loop, or by `means[idx]`): every element is re-scored under the new table (ref vmap.py:236-275,
    scan.py:417-503, incremental.py: an UnknownChange argument).  One plate, a plate of plates (the reference's
    masking notebook's image model, without the mask), a scan over a table, a table indexed by a sampled integer."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, numpy as jnp
    n_ = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    eq = lambda a, b: np.array_equal(n_(a), np.asarray(b))
    rng = np.random.default_rng(seed)

    @G.gen
    def px(m):
        return G.normal(m, 1.0) @ "pixel"

    @O.gen
    def opx(m):
        return O.normal(m, np.float32(1.0)) @ "pixel"
    k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
    k2, ok2 = G.split(G.key(seed + 1), B), O.split(O.key(seed + 1), B)
    m0, m1 = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    pl, opl = px.vmap(in_axes=(0,)), O.Vmap(opx, in_axes=(0,))
    tr, otr = pl.simulate(k, (jnp.array(m0),)), opl.simulate(ok, (m0,))
    new, w, _, _ = tr.edit(k2, G.Update(C.n()), (Diff(jnp.array(m1), G.UnknownChange),))
    onew, ow, _ = opl.update(ok2, otr, O.ChoiceMap(), (m1,))
    assert np.any(n_(w) != 0.0) and eq(w, ow) and eq(new.get_score(), onew.get_score())
    # a plate of plates
    a = max(3, n // 4)
    M0, M1 = rng.normal(size=(a, n)).astype(np.float32), rng.normal(size=(a, n)).astype(np.float32)
    pp, opp = px.vmap(in_axes=(0,)).vmap(in_axes=(0,)), O.Vmap(O.Vmap(opx, in_axes=(0,)), in_axes=(0,))
    tr, otr = pp.simulate(k, (jnp.array(M0),)), opp.simulate(ok, (M0,))
    assert eq(tr.get_score(), otr.get_score())
    new, w, _, _ = tr.edit(k2, G.Update(C.n()), (Diff(jnp.array(M1), G.UnknownChange),))
    onew, ow, _ = opp.update(ok2, otr, O.ChoiceMap(), (M1,))
    assert np.any(n_(w) != 0.0) and eq(w, ow) and eq(new.get_score(), onew.get_score())
    # a scan whose scanned input is a table
    @G.gen
    def step(c, x):
        z = G.normal(c + x, 1.0) @ "z"
        return z, z

    @O.gen
    def ostep(c, x):
        z = O.normal((c + x).astype(np.float32), np.float32(1.0)) @ "z"
        return z, z
    sc, osc = step.scan(n=n), O.Scan(ostep, n)
    tr, otr = sc.simulate(k, (0.0, jnp.array(m0))), osc.simulate(ok, (np.float32(0.0), m0))
    new, w, _, _ = tr.edit(k2, G.Update(C.n()), (Diff.no_change(0.0), Diff(jnp.array(m1), G.UnknownChange)))
    onew, ow = O.scan_edit(osc, ok2, otr, (np.float32(0.0), m1), update=O.ChoiceMap())
    assert np.any(n_(w) != 0.0) and eq(w, ow) and eq(new.get_score(), onew.get_score())
    # a table indexed by a sampled integer
    @G.gen
    def pick(means):
        i = G.categorical(logits=jnp.zeros(n)) @ "i"
        return G.normal(means[i], 1.0) @ "x"

    @O.gen
    def opick(means):
        i = O.categorical(np.zeros(n, np.float32)) @ "i"
        return O.normal(np.asarray(means, np.float32)[i], np.float32(1.0)) @ "x"
    tr, otr = pick.simulate(k, (jnp.array(m0),)), opick.simulate(ok, (m0,))
    assert eq(tr.get_score(), otr.get_score())
    new, w, _, _ = tr.edit(k2, G.Update(C.n()), (Diff(jnp.array(m1), G.UnknownChange),))
    onew, ow, _ = opick.update(ok2, otr, O.ChoiceMap(), (m1,))
    assert np.any(n_(w) != 0.0) and eq(w, ow) and eq(new.get_score(), onew.get_score())


def check_masked_image_model(B=6, size=24, seed=71):
    """The reference's masking notebook (docs/cookbook/inactive/expressivity/masking.ipynb c24-c30):
    `single_pixel.mask().vmap(in_axes=(0,)).vmap(in_axes=(0,))` over a 2-D table of flags, then `Update` under a new
    table of flags (a growing circle): scores, the masked choices `[:, :, "pixel"]` and the weights of a chain of edits,
    bit for bit against the oracle; under a batch of keys and under ONE key."""
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, numpy as jnp
    n_ = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    eq = lambda a, b: np.array_equal(n_(a), np.asarray(b))

    @G.gen
    def single_pixel():
        pixel = G.normal(0.0, 1.0) @ "pixel"
        return pixel

    @O.gen
    def osingle_pixel():
        pixel = O.normal(np.float32(0.0), np.float32(1.0)) @ "pixel"
        return pixel
    image_model = single_pixel.mask().vmap(in_axes=(0,)).vmap(in_axes=(0,))
    oimage_model = O.Vmap(O.Vmap(O.MaskCombinator(osingle_pixel), in_axes=(0,)), in_axes=(0,))

    def circle(radius):
        y, x = np.ogrid[:size, :size]
        return np.sqrt((x - size // 2) ** 2 + (y - size // 2) ** 2) <= radius
    for batch in ((B,), ()):
        k = G.split(G.key(seed), B) if batch else G.key(seed)
        ok = O.split(O.key(seed), B) if batch else O.key(seed)
        tr, otr = image_model.simulate(k, (jnp.array(circle(size // 3)),)), oimage_model.simulate(ok, (circle(size // 3),))
        assert eq(tr.get_score(), otr.get_score())
        for i in range(4):
            m = tr.get_choices()[:, :, "pixel"]
            om = otr.get_choices()[("pixel",)]
            assert eq(m.value, om.value) and eq(m.flag, np.broadcast_to(om.flag, n_(m.flag).shape))
            k2 = G.split(G.key(seed + 1 + i), B) if batch else G.key(seed + 1 + i)
            ok2 = O.split(O.key(seed + 1 + i), B) if batch else O.key(seed + 1 + i)
            new_mask = circle(2 * i + 1)
            tr, w, _, _ = tr.edit(k2, G.Update(C.n()), (Diff(jnp.array(new_mask), G.UnknownChange),))
            otr, ow, _ = oimage_model.update(ok2, otr, O.ChoiceMap(), (new_mask,))
            assert eq(w, ow) and eq(tr.get_score(), otr.get_score()), (i, n_(w), ow)
            assert int(n_(tr.get_choices()[:, :, "pixel"].flag).sum()) == int(new_mask.sum()) * (B if batch else 1)


def check_sweep_verdict():
    """gmx_sweep_verdict (include/genmi.h): 0 when nothing happened, the plan's overflow word when no status word is set,
    2 as soon as one is; and ShardedBootstrapSweep.finish() reads exactly that word: a sweep whose peer exchange gave up
    waiting raises instead of handing back stale particles"""
    from ctypes import c_void_p
    import genjax_amd as G
    from genjax_amd import _lib, workloads
    from genjax_amd.inference.sharded import ShardedBootstrapSweep
    be = _lib.get()
    dev = be.device
    for overflow, words, want in ((0, [0, 0], 0), (1, [0, 0], 1), (0, [], 0), (1, [0, 7, 0], 2), (0, [0, 0, 0, 1], 2), (5, [], 5)):
        ov = torch.tensor([overflow], dtype=torch.int64, device=dev)
        ws = [torch.tensor([w], dtype=torch.int64, device=dev) for w in words]
        out = torch.full((1,), -1, dtype=torch.int64, device=dev)
        arr = (c_void_p * max(1, len(ws)))(*[c_void_p(w.data_ptr()) for w in ws])
        be.check(be.c.gmx_sweep_verdict(be.ptr(ov), arr, len(ws), be.ptr(out), be.stream()), "gmx_sweep_verdict")
        assert int(out.item()) == want, (overflow, words, int(out.item()))
    assert be.c.gmx_sweep_verdict(be.ptr(ov), arr, 5, be.ptr(out), be.stream()) != 0          # at most four status words

    class _Solo:
        @staticmethod
        def get_rank(): return 0
        @staticmethod
        def get_world_size(): return 1
    n, T = 2048, 3
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    sw = ShardedBootstrapSweep(init, step, n, T, _Solo, always_communicate=True).prepare(G.key(3), torch.from_numpy(ys))
    sw.launch()
    sw.finish()
    assert int(sw.verdict.item()) == 0 and sw.reruns == 0
    words = sw._status_words()
    if words:            # a communicator with status words (peer-mapped exchanges): set one, fold again, finish must raise
        words[-1].fill_(1)
        sw._enqueue_verdict()
        sw._finished = False
        try:
            sw.finish()
        except RuntimeError as e:
            assert "did not arrive" in str(e)
        else:
            raise AssertionError("finish() accepted a sweep whose status word was set")
        words[-1].zero_()
    return len(words)


def check_rows_of_logits_at_one_site(B=257, J=12, seed=5):
    """Softmax regression written with ONE categorical site: `z ~ categorical(logits [J, 3])` per particle (row j /
    category k on the site key's counter j K + k: what jax.random.categorical does for logits of that shape): simulate,
    importance with the J labels given, and `update` of the weight, against the oracle bit for bit"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp, ChoiceMap as C
    xs = np.linspace(-2.0, 2.0, J).astype(np.float32)
    zs = (np.arange(J) % 3).astype(np.int32)

    @G.gen
    def model(xs):
        w = G.normal(0.0, 1.0) @ "w"
        z = G.categorical(logits=jnp.stack([w * xs, jnp.zeros_like(xs), -w * xs], axis=-1)) @ "z"
        return z

    @O.gen
    def omodel(xs):
        w = np.asarray(O.normal(np.float32(0.0), np.float32(1.0)) @ "w", np.float32)
        wx = (w[..., None] * xs).astype(np.float32)
        z = O.categorical(np.stack([wx, np.zeros_like(wx), (-wx).astype(np.float32)], axis=-1)) @ "z"
        return z
    keys, okeys = G.split(G.key(seed), B), O.split(O.key(seed), B)
    tr = G.vmap(lambda k: model.simulate(k, (xs,)))(keys)
    otr = omodel.simulate(okeys, (xs,))
    z = _np(tr.get_choices()["z"])
    assert z.shape == (B, J) and np.array_equal(z, otr.get_choices()["z"])
    assert np.array_equal(_np(tr.get_retval()), otr.get_retval())
    assert np.array_equal(_np(tr.get_score()), otr.get_score())
    tr2, w2 = G.vmap(lambda k: model.importance(k, C.kw(z=zs), (xs,)))(keys)
    otr2, ow2 = omodel.importance(okeys, O.ChoiceMap.kw(z=np.broadcast_to(zs, (B, J))), (xs,))
    assert np.array_equal(_np(w2), ow2) and np.array_equal(_np(tr2.get_score()), otr2.get_score())
    new_w = np.random.default_rng(seed).normal(size=B).astype(np.float32)
    dev = G._lib.get().device
    tr3, w3, _, _ = tr2.update(G.key(seed + 1), C.kw(w=torch.from_numpy(new_w).to(dev)))
    otr3, ow3, _ = omodel.update(O.split(O.key(seed + 1), B), otr2, O.C.d({"w": new_w}), (xs,))
    assert np.array_equal(_np(w3), ow3) and np.array_equal(_np(tr3.get_score()), otr3.get_score())
    return float(_np(w2).mean())


def check_mixture_with_latent_means(B=129, J=6, seed=8, n_comp=3):
    """A Gaussian mixture whose component means are LATENT: `mus ~ normal(0_3, 5)`, `zs ~ categorical(logits [J, 3])`,
    `ys ~ normal(mus[zs], 1)` — values computed in the model read at traced indices (tracer.SymArray: a chain of
    selects); the means also as a plate's return values (`cluster.vmap()(...) @ "clusters"`, then `means[z]`).  With
    more than 16 components the means are a long vector-valued site (one counted loop per particle; its values live in
    memory): a read at a traced index is then a search loop over the stored values (engine.StepInput._read_at).
    simulate / importance / update against the oracle, bit for bit"""
    K_ = int(n_comp)
    import genjax_amd as G
    from genjax_amd import numpy as jnp, ChoiceMap as C
    yv = np.linspace(-4.0, 4.0, J).astype(np.float32)

    @G.gen
    def mix():
        mus = G.normal(jnp.zeros(K_), 5.0) @ "mus"
        zs = G.categorical(logits=jnp.zeros((J, K_))) @ "zs"
        G.normal(mus[zs], 1.0) @ "ys"
        return zs

    @O.gen
    def omix():
        mus = O.normal(np.zeros(K_, np.float32), np.float32(5.0)) @ "mus"
        zs = O.categorical(np.zeros((J, K_), np.float32)) @ "zs"
        O.normal(np.take_along_axis(mus, zs, axis=-1), np.float32(1.0)) @ "ys"
        return zs

    @G.gen
    def cluster(c):
        return G.normal(c, 5.0) @ "mean"

    @G.gen
    def mix2(centres):
        means = cluster.vmap(in_axes=(0,))(centres) @ "clusters"
        z = G.categorical(logits=jnp.array([0.0, 0.5, -0.5])) @ "z"
        G.normal(means[z] * 0.5 + means[0], 1.0) @ "y"
        return z

    @O.gen
    def ocluster(c):
        return O.normal(c, np.float32(5.0)) @ "mean"

    @O.gen
    def omix2(centres):
        means = O.Vmap(ocluster, in_axes=(0,))(centres) @ "clusters"
        z = O.categorical(np.array([0.0, 0.5, -0.5], np.float32)) @ "z"
        picked = np.take_along_axis(means, np.asarray(z)[..., None], axis=-1)[..., 0]
        O.normal((picked * np.float32(0.5) + means[..., 0]).astype(np.float32), np.float32(1.0)) @ "y"
        return z
    keys, okeys = G.split(G.key(seed), B), O.split(O.key(seed), B)
    dev = G._lib.get().device
    tr = G.vmap(lambda k: mix.simulate(k, ()))(keys)
    otr = omix.simulate(okeys, ())
    assert np.array_equal(_np(tr.get_choices()["zs"]), otr.get_choices()["zs"])
    assert np.array_equal(_np(tr.get_choices()["ys"]), otr.get_choices()["ys"])
    assert np.array_equal(_np(tr.get_score()), otr.get_score())
    tr2, w2 = G.vmap(lambda k: mix.importance(k, C.kw(ys=yv), ()))(keys)
    otr2, ow2 = omix.importance(okeys, O.ChoiceMap.kw(ys=np.broadcast_to(yv, (B, J))), ())
    assert np.array_equal(_np(w2), ow2)
    new_mus = np.random.default_rng(seed).normal(size=(B, K_)).astype(np.float32)
    tr3, w3, _, _ = tr2.update(G.key(seed + 1), C.kw(mus=torch.from_numpy(new_mus).to(dev)))
    otr3, ow3, _ = omix.update(O.split(O.key(seed + 1), B), otr2, O.C.d({"mus": new_mus}), ())
    assert np.array_equal(_np(w3), ow3) and np.array_equal(_np(tr3.get_score()), otr3.get_score())
    centres = np.array([-3.0, 0.0, 3.0], np.float32)
    tr4, w4 = G.vmap(lambda k: mix2.importance(k, C.kw(y=1.5), (centres,)))(keys)
    otr4, ow4 = omix2.importance(okeys, O.ChoiceMap.kw(y=np.full(B, 1.5, np.float32)), (centres,))
    assert np.array_equal(_np(tr4.get_choices()["z"]), otr4.get_choices()["z"])
    assert np.array_equal(_np(w4), ow4) and np.array_equal(_np(tr4.get_score()), otr4.get_score())

    # the whole Dirichlet mixture: latent weights too (`probs=` of shape [J, 3] keeps its rows — it used to be read as
    # ONE categorical over 3 J categories)
    @G.gen
    def mix3():
        wts = G.dirichlet(jnp.ones(3)) @ "w"
        mus = G.normal(jnp.array([-2.0, 0.0, 2.0]), 1.0) @ "mus"
        zs = G.categorical(probs=jnp.stack([wts] * J)) @ "zs"
        G.normal(mus[zs], 0.5) @ "ys"
        return zs

    @O.gen
    def omix3():
        wts = np.asarray(O.dirichlet(np.ones(3, np.float32)) @ "w", np.float32)
        mus = O.normal(np.array([-2.0, 0.0, 2.0], np.float32), np.float32(1.0)) @ "mus"
        lw = O.log(wts)
        zs = O.categorical(np.broadcast_to(lw[..., None, :], lw.shape[:-1] + (J, 3)).copy()) @ "zs"
        O.normal(np.take_along_axis(mus, zs, axis=-1), np.float32(0.5)) @ "ys"
        return zs
    tr5, w5 = G.vmap(lambda k: mix3.importance(k, C.kw(ys=yv), ()))(keys)
    otr5, ow5 = omix3.importance(okeys, O.ChoiceMap.kw(ys=np.broadcast_to(yv, (B, J))), ())
    assert np.array_equal(_np(tr5.get_choices()["zs"]), otr5.get_choices()["zs"]) and np.array_equal(_np(w5), ow5)
    return float(_np(w2).mean())


def check_slices_of_a_long_per_particle_vector(B=65, N=50, seed=4):
    """`normal(rho * f(ys), 1) @ "y"` with ys a PER-PARTICLE vector of N > 16 elements (one [N, n] input slot read at a
    loop's iteration number) and f a slice: `ys[1:]`, `ys[10:40]`, `ys[1:] - ys[:-1]` (a base offset on the step read),
    `ys[::-1]`, `ys[::2]` (plain element reads) — a sliced view used to read element t of the WHOLE leaf, silently;
    the same slices of a LATENT long vector (`x ~ normal(0_N, 1)`, then `normal(x[1:] - x[:-1], 0.1)`: the stored
    values read back).  Importance weights against the oracle, bit for bit"""
    import genjax_amd as G
    from genjax_amd import ChoiceMap as C
    rng = np.random.default_rng(seed)
    ysb = np.cumsum(rng.normal(size=(B, N)), axis=1).astype(np.float32)
    dev = G._lib.get().device
    keys, okeys = G.split(G.key(seed), B), O.split(O.key(seed), B)
    cases = {
        "tail": lambda y: y[..., 1:], "head": lambda y: y[..., :-1], "mid": lambda y: y[..., 10:40],
        "diff": lambda y: y[..., 1:] - y[..., :-1], "reversed": lambda y: y[..., ::-1], "every other": lambda y: y[..., ::2],
        "tail of tail": lambda y: y[..., 2:][..., 3:],
    }
    for name, f in cases.items():
        @G.gen
        def model(ys):
            rho = G.uniform(-1.0, 1.0) @ "rho"
            G.normal(rho * f(ys), 1.0) @ "y"
            return rho

        @O.gen
        def omodel(ys):
            rho = np.asarray(O.uniform(np.float32(-1.0), np.float32(1.0)) @ "rho", np.float32)
            O.normal((rho[..., None] * np.asarray(f(ys), np.float32)).astype(np.float32), np.float32(1.0)) @ "y"
            return rho
        m = f(ysb).shape[-1]
        obs = np.linspace(-1.0, 1.0, m).astype(np.float32)
        tr, w = G.vmap(lambda k, y: model.importance(k, C.kw(y=obs), (y,)))(keys, torch.from_numpy(ysb).to(dev))
        otr, ow = omodel.importance(okeys, O.ChoiceMap.kw(y=np.broadcast_to(obs, (B, m))), (ysb,))
        assert np.array_equal(_np(w), ow), name

        @G.gen
        def walk():
            x = G.normal(np.zeros(N, np.float32), 1.0) @ "x"
            G.normal(f(x), 0.5) @ "d"
            return x[3]

        @O.gen
        def owalk():
            x = np.asarray(O.normal(np.zeros(N, np.float32), np.float32(1.0)) @ "x", np.float32)
            O.normal(np.asarray(f(x), np.float32), np.float32(0.5)) @ "d"
            return x[..., 3]
        tr2, w2 = G.vmap(lambda k: walk.importance(k, C.kw(d=obs), ()))(keys)
        otr2, ow2 = owalk.importance(okeys, O.ChoiceMap.kw(d=np.broadcast_to(obs, (B, m))), ())
        assert np.array_equal(_np(w2), ow2), (name, "latent")
        assert np.array_equal(_np(tr2.get_retval()), otr2.get_retval()), (name, "latent retval")

    # ... and as what a long scan / a large plate runs over: step t / element t reads element base + t
    @G.gen
    def step(c, x):
        z = G.normal(c * 0.5 + x, 1.0) @ "z"
        return z, z

    @G.gen
    def elem(mu, x):
        return G.normal(mu + x, 1.0) @ "v"

    @O.gen
    def ostep(c, x):
        z = O.normal((c * np.float32(0.5) + x).astype(np.float32), np.float32(1.0)) @ "z"
        return z, z

    @O.gen
    def oelem(mu, x):
        return O.normal((mu + x).astype(np.float32), np.float32(1.0)) @ "v"
    for name in ("tail", "mid"):
        f = cases[name]
        T_ = f(ysb).shape[-1]

        @G.gen
        def over(ys):
            mu = G.normal(0.0, 1.0) @ "mu"
            cT, _ = G.Scan(step, T_)(mu, f(ys)) @ "chain"
            elem.vmap(in_axes=(None, 0))(cT, f(ys)) @ "plate"
            return cT

        @O.gen
        def oover(ys):
            mu = O.normal(np.float32(0.0), np.float32(1.0)) @ "mu"
            cT, _ = O.Scan(ostep, T_)(mu, f(ys)) @ "chain"          # (the oracle's step / plate axis is the last one)
            O.Vmap(oelem, in_axes=(None, 0))(cT, f(ys)) @ "plate"
            return cT
        tr3 = G.vmap(lambda k, y: over.simulate(k, (y,)))(keys, torch.from_numpy(ysb).to(dev))
        otr3 = oover.simulate(okeys, (ysb,))
        assert np.array_equal(_np(tr3.get_score()), otr3.get_score()), (name, "scan and plate over the slice")
        assert np.array_equal(_np(tr3.get_retval()), otr3.get_retval()), (name, "scan and plate over the slice: retval")
    return len(cases)


def check_changed_per_particle_vector_argument(B=65, N=30, seed=12):
    """`update` with a CHANGED argument that is one vector of N > 16 elements per particle, mapped over by a large plate
    and scanned over by a long scan: every element / step reads its own element at the loop's iteration number — a node
    of its own, which the change propagation did not see as changed (the elements kept their old scores: wrong weights,
    silently; found by giving the random-model grammar per-particle tables).  Weights and scores against the oracle"""
    import genjax_amd as G
    from genjax_amd import Diff
    rng = np.random.default_rng(seed)
    xs1, xs2 = (rng.normal(size=(B, N)).astype(np.float32) for _ in range(2))
    dev = G._lib.get().device

    @G.gen
    def elem(mu, x):
        return G.normal(mu + x, 1.0) @ "v"

    @G.gen
    def step(c, x):
        z = G.normal(c * 0.5 + x, 1.0) @ "z"
        return z, z

    @G.gen
    def model(xs):
        mu = G.normal(0.0, 1.0) @ "mu"
        elem.vmap(in_axes=(None, 0))(mu, xs) @ "plate"
        cT, _ = G.Scan(step, N)(mu, xs) @ "chain"
        G.normal(mu + xs * 0.5, 2.0) @ "vec"
        return cT

    @O.gen
    def oelem(mu, x):
        return O.normal((mu + x).astype(np.float32), np.float32(1.0)) @ "v"

    @O.gen
    def ostep(c, x):
        z = O.normal((c * np.float32(0.5) + x).astype(np.float32), np.float32(1.0)) @ "z"
        return z, z

    @O.gen
    def omodel(xs):
        mu = np.asarray(O.normal(np.float32(0.0), np.float32(1.0)) @ "mu", np.float32)
        O.Vmap(oelem, in_axes=(None, 0))(mu, xs) @ "plate"
        cT, _ = O.Scan(ostep, N)(mu, xs) @ "chain"
        O.normal((mu[..., None] + xs * np.float32(0.5)).astype(np.float32), np.float32(2.0)) @ "vec"
        return cT
    k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
    t1, t2 = torch.from_numpy(xs1).to(dev), torch.from_numpy(xs2).to(dev)
    tr, otr = model.simulate(k, (t1,)), omodel.simulate(ok, (xs1,))
    assert np.array_equal(_np(tr.get_score()), otr.get_score())
    k2, ok2 = G.split(G.key(seed + 1), B), O.split(O.key(seed + 1), B)
    new, w, _, _ = model.update(k2, tr, G.ChoiceMap.empty(), (Diff(t2, G.UnknownChange),))
    onew, ow, _ = omodel.update(ok2, otr, O.ChoiceMap(), (xs2,))
    assert np.array_equal(_np(w), ow), "update weight under a changed per-particle vector"
    assert np.array_equal(_np(new.get_score()), onew.get_score())
    assert float(np.abs(ow).max()) > 1.0           # (the weights are not trivially zero)

    # ... and a [B, n, n2] argument of a plate of plates (engine.StepInput2: rows picked by the loops' own numbers)
    n1, n2 = 3, 20
    zs1, zs2 = (rng.normal(size=(B, n1, n2)).astype(np.float32) for _ in range(2))

    @G.gen
    def rows(mu, xs):
        return elem.vmap(in_axes=(None, 0))(mu, xs) @ "r"

    @G.gen
    def model2(zs):
        mu = G.normal(0.0, 1.0) @ "mu"
        rows.vmap(in_axes=(None, 0))(mu, zs) @ "pp"
        return mu

    @O.gen
    def oelem2(mu, x):          # (the oracle's plates keep their axes behind the batch: the shared value is padded)
        mu = np.asarray(mu, np.float32)
        while mu.ndim < np.ndim(x):
            mu = mu[..., None]
        return O.normal((mu + x).astype(np.float32), np.float32(1.0)) @ "v"

    @O.gen
    def orows(mu, xs):
        return O.Vmap(oelem2, in_axes=(None, 0))(mu, xs) @ "r"

    @O.gen
    def omodel2(zs):
        mu = np.asarray(O.normal(np.float32(0.0), np.float32(1.0)) @ "mu", np.float32)
        O.Vmap(orows, in_axes=(None, 0))(mu, zs) @ "pp"
        return mu
    u1, u2 = torch.from_numpy(zs1).to(dev), torch.from_numpy(zs2).to(dev)
    tr2, otr2 = model2.simulate(k, (u1,)), omodel2.simulate(ok, (zs1,))
    assert np.array_equal(_np(tr2.get_score()), otr2.get_score())
    new2, w2, _, _ = model2.update(k2, tr2, G.ChoiceMap.empty(), (Diff(u2, G.UnknownChange),))
    onew2, ow2, _ = omodel2.update(ok2, otr2, O.ChoiceMap(), (zs2,))
    assert np.array_equal(_np(w2), ow2), "update weight under a changed [B, n, n2] argument"
    assert np.array_equal(_np(new2.get_score()), onew2.get_score())
    return float(ow.mean())


def check_gather_by_index_vector(B=33, seed=21):
    """`normal(means[zs] + s, 1) @ "y"`: the vectorised mixture likelihood with the assignments GIVEN — `means` a table
    of 3 (registers) or 20 (memory) components, `zs` an integer vector of 8 or 30 elements given as a launch-uniform
    table or one per particle: element j reads means at zs[j] (lazily, in the site's loop, when zs is long).
    importance, and `update` under changed assignments, against the oracle bit for bit"""
    import genjax_amd as G
    from genjax_amd import ChoiceMap as C, Diff, numpy as jnp
    rng = np.random.default_rng(seed)
    dev = G._lib.get().device
    n = 0
    for N, K in ((8, 3), (30, 3), (30, 20)):
        zs, zs2 = (rng.integers(0, K, size=(B, N)).astype(np.int32) for _ in range(2))
        zt, zt2 = (rng.integers(0, K, size=N).astype(np.int32) for _ in range(2))
        means = np.linspace(-3.0, 3.0, K).astype(np.float32)
        obs = rng.normal(size=N).astype(np.float32)

        @G.gen
        def model(means, zs):
            s = G.normal(0.0, 1.0) @ "s"
            G.normal(means[zs] + s, 1.0) @ "y"
            return s

        @O.gen
        def omodel(means, zs):
            s = np.asarray(O.normal(np.float32(0.0), np.float32(1.0)) @ "s", np.float32)
            O.normal((means[zs] + s[..., None]).astype(np.float32), np.float32(1.0)) @ "y"
            return s
        k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
        for za, zb, oa, ob in ((torch.from_numpy(zs).to(dev), torch.from_numpy(zs2).to(dev), zs, zs2),
                               (jnp.array(zt), jnp.array(zt2), zt, zt2)):
            tr, w = model.importance(k, C.kw(y=obs), (jnp.array(means), za))
            otr, ow = omodel.importance(ok, O.ChoiceMap.kw(y=np.broadcast_to(obs, (B, N))), (means, oa))
            assert np.array_equal(_np(w), ow), (N, K, "importance")
            new, wu, _, _ = model.update(G.split(G.key(seed + 1), B), tr, C.empty(),
                                         (Diff.no_change(jnp.array(means)), Diff(zb, G.UnknownChange)))
            onew, owu, _ = omodel.update(O.split(O.key(seed + 1), B), otr, O.ChoiceMap(), (means, ob))
            assert np.array_equal(_np(wu), owu) and np.array_equal(_np(new.get_score()), onew.get_score()), (N, K, "update")
            n += 1
    return n


def check_sweep_with_vector_observations(n=2048, T=4, m=24, seed=3):
    """BootstrapSweep over a state-space model whose step emits a VECTOR of m observations (`y_t ~ normal(x_t * c, 1)`,
    ys of shape [T, m]): the step program holds a long vector-valued site (one counted loop per particle for m > 16) —
    log-ML and the final resampled states against the oracle's sweep, bit for bit"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from genjax_amd.inference.smc import BootstrapSweep
    tab = np.linspace(0.5, 1.5, m).astype(np.float32)

    @G.gen
    def init():
        x = G.normal(0.0, 1.0) @ "x"
        G.normal(x * jnp.array(tab), 1.0) @ "y"
        return x

    @G.gen
    def step(xp):
        x = G.normal(0.9 * xp, 0.5) @ "x"
        G.normal(x * jnp.array(tab), 1.0) @ "y"
        return x

    @O.gen
    def oinit():
        x = O.normal(np.float32(0.0), np.float32(1.0)) @ "x"
        O.normal((np.asarray(x, np.float32)[..., None] * tab).astype(np.float32), np.float32(1.0)) @ "y"
        return x

    @O.gen
    def ostep(xp):
        x = O.normal((np.float32(0.9) * xp).astype(np.float32), np.float32(0.5)) @ "x"
        O.normal((np.asarray(x, np.float32)[..., None] * tab).astype(np.float32), np.float32(1.0)) @ "y"
        return x
    ys = np.random.default_rng(seed).normal(size=(T, m)).astype(np.float32)
    sw = BootstrapSweep(init, step, n, T).prepare(G.key(seed), torch.from_numpy(ys))
    sw.launch()
    ref = oracle_bootstrap_sweep(oinit, ostep, n, T, ys, O.key(seed))
    assert sw.log_ml() == ref["log_ml"], (sw.log_ml(), ref["log_ml"])
    assert [int(t) for t in sw.totals.cpu().numpy().view(np.uint64)] == [h["total"] for h in ref["hist"]]
