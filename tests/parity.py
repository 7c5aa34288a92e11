"""Shared parity drivers: run the same workload through genjax_amd (HIP, or the
CPU hostsim harness in `-m "not gpu"` tests) and through the CPU oracle."""
from __future__ import annotations

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
        anc = O.ancestors(kind, k_res, cdf)
        log_ml += O.log_ml_increment(M, total, shift, n)
        hist.append(dict(x=x, lw=lw, cdf=cdf, total=total, M=M, anc=anc))
    return dict(log_ml=log_ml, x=x, lw=lw, anc=anc, hist=hist)


def check_lgssm_sweep(n=4096, T=5, seed=314159, capture=False, specialize=True):
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference.smc import BootstrapSweep
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    sw = BootstrapSweep(init, step, n, T, specialize=specialize).prepare(G.key(seed), torch.from_numpy(ys))
    if capture:
        sw.capture()
    sw.launch()
    log_ml = sw.log_ml()
    x, lw, anc = sw.state()
    oi, os_ = workloads.make_lgssm(O)
    ref = oracle_bootstrap_sweep(oi, os_, n, T, ys, O.key(seed))
    return dict(
        log_ml=log_ml, log_ml_oracle=ref["log_ml"], kalman=workloads.kalman_log_ml(ys),
        ancestors_equal=bool(np.array_equal(anc.cpu().numpy(), ref["anc"])),
        x_equal=bool(np.array_equal(x.cpu().numpy(), ref["x"])),
        lw_max_abs_diff=float(np.max(np.abs(lw.cpu().numpy() - ref["lw"]))),
        totals_equal=bool(np.array_equal(sw.totals.cpu().numpy().view(np.uint64),
                                         np.array([h["total"] for h in ref["hist"]], dtype=np.uint64))),
    )


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
