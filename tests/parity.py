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
