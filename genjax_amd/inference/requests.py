"""`genjax.inference.requests` (src/genjax/inference/requests.py): `Rejuvenate` (hot path) and
`HMC` / `SafeHMC` (src/genjax/_src/inference/requests/hmc.py:138-223).

An HMC edit is ONE launch: the L leapfrog steps, the L + 1 gradients of the model's `assess` with
respect to the selected choices (reverse-mode over the site-program IR, genjax_amd/autodiff.py),
the momenta draw and the final re-scoring are one straight-line program per particle — where the
reference traces `jax.grad` inside `lax.scan`.

Followed literally, including what looks like a slip: the leapfrog carry returns the gradient it was
GIVEN (`return (new_trace, values, gradient, momenta)`, hmc.py:195), so the first half-kick of every
step uses the gradient at the INITIAL point, the second half-kick the fresh one.
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from .. import tracer as T
from ..autodiff import grad
from ..core.choice_map import ChoiceMap, Selection
from ..core.generative import Diff, DiffAnnotate, EditRequest
from ..engine import Compiled, Flat, Tracing, leaf_spec, resolve, unflatten
from ..static import (Rejuvenate, _broadcast_score, _build_trace, _Ctx, _emit_rec, _gfkey, _rec_score, _selkey,
                      _trace_tree, call_gen_fn)
from ..tracer import Expr
from ..engine import new_cache as _new_program_cache

_CACHE = _new_program_cache()


class HMC(EditRequest):
    """Hamiltonian Monte Carlo move on the selected addresses: leapfrog with step `eps`, `L` steps;
    the returned weight is the accept-reject log-ratio alpha (hmc.py:141-214)."""
    __match_args__ = ("selection", "eps", "L")

    def __init__(self, selection: Selection, eps, L: int = 10):
        self.selection, self.eps, self.L = selection, eps, int(L)

    def edit(self, key, tr, argdiffs):
        assert Diff.static_check_no_change(argdiffs)              # hmc.py:161
        return _run_hmc(self, key, tr, argdiffs)


def SafeHMC(selection: Selection, eps, L: int = 10):
    """hmc.py:217-227: HMC whose retdiff is asserted unchanged."""
    def retdiff_assertion(retdiff):
        assert Diff.static_check_no_change(retdiff)
        return retdiff
    return DiffAnnotate(HMC(selection, eps, L), retdiff_fn=retdiff_assertion)


# ---------------------------------------------------------------------------
def _prev_choices(p) -> ChoiceMap:
    if "value" in p:
        return ChoiceMap.choice(p["value"])
    inner = p["vmap"] if "vmap" in p else p
    cm = ChoiceMap.empty()
    for a, s in inner["sub"].items():
        cm = cm.set(a, _prev_choices(s))
    return cm


def _prev_score(p):
    """Trace.get_score() of the symbolic previous trace, in the order the host classes sum."""
    if "value" in p or "vmap" in p:
        return p["score"].value
    acc = None
    for s in p["sub"].values():
        v = _prev_score(s)
        acc = v if acc is None else acc + v
    return acc if acc is not None else T.lift(0.0)


def _is_float(v) -> bool:
    if isinstance(v, np.ndarray):
        return all(isinstance(x, Expr) and x.dtype == "f32" for x in v.reshape(-1))
    return isinstance(v, Expr) and v.dtype == "f32"


def _flat(v) -> list:
    return list(v.reshape(-1)) if isinstance(v, np.ndarray) else [v]


def _run_hmc(req: HMC, key, trace, argdiffs):
    from ..distributions import normal as _normal
    be = _lib.get()
    gen_fn = trace.get_gen_fn()
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    flat = Flat()
    atree = flat.add(args)
    ptree = flat.add(_trace_tree(trace))
    etree = flat.add(req.eps)
    batch = tuple(trace.batch_shape)
    specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
    ck = (_gfkey(gen_fn), atree, ptree, etree, specs, _selkey(req.selection), req.L, len(batch))
    ent = _CACHE.get(ck)
    if ent is None:
        tr = Tracing(len(batch))
        ctx = _Ctx(tr)
        g = tr.graph
        with T.tracing(g):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
            sargs = unflatten(atree, lambda j: syms[j].value)
            sprev = unflatten(ptree, lambda j: syms[j])
            eps = T.as_float(unflatten(etree, lambda j: syms[j].value))
            chm_all = _prev_choices(sprev)                        # address -> Sym
            sel_addrs = sorted((a for a in chm_all.addresses()
                                if req.selection[a] and _is_float(chm_all[a].value)), key=repr)
            if not sel_addrs:
                raise ValueError("HMC: the selection holds no differentiable choice")
            values = {a: chm_all[a].value for a in sel_addrs}
            # a selected vector-valued site of MORE than 16 elements: positions, momenta and gradients are vectors in
            # memory (recipes over reads of the launch's own outputs, tracer.LazyVec), every stage of the leapfrog one
            # counted loop per vector-valued site that reads them (static._vector_site_loop stores d term_j / d v_j)
            long_addrs = [a for a in sel_addrs if T._long_vector(values[a])]
            for a in sel_addrs:
                if a not in long_addrs and isinstance(values[a], np.ndarray) and values[a].size > 16:
                    raise NotImplementedError(
                        f"HMC on the vector-valued site {a!r} of {values[a].size} elements with {values[a].ndim} axes: a selected "
                        "site beyond 16 elements is ONE long axis per particle")

            def model_score_and_grads(vals):
                con = chm_all
                gvecs = {}
                for a in sel_addrs:
                    if a in long_addrs:
                        gvecs[a] = T.GradVec(vals[a])
                        con = con.set(a, gvecs[a])
                    else:
                        con = con.set(a, vals[a])
                ctx.store_sites = False
                leaves = [x for a in sel_addrs if a not in long_addrs for x in _flat(vals[a])]
                ctx.grad_wrt = leaves          # (long vector sites accumulate d score / d leaf inside their own loops)
                ctx.grad_vecs = list(gvecs.values())
                try:
                    _, _, _, s = call_gen_fn(ctx, "assess", gen_fn, None, sargs, con, None, None, None, ())
                finally:
                    ctx.grad_wrt = None
                    ctx.grad_vecs = None
                gs = grad(T.as_float(s), leaves) if leaves else []
                out, k = {}, 0
                for a in long_addrs:
                    gv = gvecs[a]
                    if gv.consumed != len(gv.reads) or gv.gathers_consumed != len(gv.gathers) or not gv.contribs:
                        raise NotImplementedError(
                            f"HMC on the vector-valued site {a!r} of {gv.n} elements: the model reads its elements other than "
                            "element by element in a loop of its own length (a static or traced index, a plate over them) — "
                            "the gradient of a long vector is taken through vector-valued sites and sums that loop over it "
                            "(`normal(theta, sigma) @ 'y'`, `normal(a * theta + b, s)`, `jnp.sum(theta)`, `jnp.mean(theta ** 2)`); "
                            "select at most 16 elements otherwise")
                    # the adjoint each consuming site's score reaches the model score with (1 for a plain sum of site scores)
                    adjs = grad(T.as_float(s), [Expr(sv) for sv, _ in gv.contribs])
                    tot = None
                    for (_, alias), adj in reversed(list(zip(gv.contribs, adjs))):      # reverse mode: later sites first
                        term = alias * adj
                        tot = term if tot is None else tot + term
                    out[a] = tot
                for a in sel_addrs:
                    if a in long_addrs:
                        continue
                    v = vals[a]
                    if isinstance(v, np.ndarray):
                        arr = np.empty(v.size, dtype=object)
                        arr[:] = gs[k:k + v.size]
                        out[a] = arr.reshape(v.shape)
                        k += v.size
                    else:
                        out[a] = gs[k]
                        k += 1
                return out

            original_model_score = _prev_score(sprev)                        # tr.get_score()
            grad0 = model_score_and_grads(values)                            # selection_gradient (hmc.py:164)
            kexpr = Expr(g.add("LDKEY", dtype="key"))
            sub_key = Expr(g.add("KDERIVE", (kexpr.node,), imm=1, dtype="key"))   # key, sub_key = split(key)
            momenta = {}
            mom_scores = []
            for i, a in enumerate(sel_addrs):                                # sample_momenta (hmc.py:119-130)
                ki = Expr(g.add("KDERIVE", (sub_key.node,), imm=i, dtype="key"))
                if a in long_addrs:            # J momenta from the ONE key, element j on counter j, stored; their score summed in order
                    from ..static import _vector_site_loop
                    n_a = T._long_vector(values[a])
                    out_ = _vector_site_loop(ctx, "simulate", _normal, ki, (T.LazyVec(n_a, lambda i_: 0.0), 1.0), None, None, None)
                    if out_ is None:
                        raise NotImplementedError(f"HMC on the vector-valued site {a!r}: the momenta of {n_a} elements do not run as a loop here")
                    momenta[a] = out_[1]
                    mom_scores.append(out_[3])
                    continue
                zeros = values[a] * 0.0 if isinstance(values[a], np.ndarray) else 0.0
                if isinstance(values[a], np.ndarray):
                    zeros = np.full(values[a].shape, 0.0, dtype=object)
                momenta[a] = _normal.sym_sample(ki, (zeros, 1.0))
                mom_scores.append(_normal.sym_logpdf(momenta[a], (0.0, 1.0)))
            original_momenta_score = _seq(mom_scores)
            half = eps / 2.0
            for _ in range(req.L):                                           # hmc.py:168-192
                momenta = {a: momenta[a] + half * grad0[a] for a in sel_addrs}      # the carried (initial) gradient
                values = {a: values[a] + eps * momenta[a] for a in sel_addrs}
                grads = model_score_and_grads(values)
                momenta = {a: momenta[a] + half * grads[a] for a in sel_addrs}
            # the final trace: every site re-scored at the final values (what L Updates leave behind)
            for a in long_addrs:                  # the final positions, evaluated once: the new trace's values
                values[a] = _store_lazy(tr, values[a], T._long_vector(values[a]))
            con = chm_all
            for a in sel_addrs:
                con = con.set(a, values[a])
            ctx.store_sites = True
            ctx.mark_changed([values[a] for a in sel_addrs])
            rec, retval, _, _ = call_gen_fn(ctx, "generate", gen_fn, None, sargs, con, None, None, None, ())
            ret_changed = ctx.args_changed(retval)                 # the retdiff the L Updates would report
            final_model_score = _rec_score(rec)
            final_terms = []
            for a in sel_addrs:
                if a in long_addrs:
                    from ..static import _vector_site_loop
                    ctx.store_sites = False
                    out_ = _vector_site_loop(ctx, "assess", _normal, None, (0.0, 1.0), momenta[a] * -1.0, None, None)
                    ctx.store_sites = True
                    final_terms.append(out_[3])
                else:
                    final_terms.append(_normal.sym_logpdf(momenta[a] * -1.0, (0.0, 1.0)))
            final_momenta_score = _seq(final_terms)
            alpha = final_model_score - original_model_score + final_momenta_score - original_momenta_score
            otree = _emit_rec(tr, rec)
            wo = tr.emit_output(alpha)
        ent = (Compiled(tr), otree, wo, ret_changed)
        _CACHE[ck] = ent
    comp, otree, wo, ret_changed = ent
    outs = comp.run(flat.leaves, batch, key)
    new_tr = _build_trace(otree, outs, flat.leaves, args)
    w = _broadcast_score(resolve(wo, outs, flat.leaves), batch, be.device)
    retdiff = Diff.unknown_change(new_tr.get_retval()) if ret_changed else Diff.no_change(new_tr.get_retval())
    return new_tr, w, retdiff, HMC(req.selection, req.eps, req.L)


def _store_lazy(tr, vec, n):
    """a long vector kept as a recipe, evaluated ONCE: a counted loop stores element j; what comes back reads that output"""
    g = tr.graph
    g.loop_begin(n)
    with T.tracing(g):
        t = Expr(g.add("LDT", dtype="i32"))
        origin = tr.store_step(T.as_float(T._elem(vec, t)), n)
    g.loop_end()
    return tr.alias_step_input(origin, "f32", n)


def _seq(terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


__all__ = ["Rejuvenate", "HMC", "SafeHMC"]
