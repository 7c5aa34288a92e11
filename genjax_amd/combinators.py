"""Plate and sequence combinators: `Vmap` / `repeat` / `Scan` (simulate / generate / assess).

Reference: src/genjax/_src/generative_functions/combinators/vmap.py:180-218
(`sub_keys = split(key, n)`, inner GFI per index, score / weight summed over
the plate) and repeat.py:28-42.  Used as callees of a `@gen` function:

    thetas = school.vmap(in_axes=(None, None, 0))(mu, tau, sigmas) @ "schools"

The plate is unrolled at trace time (plates on this path are small — eight
schools, a handful of mixture components; a plate over the DATA is the
particle axis itself).  Inner addresses keep their names; their values gain a
trailing plate axis: `chm["schools", "theta"]` has shape [N, n], and the
reference's slice spelling `chm["schools", :, "theta"]` addresses the same
entry.  `edit_index` / `IndexRequest` are next-tier.

`Scan` (scan.py:200-294, 638-664): `kernel.scan(n=T)((carry, xs))` runs the kernel T times, threading the
carry; the key is CHAINED, key_t = fold_in(key_{t-1}, t) (scan.py:213); score / weight = sum over steps;
values gain a leading step axis ([N, T]); the return value is (final carry, stacked outputs).  Also unrolled
at trace time: for short sequences inside one model (a whole 100-step SSM belongs in smc.BootstrapSweep).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .core.choice_map import ChoiceMap
from .core.generative import GenerativeFunction
from . import tracer as T
from .tracer import Expr


SCAN_UNROLL_MAX = 16      # longer scans run as a counted loop in the site program (Scan._trace_loop)
VMAP_UNROLL_MAX = 16      # larger plates run as a counted loop too (Vmap._trace_loop): one iteration per element


def _loop_at(v, t, what="a plate of more than 16 elements"):
    """element t (the iteration number of a counted loop) of a mapped argument / per-element constraint / previous
    value: a launch-uniform table, or a per-particle [n, T] leaf read step by step"""
    from .engine import StepInput, Sym
    from .numpy import RuntimeTable, TableArray
    if isinstance(v, Sym):
        v = v.value
    if v is None:
        return None
    if isinstance(v, tuple):
        return tuple(_loop_at(x, t, what) for x in v)
    if isinstance(v, dict):
        return {k: _loop_at(x, t, what) for k, x in v.items()}
    if isinstance(v, (RuntimeTable, TableArray, StepInput)):
        return v[t]
    if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
        return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
    raise NotImplementedError(f"{what}: mapped arguments, per-element constraints and previous values must be "
                              "launch-uniform vectors (tables) or per-particle [n, T] arrays")


def _shape_of(tree):
    """structure of a flattened carry, without the leaf numbering"""
    k = tree[0]
    if k in ("none", "leaf"):
        return k
    if k in ("tuple", "list"):
        return (k, tuple(_shape_of(x) for x in tree[1]))
    if k == "dict":
        return (k, tuple((a, _shape_of(x)) for a, x in tree[1].items()))
    return (k, tree[1])


def _axis_len(a):
    if isinstance(a, np.ndarray):
        return a.shape[0] if a.ndim else None
    if isinstance(a, (list, tuple)):
        return len(a)
    return None


def _take(a, j):
    if isinstance(a, np.ndarray):
        v = a[j]
        return v.item() if isinstance(v, np.ndarray) and v.ndim == 0 and v.dtype == object else v
    if isinstance(a, (list, tuple)):
        return a[j]
    raise TypeError("vmap: a mapped argument must have a leading plate axis")


def _index_chm(chm: ChoiceMap, j, n):
    """Constraint / previous values of plate element j: values that carry the plate axis first, plus
    whatever sits under the explicit integer address j (`C[j, "x"].set(v)`: constraint.get_submap(idx),
    vmap.py:201, scan.py:262)."""
    from .engine import Sym
    explicit = chm.get_submap(j) if j in chm._children else None
    if explicit is not None:
        rest = ChoiceMap(chm._value, {a: c for a, c in chm._children.items() if not isinstance(a, int)})
        return _index_chm(rest, j, n).merge(explicit) if not rest.static_is_empty() else explicit
    if any(isinstance(a, int) for a in chm._children):
        chm = ChoiceMap(chm._value, {a: c for a, c in chm._children.items() if not isinstance(a, int)})

    def pick(v):
        if isinstance(v, Sym):
            inner = v.value
            if isinstance(inner, np.ndarray) and inner.ndim >= 1 and inner.shape[0] == n:
                return _take(inner, j)
            return inner
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == n:
            return _take(v, j)
        return v
    return chm.map_values(pick)


def _stack(vals):
    from .engine import Sym
    vals = [v.value if isinstance(v, Sym) else v for v in vals]
    if all(v is None for v in vals):
        return None
    if isinstance(vals[0], (tuple, list)):
        return type(vals[0])(_stack([v[k] for v in vals]) for k in range(len(vals[0])))
    arrs = [np.asarray(v, dtype=object) if not (isinstance(v, np.ndarray) and v.dtype == object) else v for v in vals]
    return np.stack(arrs, axis=0)


class Vmap(GenerativeFunction):
    def __init__(self, gen_fn, in_axes=0):
        self.gen_fn, self.in_axes = gen_fn, in_axes

    def _axes(self, args):
        ax = self.in_axes
        if isinstance(ax, int) or ax is None:
            return (ax,) * len(args)
        ax = tuple(ax)
        if len(ax) != len(args):
            raise ValueError("vmap: in_axes does not match the number of arguments")
        return ax

    def _plate_size(self, args, axes):
        sizes = {_axis_len(a) for a, ax in zip(args, axes) if ax is not None}
        sizes.discard(None)
        if len(sizes) != 1:
            raise ValueError(f"vmap: cannot infer the plate size from the mapped arguments ({sizes})")
        return sizes.pop()

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        """Called by static.call_gen_fn while tracing a parent `@gen` function."""
        from .static import _CallRec, _SiteRec, _store_site, call_gen_fn
        if mode not in ("simulate", "generate", "assess"):
            return self._trace_edit(ctx, mode, key, args, constraint, prev, req, req_leaves, addr)
        axes = self._axes(args)
        n = self._plate_size(args, axes)
        if n > VMAP_UNROLL_MAX:
            return self._trace_loop(ctx, mode, key, args, axes, constraint, n, req_leaves, addr)
        from .static import _rec_score
        g = ctx.tr.graph
        keep = ctx.store_sites
        ctx.store_sites = False
        recs, rets = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        for j in range(n):
            kj = Expr(g.add("KDERIVE", (key.node,), imm=j, dtype="key")) if key is not None else None   # split(key, n)[j]
            args_j = tuple(_take(a, j) if ax is not None else a for a, ax in zip(args, axes))
            con_j = _index_chm(constraint, j, n)
            rec, ret, w, s = call_gen_fn(ctx, mode, self.gen_fn, kj, args_j, con_j, None, None, req_leaves, addr)
            recs.append(rec)
            rets.append(ret)
            if keep:
                for r in _leaves(rec):
                    ctx.tr.prestore(r.value)
                    if not isinstance(rec, _SiteRec):      # a bare distribution's plate score is the sum
                        ctx.tr.prestore(r.score)
            if w is not None:
                weight = weight + w                    # w = sum over the plate (vmap.py:214)
            score = score + (s if mode == "assess" else _rec_score(rec))
        ctx.store_sites = keep
        merged = _merge(recs, self.gen_fn)
        retval = _stack(rets)
        if isinstance(merged, _SiteRec):
            # a distribution under vmap is a vector-valued site with split keys; its
            # score is the plate sum
            merged.score = score
            out = merged
        else:
            out = _CallRec(self)
            out.sites = merged.sites
            out.retval = retval
            out.plate_score = score
        if keep:
            for r in _leaves(out):
                _store_site(ctx, r)
        if mode in ("simulate", "assess"):
            return out, retval, None, score
        return out, retval, weight, None

    def _trace_loop(self, ctx, mode, key, args, axes, constraint, n, req_leaves, addr):
        """simulate / generate / assess of a LARGE plate (vmap.py:180-218: `jax.vmap` over any n) as a counted loop IN
        the site program: the inner function is traced ONCE; iteration j runs element j with key split(key, n)[j]
        (= fold_in(key, j): OP_KDERIVER on the iteration number), reads element j of the mapped arguments and of the
        per-element constraints (tables, or per-particle [n, T] leaves read step by step), writes element j of every
        site's [T, n] value / score (seen as [n, T], a plate) and adds its weight / score to loop-carried sums — in
        element order, as the unrolled form does.  One launch runs the whole plate of a particle."""
        from .engine import StepOutput, Sym
        from .static import _CallRec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        ctx.store_sites = False
        what = "a plate of more than 16 elements"
        zero = g.const_f32(0.0)
        wvar = g.loop_var(zero) if mode == "generate" else None
        svar = g.loop_var(zero)
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            k_t = Expr(g.add("KDERIVER", (key.node, t.node), dtype="key")) if key is not None else None
            args_t = tuple(_loop_at(a, t, what) if ax is not None else a for a, ax in zip(args, axes))
            con_t = _loop_step_constraint(constraint, t, n, _loop_at, what) if constraint is not None else None
            rec, ret, w, s_ = call_gen_fn(ctx, mode, self.gen_fn, k_t, args_t, con_t, None, None, req_leaves, addr)
            score_t = s_ if mode == "assess" else _rec_score(rec)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                if isinstance(sc, np.ndarray):
                    raise NotImplementedError(f"{what}: a site with a vector-valued SCORE")
                if keep:
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n), None)
                    r.value = StepOutput(r.origins[0], n)
                    r.score = StepOutput(r.origins[1], n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Sym):
                    v = v.value
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                return StepOutput(tr.store_step(v, n), n)
            rets = stack_out(ret) if (keep or not isinstance(rec, _SiteRec)) else None
            updates = []
            if wvar is not None and w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites = keep

        def drop_retvals(r):
            if isinstance(r, _CallRec):
                r.retval = None
                for x in r.sites.values():
                    drop_retvals(x)
        score = Expr(svar)
        if isinstance(rec, _SiteRec):
            # a distribution under vmap is a vector-valued site with split keys; its score is the plate sum
            out = _SiteRec(rec.gen_fn, rec.value, score)
            retval = rec.value if rets is None else rets
            if not keep:
                retval = rets
        else:
            drop_retvals(rec)
            out = _CallRec(self)
            out.sites = rec.sites
            out.retval = rets
            out.plate_score = score
            retval = rets
        if mode in ("simulate", "assess"):
            return out, retval, None, score
        return out, retval, Expr(wvar), None

    def _trace_edit_loop(self, ctx, kind, key, args, axes, constraint, inner_prev, req, n, req_leaves, addr):
        """Update / IndexRequest of a LARGE plate as a counted loop (the loop form of _trace_edit): iteration j edits
        element j — `Update`: with key split(key, n)[j] and element j of the constraint; `IndexRequest(idx, request)`:
        `request` with the caller's key where idx == j (a Python int or one index per particle: the same test), a plain
        carry-over elsewhere — reading element j of the previous trace and writing element j of the new one."""
        from .engine import StepInput, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        ctx.store_sites = False
        what = "editing a plate of more than 16 elements"
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())

        def prev_at(v, t):
            if isinstance(v, Sym):
                inner = v.value
                if isinstance(inner, (StepInput, RuntimeTable, TableArray)):
                    return Sym(inner[t], None)
                return v
            if isinstance(v, dict):
                return {k: (None if k == "retval" else prev_at(x, t)) for k, x in v.items()}
            if isinstance(v, tuple):
                return tuple(prev_at(x, t) for x in v)
            return v
        zero = g.const_f32(0.0)
        wvar, svar = g.loop_var(zero), g.loop_var(zero)
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            args_t = tuple(_loop_at(a, t, what) if ax is not None else a for a, ax in zip(args, axes))
            prev_t = prev_at(inner_prev, t)
            if kind == "index":
                sub = req.sub
                m_ = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                con_ = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                saved = set(ctx.changed)
                rec, ret, w, _ = call_gen_fn(ctx, m_, self.gen_fn, key, args_t, con_, prev_t, sub, req_leaves, addr)
                ctx.changed = saved
                ctx.memo.clear()
                old, old_ret, w_old, _ = call_gen_fn(ctx, "update", self.gen_fn, None, args_t, ChoiceMap.empty(), prev_t,
                                                     carry_over, req_leaves, addr)
                here = t == req.idx
                zero_e = Expr(zero)
                rec = _select_rec(here, rec, old)
                ret = _select_tree(here, ret, old_ret)
                w = T.where(here, w if w is not None else zero_e, w_old if w_old is not None else zero_e)
            else:
                k_t = Expr(g.add("KDERIVER", (key.node, t.node), dtype="key")) if key is not None else None
                rec, ret, w, _ = call_gen_fn(ctx, "update", self.gen_fn, k_t, args_t,
                                             _loop_step_constraint(constraint, t, n, _loop_at, what), prev_t,
                                             req if kind == "update" else carry_over, req_leaves, addr)
            score_t = _rec_score(rec)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                dis = r.discard.value if isinstance(r.discard, Sym) else r.discard
                if isinstance(sc, np.ndarray):
                    raise NotImplementedError(f"{what}: a site with a vector-valued SCORE")
                if keep:
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n),
                                 tr.store_step(dis, n) if dis is not None else None)
                    r.value = StepOutput(r.origins[0], n)
                    r.score = StepOutput(r.origins[1], n)
                    r.discard = StepOutput(r.origins[2], n) if dis is not None else None

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Sym):
                    v = v.value
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                return StepOutput(tr.store_step(v, n), n)
            rets = stack_out(ret)
            updates = []
            if w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites = keep
        if isinstance(rec, _SiteRec):
            raise NotImplementedError("editing a plate of bare distributions")

        def drop_retvals(r):
            if isinstance(r, _CallRec):
                r.retval = None
                for x in r.sites.values():
                    drop_retvals(x)
        drop_retvals(rec)
        out = _CallRec(self)
        out.sites = rec.sites
        out.retval = rets
        out.plate_score = Expr(svar)
        return out, rets, Expr(wvar), None

    def _trace_edit(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        """Vmap.edit (vmap.py:334-362): `Update(constraint)` edits every element with keys split(key, n)
        and its slice of the constraint (edit_choice_map :236-275); `IndexRequest(idx, request)` applies
        `request` to element idx with the caller's key, everything else is carried over (edit_index
        :277-332).  Any other request raises, as in the reference."""
        from .core.generative import NotSupportedEditRequest
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, _store_site, call_gen_fn
        kind = req.kind if req is not None else "empty"
        if prev is None or "vmap" not in prev:
            raise NotImplementedError("editing a plate of bare distributions (its per-element scores are not kept)")
        if not (mode == "update" or kind in ("update", "index", "empty")):
            raise NotSupportedEditRequest(f"Vmap.edit answers Update and IndexRequest (got {kind!r}), vmap.py:342-362")
        inner_prev = prev["vmap"]
        axes = self._axes(args)
        n = self._plate_size(args, axes)
        if n > VMAP_UNROLL_MAX:
            return self._trace_edit_loop(ctx, kind if kind != "empty" else "update", key, args, axes, constraint,
                                         inner_prev, req, n, req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        ctx.store_sites = False
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())
        recs, rets = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        for j in range(n):
            args_j = tuple(_take(a, j) if ax is not None else a for a, ax in zip(args, axes))
            prev_j = _index_prev(inner_prev, j)
            if kind == "index":
                traced = not isinstance(req.idx, int)
                if traced or j == req.idx:
                    sub = req.sub
                    sub_mode = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                    sub_con = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                    saved = set(ctx.changed)
                    rec, ret, w, _ = call_gen_fn(ctx, sub_mode, self.gen_fn, key, args_j, sub_con, prev_j, sub,
                                                 req_leaves, addr)
                if traced:      # one index per particle: element j is the edited one where idx == j
                    ctx.changed = saved
                    ctx.memo.clear()
                    old, old_ret, _, _ = call_gen_fn(ctx, "update", self.gen_fn, None, args_j, ChoiceMap.empty(),
                                                     prev_j, carry_over, req_leaves, addr)
                    here = req.idx == j
                    rec = _select_rec(here, rec, old)
                    ret = _select_tree(here, ret, old_ret)
                    w = T.where(here, w, 0.0) if w is not None else None
                    ctx.mark_changed([r.value for r in _leaves(rec)])
                elif j != req.idx:      # untouched: an empty Update with unchanged arguments recomputes nothing
                    rec, ret, w, _ = call_gen_fn(ctx, "update", self.gen_fn, None, args_j, ChoiceMap.empty(), prev_j,
                                                 carry_over, req_leaves, addr)
            else:
                kj = Expr(g.add("KDERIVE", (key.node,), imm=j, dtype="key")) if key is not None else None
                rec, ret, w, _ = call_gen_fn(ctx, "update", self.gen_fn, kj, args_j, _index_chm(constraint, j, n), prev_j,
                                             carry_over if req is None or kind != "update" else req, req_leaves, addr)
            recs.append(rec)
            rets.append(ret)
            if keep:
                for r in _leaves(rec):
                    ctx.tr.prestore(r.value)
                    ctx.tr.prestore(r.score)
                    ctx.tr.prestore(r.discard)
            if w is not None:
                weight = weight + w
            score = score + _rec_score(rec)
        ctx.store_sites = keep
        merged = _merge(recs, self.gen_fn)
        out = _CallRec(self)
        out.sites = merged.sites
        out.retval = _stack(rets)
        out.plate_score = score
        if keep:
            for r in _leaves(out):
                _store_site(ctx, r)
        return out, out.retval, weight, None

    # direct use: split(key, n) of the caller's key itself (vmap.py:186)
    def simulate(self, key, args):
        from .static import run_gfi
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .static import run_gfi
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        from .static import run_gfi
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)


def _flat_exprs(tree):
    from .engine import Sym
    if isinstance(tree, Sym):
        tree = tree.value
    if isinstance(tree, (tuple, list)):
        return [e for a in tree for e in _flat_exprs(a)]
    return [] if tree is None else [tree]


def _select_tree(c, new, old):
    from .engine import Sym
    new = new.value if isinstance(new, Sym) else new
    old = old.value if isinstance(old, Sym) else old
    if isinstance(new, (tuple, list)):
        return type(new)(_select_tree(c, a, b) for a, b in zip(new, old))
    if new is None:
        return None
    return T.where(c, new, old)


def _select_rec(c, new, old):
    """where(c, new record, old record), leaf by leaf (same structure)."""
    from .static import _CallRec, _SiteRec
    if isinstance(new, _SiteRec):
        keep = old.value
        disc = _select_tree(c, new.discard if new.discard is not None else keep, keep)
        return _SiteRec(new.gen_fn, _select_tree(c, new.value, old.value), _select_tree(c, new.score, old.score), disc)
    out = _CallRec(new.gen_fn)
    for a in new.sites:
        out.sites[a] = _select_rec(c, new.sites[a], old.sites[a])
    out.retval = _select_tree(c, new.retval, old.retval)
    return out


def _vmap_edit(self, key, trace, edit_request, argdiffs):
    from .static import run_edit
    return run_edit(self, key, trace, edit_request, argdiffs)


Vmap.edit = _vmap_edit


def _index_prev(prev, j):
    """Element j of a symbolic previous plate trace (values / scores carry the plate axis first)."""
    from .engine import Sym

    def pick(v):
        if isinstance(v, Sym):
            inner = v.value
            if isinstance(inner, np.ndarray) and inner.dtype == object and inner.ndim >= 1:
                return Sym(_take(inner, j), None)
            return v
        if isinstance(v, dict):
            return {k: pick(x) for k, x in v.items()}
        if isinstance(v, tuple):
            return tuple(pick(x) for x in v)
        return v
    return pick(prev)


def _loop_step_constraint(chm: ChoiceMap, t, n, at_step, what="scan of more than 16 steps"):
    """The constraint of iteration t of a counted loop: values that carry the step axis first are read at t; what sits
    under an explicit integer address (`C[..., 2, "y"].set(v)`: constraint.get_submap(idx), scan.py:262) becomes a
    MASKED constraint `Mask(v, t == 2)` — the leaf then takes the constrained branch at that step only (OP_SEL), which
    is what the reference's per-step `get_submap` amounts to.  An address constrained at every step wins over an
    explicit one, as in the unrolled form (_index_chm)."""
    from .core.mask import Mask
    from .engine import StepInput, Sym
    from .numpy import RuntimeTable, TableArray
    if chm is None or chm.static_is_empty():
        return ChoiceMap.empty()
    explicit = {a: c for a, c in chm._children.items() if isinstance(a, int)}
    rest = ChoiceMap(chm._value, {a: c for a, c in chm._children.items() if not isinstance(a, int)}) if explicit else chm

    def pick(v):
        inner = v.value if isinstance(v, Sym) else v
        if isinstance(inner, (RuntimeTable, TableArray, StepInput)) and inner.shape[0] == n:
            return inner[t]
        if isinstance(inner, np.ndarray) and inner.ndim >= 1 and inner.shape[0] == n:
            return at_step(inner, t)
        return inner
    out = rest.map_values(pick)
    if not explicit:
        return out
    taken = set(out.addresses())
    per_addr = OrderedDict()
    for i, sub in sorted(explicit.items()):
        if not 0 <= i < n:
            raise IndexError(f"{what}: constraint at step {i} of {n}")
        for a in sub.addresses():
            if a in taken:
                continue
            v = sub[a]
            per_addr.setdefault(a, []).append((i, v.value if isinstance(v, Sym) else v))
    for a, items in per_addr.items():
        val, flag = None, None
        for i, v in items:
            here = t == i
            val = v if val is None else T.where(here, v, val)
            flag = here if flag is None else (flag | here)
        out = out.set(a, Mask(val, flag))
    return out


class Scan(GenerativeFunction):
    """scan.py:140-294: kernel (carry, x) -> (carry, y), repeated `length` times."""

    def __init__(self, kernel_gen_fn, length=None):
        self.kernel_gen_fn, self.length = kernel_gen_fn, length

    def _length(self, scanned_in):
        if self.length is not None:
            return int(self.length)
        lens = set()

        def visit(v):
            if isinstance(v, (tuple, list)) and not (isinstance(v, list) and v and not isinstance(v[0], (tuple, list, np.ndarray))):
                for x in v:
                    visit(x)
            else:
                n = _axis_len(v)
                if n is not None:
                    lens.add(n)
        visit(scanned_in)
        if len(lens) != 1:
            raise ValueError("scan: pass n= or scanned inputs with one common leading length")
        return lens.pop()

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        from .static import _CallRec, _SiteRec, _rec_score, _store_site, call_gen_fn
        if mode not in ("simulate", "generate", "assess"):
            return self._trace_edit(ctx, mode, key, args, constraint, prev, req, req_leaves, addr)
        if len(args) != 2:
            raise TypeError("scan: arguments are (carry, scanned_in)")
        carry, scanned_in = args
        n = self._length(scanned_in)
        if n > SCAN_UNROLL_MAX:
            return self._trace_loop(ctx, mode, key, carry, scanned_in, constraint, n, req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        ctx.store_sites = False
        recs, outs = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        for t in range(n):
            if key is not None:                       # key = fold_in(key, count): the chain of scan.py:213
                key = Expr(g.add("KDERIVE", (key.node,), imm=t, dtype="key"))
            x_t = _tree_take(scanned_in, t)
            con_t = _index_chm(constraint, t, n)
            rec, ret, w, s = call_gen_fn(ctx, mode, self.kernel_gen_fn, key, (carry, x_t), con_t, None, None,
                                         req_leaves, addr)
            if not (isinstance(ret, tuple) and len(ret) == 2):
                raise TypeError("scan: the kernel must return (carry, output)")
            carry, y_t = ret
            recs.append(rec)
            outs.append(y_t)
            if keep:
                for r in _leaves(rec):
                    ctx.tr.prestore(r.value)
                    if not isinstance(rec, _SiteRec):
                        ctx.tr.prestore(r.score)
            if w is not None:
                weight = weight + w
            score = score + (s if mode == "assess" else _rec_score(rec))
        ctx.store_sites = keep
        merged = _merge(recs, self.kernel_gen_fn)
        retval = (carry, _stack(outs))
        if isinstance(merged, _SiteRec):
            merged.score = score
            out = merged
        else:
            out = _CallRec(self)
            out.sites = merged.sites
            out.retval = retval
            out.plate_score = score
        if keep:
            for r in _leaves(out):
                _store_site(ctx, r)
        if mode in ("simulate", "assess"):
            return out, retval, None, score
        return out, retval, weight, None

    def _trace_loop(self, ctx, mode, key, carry, scanned_in, constraint, n, req_leaves, addr):
        """simulate / generate / assess of a LONG scan (scan.py:200-294, 638-664) as a counted loop IN the site
        program (OP_LOOP ... OP_ENDLOOP, gmx_program.h) — what `jax.lax.scan` is to the reference: the kernel is
        traced ONCE; the chained key (key <- fold_in(key, t), scan.py:213), the carry and the running weight / score
        are loop-carried registers; every site's value and score of step t go to element t of a [T, n] leaf (seen
        as [n, T], like a plate); scanned inputs and per-step constraints are read at index t (tables, or
        step-indexed per-particle leaves).  One launch runs all T steps of a particle."""
        from .engine import StepInput, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        ctx.store_sites = False

        def flat_carry(v, out):
            if v is None:
                return ("none",)
            if isinstance(v, Sym):
                v = v.value
            if isinstance(v, (tuple, list)):
                return (type(v).__name__, [flat_carry(x, out) for x in v])
            if isinstance(v, dict):
                return ("dict", {k: flat_carry(x, out) for k, x in v.items()})
            if isinstance(v, np.ndarray) and v.dtype == object:
                return ("array", v.shape, [flat_carry(x, out) for x in v.reshape(-1)])
            out.append(T.lift(v))
            return ("leaf", len(out) - 1)

        def rebuild(tree, leaves):
            k = tree[0]
            if k == "none":
                return None
            if k == "leaf":
                return leaves[tree[1]]
            if k in ("tuple", "list"):
                seq = [rebuild(x, leaves) for x in tree[1]]
                return tuple(seq) if k == "tuple" else seq
            if k == "dict":
                return {a: rebuild(x, leaves) for a, x in tree[1].items()}
            arr = np.empty(len(tree[2]), dtype=object)
            for i, x in enumerate(tree[2]):
                arr[i] = rebuild(x, leaves)
            return arr.reshape(tree[1])

        def at_step(v, t):
            """element t of a scanned input / a per-step constraint"""
            if isinstance(v, Sym):
                v = v.value
            if v is None:
                return None
            if isinstance(v, tuple):
                return tuple(at_step(x, t) for x in v)
            if isinstance(v, dict):
                return {k: at_step(x, t) for k, x in v.items()}
            if isinstance(v, (RuntimeTable, TableArray, StepInput)):
                return v[t]
            if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
                return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
            raise NotImplementedError("scan of more than 16 steps: scanned inputs and per-step constraints must be "
                                      "launch-uniform vectors (tables) or per-particle [n, T] arrays")

        def step_constraint(chm, t):
            return _loop_step_constraint(chm, t, n, at_step)

        leaves0 = []
        ctree = flat_carry(carry, leaves0)
        cvars = [g.loop_var(e.node) for e in leaves0]
        kvar = g.loop_var(key.node) if key is not None else None
        zero = g.const_f32(0.0)
        wvar = g.loop_var(zero) if mode == "generate" else None
        svar = g.loop_var(zero)
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            k_t = Expr(g.add("KDERIVER", (kvar, t.node), dtype="key")) if kvar is not None else None
            x_t = at_step(scanned_in, t)
            con_t = step_constraint(constraint, t) if constraint is not None else None
            carry_in = rebuild(ctree, [Expr(v) for v in cvars])
            rec, ret, w, s = call_gen_fn(ctx, mode, self.kernel_gen_fn, k_t, (carry_in, x_t), con_t, None, None,
                                         req_leaves, addr)
            if not (isinstance(ret, tuple) and len(ret) == 2):
                raise TypeError("scan: the kernel must return (carry, output)")
            carry_out, y_t = ret
            score_t = s if mode == "assess" else _rec_score(rec)
            # this step's trace: element t of every site's [T, n] value / score
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                if isinstance(sc, np.ndarray):
                    raise NotImplementedError("scan of more than 16 steps: a site with a vector-valued SCORE")
                if keep:
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n), None)
                    r.value = StepOutput(r.origins[0], n)
                    r.score = StepOutput(r.origins[1], n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                return StepOutput(tr.store_step(v, n), n)
            ys = stack_out(y_t)
            # loop-carried updates: carry, key chain, running weight and score (added in step order, as unrolled)
            new_leaves = []
            ntree = flat_carry(carry_out, new_leaves)
            if _shape_of(ntree) != _shape_of(ctree):
                raise TypeError("scan: the kernel must return a carry of the same structure as it received")
            # a PARALLEL copy: a new carry may forward another carry variable ((x_new, a) from (a, b), a swap)
            updates = [(var, e.node) for var, e in zip(cvars, new_leaves)]
            if wvar is not None and w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            if kvar is not None:
                updates.append((kvar, k_t.node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites = keep

        def drop_retvals(r):
            if isinstance(r, _CallRec):
                r.retval = None
                for x in r.sites.values():
                    drop_retvals(x)
        retval = (rebuild(ctree, [Expr(v) for v in cvars]), ys)
        score = Expr(svar)
        if isinstance(rec, _SiteRec):
            out = rec
        else:
            drop_retvals(rec)
            out = _CallRec(self)
            out.sites = rec.sites
            out.retval = retval
            out.plate_score = score
        if mode in ("simulate", "assess"):
            return out, retval, None, score
        return out, retval, Expr(wvar), None

    def _trace_edit(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        """Scan.edit (scan.py:596-625): `Update(constraint)` (edit_update :509-594) and
        `Regenerate(selection)` (edit_regenerate :417-507) re-run every step with the chained key
        fold_in(key, t), the step's slice of the previous trace and the carry of the edited
        predecessor; weights and scores are summed over the steps.  `IndexRequest(idx, request)` on a scan
        (edit_index :325-416): step idx is edited with the caller's key and the carries are threaded on."""
        from .core.generative import NotSupportedEditRequest
        from .static import _CallRec, _ReqSpec, _rec_score, _store_site, call_gen_fn
        kind = req.kind if req is not None else "empty"
        if prev is None or "vmap" not in prev:
            raise NotImplementedError("editing a scan of bare distributions")
        if mode == "regen" or kind == "regen":
            sub_mode = "regen"
        elif mode == "update" or kind in ("update", "empty"):
            sub_mode = "update"
        elif kind == "index":
            sub_mode = "index"          # idx: a Python int, or one index per particle (then every step is edited in
                                        # the program and selected where idx == t, as Vmap.edit_index does)
        else:
            raise NotSupportedEditRequest(f"Scan.edit answers Update, Regenerate and IndexRequest (got {kind!r})")
        carry, scanned_in = args
        n = self._length(scanned_in)
        inner_prev = prev["vmap"]
        if n > SCAN_UNROLL_MAX:
            return self._trace_edit_loop(ctx, sub_mode, key, carry, scanned_in, constraint, inner_prev, req, kind, n,
                                         req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        ctx.store_sites = False
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())
        recs, outs = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        key0 = key
        # the carry each step STARTED from is not stored in the trace: the initial carry for step 0, and
        # the previous step's stored carry-out after that — unchanged carries therefore cost nothing
        for t in range(n):
            if key is not None and sub_mode != "index":
                key = Expr(g.add("KDERIVE", (key.node,), imm=t, dtype="key"))
            prev_t = _index_prev(inner_prev, t)
            if sub_mode == "index":
                # edit_index (scan.py:325-416): the sub-request acts on step idx with the caller's key; the
                # carries are threaded on, so the steps after it are re-scored exactly where the edit
                # reaches them (step idx + 1 for a Markov kernel) and nothing else is recomputed
                traced = not isinstance(req.idx, int)
                args_t = (carry, _tree_take(scanned_in, t))
                if traced or t == req.idx:
                    sub = req.sub
                    m_ = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                    con_ = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                    saved = set(ctx.changed)
                    rec, ret, w, _ = call_gen_fn(ctx, m_, self.kernel_gen_fn, key0, args_t, con_, prev_t, sub,
                                                 req_leaves, addr)
                if traced:
                    # where idx != t the step is only carried over (re-scored against a changed carry)
                    ctx.changed = saved
                    ctx.memo.clear()
                    old, old_ret, w_old, _ = call_gen_fn(ctx, "update", self.kernel_gen_fn, key0, args_t,
                                                         ChoiceMap.empty(), prev_t, carry_over, req_leaves, addr)
                    here = req.idx == t
                    zero = Expr(g.const_f32(0.0))
                    rec = _select_rec(here, rec, old)
                    ret = _select_tree(here, ret, old_ret)
                    w = T.where(here, w if w is not None else zero, w_old if w_old is not None else zero)
                    ctx.mark_changed([r.value for r in _leaves(rec)])
                    ctx.mark_changed(_flat_exprs(ret))
                elif t != req.idx:
                    rec, ret, w, _ = call_gen_fn(ctx, "update", self.kernel_gen_fn, key0, args_t, ChoiceMap.empty(),
                                                 prev_t, carry_over, req_leaves, addr)
            elif sub_mode == "regen":
                rec, ret, w, _ = call_gen_fn(ctx, "regen", self.kernel_gen_fn, key, (carry, _tree_take(scanned_in, t)),
                                             ChoiceMap.empty(), prev_t, req, req_leaves, addr)
            else:
                rec, ret, w, _ = call_gen_fn(ctx, "update", self.kernel_gen_fn, key, (carry, _tree_take(scanned_in, t)),
                                             _index_chm(constraint, t, n), prev_t,
                                             req if kind == "update" else carry_over, req_leaves, addr)
            carry, y_t = ret
            recs.append(rec)
            outs.append(y_t)
            if keep:
                for r in _leaves(rec):
                    ctx.tr.prestore(r.value)
                    ctx.tr.prestore(r.score)
                    ctx.tr.prestore(r.discard)
            if w is not None:
                weight = weight + w
            score = score + _rec_score(rec)
        ctx.store_sites = keep
        merged = _merge(recs, self.kernel_gen_fn)
        out = _CallRec(self)
        out.sites = merged.sites
        out.retval = (carry, _stack(outs))
        out.plate_score = score
        if keep:
            for r in _leaves(out):
                _store_site(ctx, r)
        return out, out.retval, weight, None

    def _trace_edit_loop(self, ctx, sub_mode, key, carry, scanned_in, constraint, inner_prev, req, kind, n, req_leaves,
                         addr):
        """Update / Regenerate of a LONG scan as a counted loop (the loop form of _trace_edit, as _trace_loop is of
        trace_call): iteration t edits step t with the chained key, reading element t of the previous trace's
        [n, T] values and scores and writing element t of the new ones (and of the discard).  The carry is
        loop-carried and treated as changed, so every site is re-scored — a site the edit does not reach gets
        new score == old score bit for bit and contributes exactly 0 to the weight, which is what the unrolled form
        obtains by skipping it."""
        from .engine import StepInput, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        ctx.store_sites = False
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())

        def flat_carry(v, out):
            if v is None:
                return ("none",)
            if isinstance(v, Sym):
                v = v.value
            if isinstance(v, (tuple, list)):
                return (type(v).__name__, [flat_carry(x, out) for x in v])
            if isinstance(v, np.ndarray) and v.dtype == object:
                return ("array", v.shape, [flat_carry(x, out) for x in v.reshape(-1)])
            out.append(T.lift(v))
            return ("leaf", len(out) - 1)

        def rebuild(tree, leaves):
            k = tree[0]
            if k == "none":
                return None
            if k == "leaf":
                return leaves[tree[1]]
            if k in ("tuple", "list"):
                seq = [rebuild(x, leaves) for x in tree[1]]
                return tuple(seq) if k == "tuple" else seq
            arr = np.empty(len(tree[2]), dtype=object)
            for i, x in enumerate(tree[2]):
                arr[i] = rebuild(x, leaves)
            return arr.reshape(tree[1])

        def at_step(v, t):
            if isinstance(v, Sym):
                v = v.value
            if v is None:
                return None
            if isinstance(v, tuple):
                return tuple(at_step(x, t) for x in v)
            if isinstance(v, (RuntimeTable, TableArray, StepInput)):
                return v[t]
            if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
                return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
            raise NotImplementedError("editing a scan of more than 16 steps: scanned inputs, constraints and the previous "
                                      "trace must be tables or per-particle [n, T] arrays")

        def prev_at(v, t):
            if isinstance(v, Sym):
                inner = v.value
                if isinstance(inner, (StepInput, RuntimeTable, TableArray)):
                    return Sym(inner[t], None)
                return v
            if isinstance(v, dict):
                return {k: (None if k == "retval" else prev_at(x, t)) for k, x in v.items()}
            if isinstance(v, tuple):
                return tuple(prev_at(x, t) for x in v)
            return v

        def step_constraint(chm, t):
            return _loop_step_constraint(chm, t, n, at_step, "editing a scan of more than 16 steps")

        leaves0 = []
        ctree = flat_carry(carry, leaves0)
        cvars = [g.loop_var(e.node) for e in leaves0]
        key0 = key
        kvar = g.loop_var(key.node) if (key is not None and sub_mode != "index") else None
        zero = g.const_f32(0.0)
        wvar, svar = g.loop_var(zero), g.loop_var(zero)
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            k_t = Expr(g.add("KDERIVER", (kvar, t.node), dtype="key")) if kvar is not None else None
            x_t = at_step(scanned_in, t)
            carry_in = rebuild(ctree, [Expr(v) for v in cvars])
            ctx.mark_changed(_flat_exprs(carry_in))           # a loop-carried value: changed, as far as the trace can tell
            prev_t = prev_at(inner_prev, t)
            if sub_mode == "index":
                # edit_index (scan.py:325-416) in the loop: every iteration traces BOTH the sub-request on step t (with
                # the caller's key, not a chained one) and the plain carry-over, and keeps the edit where idx == t — the
                # form the unrolled code uses for a per-particle idx; a Python-int idx is the same test against a constant
                sub = req.sub
                m_ = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                con_ = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                saved = set(ctx.changed)
                rec, ret, w, _ = call_gen_fn(ctx, m_, self.kernel_gen_fn, Expr(key0.node) if key0 is not None else None,
                                             (carry_in, x_t), con_, prev_t, sub, req_leaves, addr)
                ctx.changed = saved
                ctx.memo.clear()
                old, old_ret, w_old, _ = call_gen_fn(ctx, "update", self.kernel_gen_fn,
                                                     Expr(key0.node) if key0 is not None else None, (carry_in, x_t),
                                                     ChoiceMap.empty(), prev_t, carry_over, req_leaves, addr)
                here = t == req.idx
                zero_e = Expr(g.const_f32(0.0))
                rec = _select_rec(here, rec, old)
                ret = _select_tree(here, ret, old_ret)
                w = T.where(here, w if w is not None else zero_e, w_old if w_old is not None else zero_e)
            elif sub_mode == "regen":
                rec, ret, w, _ = call_gen_fn(ctx, "regen", self.kernel_gen_fn, k_t, (carry_in, x_t), ChoiceMap.empty(),
                                             prev_t, req, req_leaves, addr)
            else:
                rec, ret, w, _ = call_gen_fn(ctx, "update", self.kernel_gen_fn, k_t, (carry_in, x_t),
                                             step_constraint(constraint, t), prev_t,
                                             req if kind == "update" else carry_over, req_leaves, addr)
            if not (isinstance(ret, tuple) and len(ret) == 2):
                raise TypeError("scan: the kernel must return (carry, output)")
            carry_out, y_t = ret
            score_t = _rec_score(rec)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                dis = r.discard.value if isinstance(r.discard, Sym) else r.discard
                if isinstance(sc, np.ndarray):
                    raise NotImplementedError("editing a scan of more than 16 steps: a site with a vector-valued SCORE")
                if keep:
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n),
                                 tr.store_step(dis, n) if dis is not None else None)
                    r.value = StepOutput(r.origins[0], n)
                    r.score = StepOutput(r.origins[1], n)
                    r.discard = StepOutput(r.origins[2], n) if dis is not None else None

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                return StepOutput(tr.store_step(v, n), n)
            ys = stack_out(y_t)
            new_leaves = []
            ntree = flat_carry(carry_out, new_leaves)
            if _shape_of(ntree) != _shape_of(ctree):
                raise TypeError("scan: the kernel must return a carry of the same structure as it received")
            updates = [(var, e.node) for var, e in zip(cvars, new_leaves)]      # a parallel copy, as in _trace_loop
            if w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            if kvar is not None:
                updates.append((kvar, k_t.node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites = keep

        def drop_retvals(r):
            if isinstance(r, _CallRec):
                r.retval = None
                for x in r.sites.values():
                    drop_retvals(x)
        retval = (rebuild(ctree, [Expr(v) for v in cvars]), ys)
        if isinstance(rec, _SiteRec):
            raise NotImplementedError("editing a scan of bare distributions")
        drop_retvals(rec)
        out = _CallRec(self)
        out.sites = rec.sites
        out.retval = retval
        out.plate_score = Expr(svar)
        return out, retval, Expr(wvar), None

    def edit(self, key, trace, edit_request, argdiffs):
        from .static import run_edit
        return run_edit(self, key, trace, edit_request, argdiffs)

    @property
    def gen_fn(self):          # what static._build_trace names the inner function of a plate-like trace
        return self.kernel_gen_fn

    def simulate(self, key, args):
        from .static import run_gfi
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .static import run_gfi
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        from .static import run_gfi
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)


def scan(*, n=None):
    def decorator(gen_fn):
        return Scan(gen_fn, n)
    return decorator


def _tree_take(v, t):
    if v is None:
        return None
    if isinstance(v, tuple):
        return tuple(_tree_take(x, t) for x in v)
    if isinstance(v, dict):
        return {k: _tree_take(x, t) for k, x in v.items()}
    return _take(v, t)


def _leaves(rec):
    from .static import _SiteRec
    if isinstance(rec, _SiteRec):
        return [rec]
    out = []
    for r in rec.sites.values():
        out += _leaves(r)
    return out


def _merge(recs, gen_fn):
    """n per-element records (same structure) -> one record whose values carry the plate axis."""
    from .static import _CallRec, _SiteRec
    first = recs[0]
    if isinstance(first, _SiteRec):
        value = _stack([r.value for r in recs])
        score = _stack([r.score for r in recs])            # per-element scores, plate axis first
        discard = None
        if any(r.discard is not None for r in recs):       # elements not edited "discard" their kept value
            discard = _stack([r.discard if r.discard is not None else r.value for r in recs])
        return _SiteRec(first.gen_fn, value, score, discard)
    out = _CallRec(gen_fn)
    for a in first.sites:
        out.sites[a] = _merge([r.sites[a] for r in recs], first.sites[a].gen_fn)
    out.retval = _stack([r.retval for r in recs])
    return out


def vmap(*, in_axes=0):
    def decorator(gen_fn):
        return Vmap(gen_fn, in_axes)
    return decorator


class _Repeat(Vmap):
    """n independent runs of gen_fn on the same arguments (repeat.py:28-42:
    `gen_fn.contramap(lambda _idx, args: args).vmap(in_axes=(0, None))` over arange(n))."""

    def __init__(self, gen_fn, n):
        super().__init__(gen_fn, None)
        self.n = int(n)

    def _axes(self, args):
        return (None,) * len(args)

    def _plate_size(self, args, axes):
        return self.n


def repeat(*, n: int):
    def decorator(gen_fn):
        return _Repeat(gen_fn, n)
    return decorator
