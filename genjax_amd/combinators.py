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
from .core.mask import Mask
from . import tracer as T
from .tracer import Expr


SCAN_UNROLL_MAX = 16      # longer scans run as a counted loop in the site program (Scan._trace_loop)
VMAP_UNROLL_MAX = 16      # larger plates run as a counted loop too (Vmap._trace_loop): one iteration per element
PATCH_DEPTH_MAX = 32        # engine.Patched chains longer than this are folded (see _vmap_edit_index_o1)
def torch_from_host(v, device):
    import torch
    v = np.asarray(v)
    return torch.from_numpy(np.ascontiguousarray(v.astype(np.float32) if v.dtype.kind == "f" else
                                                 (v.astype(np.int32) if v.dtype.kind in "iu" else v))).to(device)


_FORCE_LOOPS = False      # static.run_gfi / run_edit retrace a program that does not fit the launch slots with every
                          # top-level plate / scan as a counted loop (one slot per leaf whatever the length)


class forced_loops:
    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global _FORCE_LOOPS
        self.keep, _FORCE_LOOPS = _FORCE_LOOPS, self.on
        return self

    def __exit__(self, *exc):
        global _FORCE_LOOPS
        _FORCE_LOOPS = self.keep
        return False


def _forced(ctx, n) -> bool:
    g = ctx.tr.graph
    return _FORCE_LOOPS and n > 1 and not getattr(g, "_in_loop", False) and not g.loop_counts


class _NoDefer(Exception):
    pass


DEFER_MIN_WORK = 1 << 18       # particles x elements from which a large plate of a small particle batch is deferred
DEFER_MAX_BATCH = 4096         # ... and the largest such batch (more particles keep the GPU busy in the loop form)


def _tree_map_plain(v, fn):
    if isinstance(v, (tuple, list)):
        return type(v)(_tree_map_plain(x, fn) for x in v)
    if isinstance(v, dict):
        return {k: _tree_map_plain(x, fn) for k, x in v.items()}
    return fn(v)


VMAP_LAUNCH_MIN = 4096      # a plate this large called directly under ONE key runs with its elements on the launch axis
NEST_UNROLL_MAX = 4       # an unrolled plate whose ELEMENT runs a counted loop keeps at most this many copies of it


def _dyn_take(v, t):
    """element t (a run-time index) of a previous plate value: a step-indexed leaf reads it directly; a short vector
    held in registers goes through a chain of selects"""
    from .engine import StepInput, StepInput2, Sym
    from .numpy import RuntimeTable, TableArray
    if isinstance(v, Sym):
        v = v.value
    if isinstance(v, (StepInput, StepInput2, RuntimeTable, TableArray)):
        return v[t]
    if isinstance(v, np.ndarray) and v.dtype == object and v.ndim >= 1:
        out = v[v.shape[0] - 1]
        for j in range(v.shape[0] - 2, -1, -1):
            out = _select_tree(t == j, v[j], out) if not isinstance(v[j], np.ndarray) else np.vectorize(
                lambda a, b, j=j: T.where(t == j, a, b), otypes=[object])(v[j], out)
        return out
    raise NotImplementedError("editing a plate: its previous values must be a per-particle [n, T] leaf")


def _loop_at(v, t, what="a plate of more than 16 elements"):
    """element t (the iteration number of a counted loop) of a mapped argument / per-element constraint / previous
    value: a launch-uniform table, or a per-particle [n, T] leaf read step by step"""
    from .engine import StepInput, StepInput2, Sym
    from .numpy import RuntimeTable, TableArray
    if isinstance(v, Sym):
        v = v.value
    if v is None:
        return None
    if isinstance(v, tuple):
        return tuple(_loop_at(x, t, what) for x in v)
    if isinstance(v, dict):
        return {k: _loop_at(x, t, what) for k, x in v.items()}
    if isinstance(v, (RuntimeTable, TableArray, StepInput, StepInput2, T.LazyVec)):
        return v[t]
    if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
        return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
    if isinstance(v, np.ndarray) and v.dtype == object and 1 <= v.shape[0] <= VMAP_UNROLL_MAX:
        return _dyn_take(v, t)           # a short vector COMPUTED in the model (registers): a chain of selects
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


def _is_container(v):
    import dataclasses
    return isinstance(v, (tuple, dict)) or (isinstance(v, list) and (not v or isinstance(v[0], (tuple, list, dict, np.ndarray)))) \
        or (dataclasses.is_dataclass(v) and not isinstance(v, type) and not getattr(v, "__gmx_static__", False))


def _tree_leaves_with_axes(v, ax, out, who="vmap"):
    """(leaf, axis) pairs of an argument under an in_axes spec that is a TREE PREFIX of it (jax.vmap's rule): an int /
    None applies to every leaf below, a tuple / list / dict must match the argument's container"""
    import dataclasses
    from .engine import Sym
    if isinstance(v, Sym):
        v = v.value
    if isinstance(ax, (tuple, list)):
        if not isinstance(v, (tuple, list)) or len(v) != len(ax):
            raise ValueError(f"{who} in_axes specification must be a tree prefix of the corresponding value, got "
                             f"specification {ax!r} for value {type(v).__name__}")
        for x, a in zip(v, ax):
            _tree_leaves_with_axes(x, a, out, who)
        return
    if isinstance(ax, dict):
        if not isinstance(v, dict) or set(v) != set(ax):
            raise ValueError(f"{who} in_axes specification must be a tree prefix of the corresponding value")
        for k in v:
            _tree_leaves_with_axes(v[k], ax[k], out, who)
        return
    if _is_container(v):
        if isinstance(v, dict):
            kids = list(v.values())
        elif isinstance(v, (tuple, list)):
            kids = list(v)
        else:
            kids = [getattr(v, f_.name) for f_ in dataclasses.fields(v) if not f_.metadata.get("static")]
        for x in kids:
            _tree_leaves_with_axes(x, ax, out, who)
        return
    out.append((v, ax))


def _tree_take_axes(v, ax, fn):
    """rebuild an argument with `fn(leaf)` applied to its mapped leaves (ax not None)"""
    import dataclasses
    from .engine import Sym
    if isinstance(v, Sym):
        v = v.value
    if ax is None and not isinstance(ax, (tuple, list, dict)):
        return v
    if isinstance(ax, (tuple, list)):
        return type(v)(_tree_take_axes(x, a, fn) for x, a in zip(v, ax))
    if isinstance(ax, dict):
        return {k: _tree_take_axes(v[k], ax[k], fn) for k in v}
    if _is_container(v):
        if isinstance(v, dict):
            return {k: _tree_take_axes(x, ax, fn) for k, x in v.items()}
        if isinstance(v, (tuple, list)):
            return type(v)(_tree_take_axes(x, ax, fn) for x in v)
        return dataclasses.replace(v, **{f_.name: _tree_take_axes(getattr(v, f_.name), ax, fn)
                                         for f_ in dataclasses.fields(v) if not f_.metadata.get("static")})
    return fn(v)


def _tree_mark_unmapped(v, ax):
    """an argument of a launch-axis plate: its UNMAPPED tensor leaves marked launch-uniform (engine.Broadcast), the
    mapped ones (axis 0) left as the per-element leaves they are"""
    import dataclasses
    import torch
    from .engine import Broadcast
    if isinstance(ax, (tuple, list)) and isinstance(v, (tuple, list)):
        return type(v)(_tree_mark_unmapped(x, a) for x, a in zip(v, ax))
    if isinstance(ax, dict) and isinstance(v, dict):
        return {k: _tree_mark_unmapped(v[k], ax[k]) for k in v}
    if _is_container(v):
        if isinstance(v, dict):
            return {k: _tree_mark_unmapped(x, ax) for k, x in v.items()}
        if isinstance(v, (tuple, list)):
            return type(v)(_tree_mark_unmapped(x, ax) for x in v)
        return dataclasses.replace(v, **{f_.name: _tree_mark_unmapped(getattr(v, f_.name), ax)
                                         for f_ in dataclasses.fields(v) if not f_.metadata.get("static")})
    if ax is None and isinstance(v, torch.Tensor):
        return Broadcast(v)
    return v


def _take(a, j):
    from .engine import StepInput2
    if isinstance(a, (StepInput2, T.LazyVec)):
        return a[j]
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
        from .core.mask import Indexed, Mask
        if isinstance(v, Indexed):         # a run-time index: element j is constrained where idx == j
            idx = v.idx.value if isinstance(v.idx, Sym) else v.idx
            val = v.value.value if isinstance(v.value, Sym) else v.value
            return Mask(val, idx == j)
        if isinstance(v, Mask):
            # a masked constraint on the whole plate: element j of its value — and of its flag when the flag carries the
            # plate axis too (a masked plate's choices given back as constraints); the same flag otherwise
            fl = v.flag.value if isinstance(v.flag, Sym) else v.flag
            if isinstance(fl, np.ndarray) and fl.ndim >= 1 and fl.shape[0] == n:
                fl = _take(fl, j)
            return Mask(pick(v.value), fl)
        if isinstance(v, Sym):
            inner = v.value
            from .engine import StepInput2
            if isinstance(inner, (np.ndarray, StepInput2)) and inner.ndim >= 1 and inner.shape[0] == n:
                return _take(inner, j)
            return inner
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == n:
            return _take(v, j)
        from .engine import StepInput2 as _S2
        if isinstance(v, _S2) and v.ndim >= 1 and v.shape[0] == n:      # (already a row of an enclosing loop's leaf)
            return _take(v, j)
        return v
    return chm.map_values(pick)


def _so(tr, origin, T):
    """a loop output as the model sees it; `trailing` = its dims after the batch ((T,) + the site's event)"""
    from .engine import StepOutput
    return StepOutput(origin, T, len(tr.outputs[origin[1]][1]))


def _readable(tr, v, n):
    """What the MODEL gets for the stacked return values of a TOP-LEVEL counted loop (a plate of more than 16 elements, a
    long scan): reads of that very output (engine.StepAlias through Tracing.alias_step_input — one more input slot bound
    to the launch's own output buffer; a thread reads only what it stored), so `jnp.sum(a)`, `a[3]`, `means[z]`, a later
    vector site over them work as on the plain stacked arrays the reference hands back (vmap.py:180-191, scan.py:221-233).
    Returned / recorded as it is, the alias IS that output (no copy).  Inside an enclosing loop, vector-valued elements
    and outputs of nested loops keep the memory-only form (engine.StepOutput)."""
    from .engine import StepOutput
    if tr.graph.loop_counts:
        return v
    if isinstance(v, Mask):
        return Mask(_readable(tr, v.value, n), _readable(tr, v.flag, n))
    if isinstance(v, (tuple, list)):
        return type(v)(_readable(tr, x, n) for x in v)
    if isinstance(v, dict):
        return {k: _readable(tr, x, n) for k, x in v.items()}
    if not isinstance(v, StepOutput) or v.vector_site or v.origin[0] != "out":
        return v
    dt, dims, spec = tr.outputs[v.origin[1]]
    if tuple(dims) != (int(n),) or not (isinstance(spec, tuple) and spec[0] == "step" and not isinstance(spec[1], list)):
        return v
    return tr.alias_step_input(v.origin, dt, int(n))


def _mask_flat(v):
    """a return-value tree with every Mask opened into (value, flag): what _Ctx.mark_changed walks"""
    if isinstance(v, Mask):
        return (_mask_flat(v.value), _mask_flat(v.flag))
    if isinstance(v, (tuple, list)):
        return tuple(_mask_flat(x) for x in v)
    if isinstance(v, dict):
        return {k: _mask_flat(x) for k, x in v.items()}
    return v


def _has_step_rows(tree):
    """does a previous-trace tree hold [n, A, T] step leaves (engine.StepInput2: the values of loops INSIDE a plate) — or,
    inside a counted loop, launch-uniform tables with a long last axis (the same values given as one [.., n, m] table:
    a row picked statically below the looped axis is not a slice of the leaf)?"""
    from .engine import StepInput2, Sym
    from .numpy import RuntimeTable
    if isinstance(tree, Sym):
        tree = tree.value
    if isinstance(tree, StepInput2):
        return True
    if isinstance(tree, RuntimeTable) and tree.ndim >= 2 and tree.shape[-1] > VMAP_UNROLL_MAX and getattr(tree, "_dyn", None) is not None:
        return True
    if isinstance(tree, dict):
        return any(_has_step_rows(v) for v in tree.values())
    if isinstance(tree, (tuple, list)):
        return any(_has_step_rows(v) for v in tree)
    return False


def _store_score_of_stacked_value(tr, r, sc, dis, n):
    """a site inside a counted loop whose VALUE a loop of its own stacked (a long vector-valued site: [T_outer, T, n]
    from static._vector_site_loop) — or that is a per-particle leaf recorded as it is — but whose SCORE is this
    iteration's scalar: the score (and a scalar discard) still go out per iteration, the value keeps its origin"""
    from .engine import StepOutput
    o_sc = tr.store_step(sc, n)
    o_dis = None
    if dis is not None:
        o_dis = dis.origin if isinstance(dis, StepOutput) else tr.store_step(dis, n)
    val = r.value.value if hasattr(r.value, "value") and not isinstance(r.value, StepOutput) else r.value
    r.origins = (val.origin, o_sc, o_dis)
    r.score = _so(tr, o_sc, n)


def _stack(vals):
    from .engine import StepOutput, Sym
    vals = [v.value if isinstance(v, Sym) else v for v in vals]
    if all(v is None for v in vals):
        return None
    # (long rows of a per-particle leaf, recorded / returned as they were given: the stacked outputs they already are)
    vals = [v.passthrough() if (hasattr(v, "passthrough") and v.passthrough() is not None) else v for v in vals]
    if any(isinstance(v, StepOutput) for v in vals):
        if not all(isinstance(v, StepOutput) for v in vals):
            raise NotImplementedError("a plate whose elements mix loop outputs and plain values")
        return StepOutput.stack(vals)
    if isinstance(vals[0], Mask):               # the elements' MaskCombinator return values: one Mask over the plate
        return Mask(_stack([v.value for v in vals]), _stack([v.flag for v in vals]))
    if isinstance(vals[0], (tuple, list)):
        return type(vals[0])(_stack([v[k] for v in vals]) for k in range(len(vals[0])))
    arrs = [np.asarray(v, dtype=object) if not (isinstance(v, np.ndarray) and v.dtype == object) else v for v in vals]
    return T.sym_array(np.stack(arrs, axis=0))       # (values in registers: a traced index into them selects, tracer.SymArray)


def _plate_project(self, key, trace, selection):
    """project of a plate / scan trace (vmap.py:220-234, scan.py:296-323): the inner function's projection per element,
    summed over the plate axis in element order.  Everything selected: the trace's score itself; nothing: 0."""
    import torch
    from .core import choice_map as cm
    from .static import DistributionTrace
    score = trace.get_score()
    if isinstance(selection, cm._All):
        return score
    zero = torch.zeros_like(score) if isinstance(score, torch.Tensor) else 0.0
    if isinstance(selection, cm._None):
        return zero
    if isinstance(trace, DistributionTrace):
        return score if selection.check() else zero
    inner = trace.inner
    w = inner.get_gen_fn().project(key, inner, selection)
    if not isinstance(w, torch.Tensor) or w.ndim <= len(trace.batch_shape):
        return w
    acc = zero
    for j in range(w.shape[-1]):
        acc = acc + w[..., j]
    return acc


class Vmap(GenerativeFunction):
    project = _plate_project

    def __init__(self, gen_fn, in_axes=0):
        self.gen_fn, self.in_axes = gen_fn, in_axes

    def _axes(self, args):
        ax = self.in_axes
        if isinstance(ax, int) or ax is None:
            return (ax,) * len(args)
        ax = tuple(ax)
        if len(ax) != len(args):
            raise ValueError("vmap in_axes specification must be a tree prefix of the corresponding value: "
                             f"{len(ax)} axes for {len(args)} arguments")
        return ax

    def _plate_size(self, args, axes):
        """the common leading length of the mapped leaves (in_axes may be a tree prefix of an argument; a Pytree's
        leaves are mapped together)"""
        pairs = []
        for a, ax in zip(args, axes):
            _tree_leaves_with_axes(a, ax, pairs)
        sizes = []
        for leaf, ax in pairs:
            if ax is None:
                continue
            if ax != 0:
                raise NotImplementedError("vmap: only axis 0 of an argument can be mapped (in_axes entries are 0 or None)")
            shp = getattr(leaf, "shape", None)
            n = _axis_len(leaf) if shp is None else (shp[0] if len(shp) else None)
            if n is None:
                raise ValueError("vmap was requested to map its argument along axis 0, which implies that its rank "
                                 "should be at least 1, but is only 0 (its shape is ())")
            sizes.append(int(n))
        if not sizes:
            raise ValueError("vmap: cannot infer the plate size: no argument is mapped")
        if len(set(sizes)) != 1:
            raise ValueError(f"vmap got inconsistent sizes for array axes to be mapped: {sorted(set(sizes))}")
        return sizes[0]

    def _empty(self, mode):
        """a zero-length plate (vmap.py / GEN-333): no choices, score 0"""
        from .static import _CallRec
        out = _CallRec(self)
        out.retval = None
        out.plate_score = 0.0
        if mode in ("simulate", "assess"):
            return out, None, None, 0.0
        return out, None, 0.0, None

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        """Called by static.call_gen_fn while tracing a parent `@gen` function."""
        from .static import _CallRec, _SiteRec, _store_site, call_gen_fn
        if mode not in ("simulate", "generate", "assess"):
            return self._trace_edit(ctx, mode, key, args, constraint, prev, req, req_leaves, addr)
        axes = self._axes(args)
        n = self._plate_size(args, axes)
        if n == 0:
            return self._empty(mode)
        # a plate OF plates (`model.repeat(n=10).repeat(n=10)`): unrolling both would need one output slot per element
        # of the product; the outer one runs as a loop whose body is the (small, unrolled) inner plate
        d = self._defer(ctx, mode, key, args, axes, constraint, n)
        if d is not None:
            return d, d.retval, None, None
        if getattr(ctx, "sitewise", False) and n >= VMAP_LAUNCH_MIN and not getattr(ctx.tr.graph, "_in_loop", False) \
                and not ctx.tr.graph.loop_counts:
            # ONE trace of the caller and a large plate: in one thread this is n iterations on one lane — the caller
            # runs site by site instead, and this plate with its elements on the launch axis (sitewise.py)
            from .sitewise import NeedsSiteBySite
            raise NeedsSiteBySite()
        nested = isinstance(self.gen_fn, (Vmap, Scan, _ScanAdapter)) and n > 4 and not getattr(ctx.tr.graph, "_in_loop", False)
        if n > VMAP_UNROLL_MAX or nested or _forced(ctx, n):
            return self._trace_loop(ctx, mode, key, args, axes, constraint, n, req_leaves, addr)
        from .static import _rec_score
        g = ctx.tr.graph
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
        recs, rets = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        # (INSIDE a counted loop an unrolled plate whose element opens a loop of its own always becomes a loop itself,
        #  whatever its size: element (t0, j, t2) of a leaf is addressed by the loops' own row-major index, which a
        #  statically picked row in the middle would break)
        probe = (len(g.nodes), g.n_out, len(ctx.tr.outputs)) if ((n > NEST_UNROLL_MAX and not g.loop_counts)
                                                                 or (g.loop_counts and n > 1)) else None
        for j in range(n):
            kj = Expr(g.add("KDERIVE", (key.node,), imm=j, dtype="key")) if key is not None else None   # split(key, n)[j]
            args_j = tuple(_tree_take_axes(a, ax, lambda v: _take(v, j)) for a, ax in zip(args, axes))
            con_j = _index_chm(constraint, j, n)
            depth0 = len(g.loop_counts)
            try:
                rec, ret, w, s = call_gen_fn(ctx, mode, self.gen_fn, kj, args_j, con_j, None, None, req_leaves, addr)
            except NotImplementedError:
                # inside a counted loop, element 0 opened a loop of its own and read a leaf at (t_outer, 0, t_inner): not
                # addressable with a static row in the middle — the reason this plate becomes a loop itself (below)
                if not (j == 0 and probe is not None and depth0 and any(nd.op == "LOOP" for nd in g.nodes[probe[0]:])):
                    raise
                while len(g.loop_counts) > depth0:
                    g.loop_end()
                rec = ret = w = s = None
            if j == 0 and probe is not None and any(nd.op == "LOOP" for nd in g.nodes[probe[0]:]):
                # element 0 opened a counted loop of its own (a long scan / a large plate somewhere inside a `@gen`
                # element): n unrolled copies would need n times its output slots.  What element 0 traced is turned
                # into dead code and the plate runs as a loop AROUND the element's loop (two nested counted loops).
                from .program import EFFECT
                for nd in g.nodes[probe[0]:]:
                    if nd.op in EFFECT:
                        nd.op, nd.args = "DEAD", ()
                g.n_out = probe[1]
                del ctx.tr.outputs[probe[2]:]
                g._cse.clear()
                ctx.store_sites, ctx.sites_deferred = keep, was_deferred
                return self._trace_loop(ctx, mode, key, args, axes, constraint, n, req_leaves, addr)
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
        merged = _merge(recs, self.gen_fn)
        retval = _stack(rets)
        if isinstance(merged, _SiteRec):
            # a distribution under vmap is a vector-valued site with split keys; its
            # score is the plate sum
            if keep and all(isinstance(r, _SiteRec) for r in recs):
                merged.escore = _stack([r.score for r in recs])      # ... and the per-element scores stay in the trace
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
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
        what = "a plate of more than 16 elements"
        zero = g.const_f32(0.0)
        wvar = g.loop_var(zero) if mode == "generate" else None
        svar = g.loop_var(zero)
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            k_t = Expr(g.add("KDERIVER", (key.node, t.node), dtype="key")) if key is not None else None
            args_t = tuple(_tree_take_axes(a, ax, lambda v: _loop_at(v, t, what)) for a, ax in zip(args, axes))
            con_t = _loop_step_constraint(constraint, t, n, _loop_at, what) if constraint is not None else None
            rec, ret, w, s_ = call_gen_fn(ctx, mode, self.gen_fn, k_t, args_t, con_t, None, None, req_leaves, addr)
            score_t = s_ if mode == "assess" else _rec_score(rec)
            for sub_ in (rec.sites.values() if not isinstance(rec, _SiteRec) else ()):
                _store_inner_plate_scores(sub_, tr, n, wanted)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                if wanted and not isinstance(val, StepOutput):   # (a loop INSIDE this one stored its own [T0, T1, n] outputs)
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n), None)
                    r.value = _so(tr, r.origins[0], n)
                    r.score = _so(tr, r.origins[1], n)
                elif wanted and not isinstance(sc, StepOutput):
                    _store_score_of_stacked_value(tr, r, sc, None, n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Sym):
                    v = v.value
                if isinstance(v, Mask):
                    return Mask(stack_out(v.value), stack_out(v.flag))
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                if isinstance(v, StepOutput):          # stacked by a loop inside this one: already [T0, T1, n]
                    return v
                if hasattr(v, "passthrough") and v.passthrough() is not None:
                    return v.passthrough()             # a long row of a per-particle leaf returned as it was given
                return _so(tr, tr.store_step(v, n), n)
            if isinstance(rec, _SiteRec) and wanted and isinstance(rec.value, StepOutput):
                rets = rec.value             # a bare distribution's plate: its stored values ARE what it returns
            else:
                rets = stack_out(ret) if (keep or not isinstance(rec, _SiteRec)) else None
            updates = []
            if wvar is not None and w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred

        def drop_retvals(r):
            if isinstance(r, _CallRec):
                r.retval = None
                for x in r.sites.values():
                    drop_retvals(x)
        score = Expr(svar)
        if isinstance(rec, _SiteRec):
            # a distribution under vmap is a vector-valued site with split keys; its score is the plate sum
            out = _SiteRec(rec.gen_fn, rec.value, score)
            if isinstance(rec.score, StepOutput):
                out.escore = rec.score          # the per-element scores the loop stored: kept beside the plate sum
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
        if isinstance(rec, _SiteRec) and retval is None and constraint is not None:
            # a bare distribution's plate whose values were GIVEN (assess; importance under a constraint on the whole
            # plate): what it returns is what it was given — a table / per-particle leaf, readable where it lies
            cv = constraint.get_value()
            cv = cv.value if isinstance(cv, Sym) else cv
            if cv is not None and not isinstance(cv, Mask) and T._long_vector(cv) == n:
                retval = cv
        retval = _readable(tr, retval, n)
        if mode in ("simulate", "assess"):
            return out, retval, None, score
        return out, retval, Expr(wvar), None

    def _trace_edit_loop(self, ctx, kind, key, args, axes, constraint, inner_prev, req, n, req_leaves, addr,
                         bare_prev=None):
        """Update / IndexRequest of a LARGE plate as a counted loop (the loop form of _trace_edit): iteration j edits
        element j — `Update`: with key split(key, n)[j] and element j of the constraint; `IndexRequest(idx, request)`:
        `request` with the caller's key where idx == j (a Python int or one index per particle: the same test), a plain
        carry-over elsewhere — reading element j of the previous trace and writing element j of the new one."""
        from .engine import StepInput, StepInput2, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
        what = "editing a plate of more than 16 elements"
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())

        def prev_at(v, t):
            if isinstance(v, Sym):
                inner = v.value
                if isinstance(inner, (StepInput, StepInput2, RuntimeTable, TableArray)):
                    return Sym(inner[t], None)
                if isinstance(inner, np.ndarray) and inner.dtype == object and inner.ndim >= 1 and inner.shape[0] == n:
                    return Sym(_dyn_take(inner, t), None)     # (a short plate run as a loop: its leaves sit in registers)
                return v
            if isinstance(v, dict):
                return {k: (None if k == "retval" else prev_at(x, t)) for k, x in v.items()}
            if isinstance(v, tuple):
                return tuple(prev_at(x, t) for x in v)
            return v
        zero = g.const_f32(0.0)
        wvar, svar = g.loop_var(zero), g.loop_var(zero)
        # did this edit change anything the plate RETURNS?  (its return values are readable by the model afterwards —
        # _readable — through an input slot of their own, which the change propagation must then see as changed)
        mark0 = (len(ctx.changed), len(ctx.changed_slots), len(ctx.changed_tables), len(ctx.changed_in_slots))
        args_moved = ctx.args_changed(args)
        touched = kind == "index" or args_moved
        g.loop_begin(n)
        with T.tracing(g):
            t = Expr(g.add("LDT", dtype="i32"))
            args_t = tuple(_tree_take_axes(a, ax, lambda v: _loop_at(v, t, what)) for a, ax in zip(args, axes))
            if bare_prev is not None:
                old_v = _dyn_take(bare_prev["value"], t)
                if bare_prev.get("escore") is not None:
                    old_s = _dyn_take(bare_prev["escore"], t)          # the element's OLD score, as the trace kept it
                elif not args_moved:
                    old_s = self.gen_fn.sym_logpdf(old_v, self.gen_fn.canon(args_t))     # (same arguments: recomputed)
                else:
                    raise NotImplementedError(
                        "update of a plate of a bare distribution (`dist.vmap()`) under CHANGED arguments needs the "
                        "elements' old scores, which this trace no longer holds (it was gathered / stacked / selected): "
                        "write the element as a `@gen` function, or update the trace the model call returned")
                prev_t = {"value": Sym(old_v, None), "score": Sym(old_s, None)}
            else:
                prev_t = prev_at(inner_prev, t)
            if kind == "index" and _has_step_rows(prev_t):
                # the element runs a counted loop itself (a plate of long scans): its outputs are stored inside that loop,
                # so there is no selecting between an edited and a carried-over copy afterwards — the element is traced
                # once, with the request, under the gate idx == j (static._gate_site)
                sub = req.sub
                m_ = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                con_ = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                here = t == req.idx
                outer = ctx.gate
                ctx.gate = here if outer is None else (outer & here)
                try:
                    rec, ret, w, _ = call_gen_fn(ctx, m_, self.gen_fn, key, args_t, con_, prev_t, sub, req_leaves, addr)
                finally:
                    ctx.gate = outer
            elif kind == "index":
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
            for sub_ in (rec.sites.values() if not isinstance(rec, _SiteRec) else ()):
                _store_inner_plate_scores(sub_, tr, n, wanted)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                dis = r.discard.value if isinstance(r.discard, Sym) else r.discard
                if wanted and not isinstance(val, StepOutput):
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n),
                                 (dis.origin if isinstance(dis, StepOutput) else tr.store_step(dis, n)) if dis is not None else None)
                    r.value = _so(tr, r.origins[0], n)
                    r.score = _so(tr, r.origins[1], n)
                    r.discard = (dis if isinstance(dis, StepOutput) else _so(tr, r.origins[2], n)) if dis is not None else None
                elif wanted and not isinstance(sc, StepOutput):
                    _store_score_of_stacked_value(tr, r, sc, dis, n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Sym):
                    v = v.value
                if isinstance(v, Mask):
                    return Mask(stack_out(v.value), stack_out(v.flag))
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                if isinstance(v, StepOutput):          # stacked by a loop inside this one: already [T0, T1, n]
                    return v
                if hasattr(v, "passthrough") and v.passthrough() is not None:
                    return v.passthrough()             # a long row of a per-particle leaf returned as it was given
                return _so(tr, tr.store_step(v, n), n)
            rets = stack_out(ret)
            updates = []
            if w is not None:
                updates.append((wvar, (Expr(wvar) + w).node))
            updates.append((svar, (Expr(svar) + score_t).node))
            g.set_vars(updates)
        g.loop_end()
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
        touched = touched or mark0 != (len(ctx.changed), len(ctx.changed_slots), len(ctx.changed_tables),
                                       len(ctx.changed_in_slots))

        def readable(v):
            r = _readable(tr, v, n)
            if touched:
                ctx.mark_changed(_mask_flat(r))
            return r
        if isinstance(rec, _SiteRec):
            # a bare distribution's plate stays one vector-valued site: values [T], score = the new plate sum
            out = _SiteRec(rec.gen_fn, rec.value, Expr(svar), rec.discard)
            if isinstance(rec.score, StepOutput):
                out.escore = rec.score
            return out, readable(rec.value), Expr(wvar), None

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
        return out, readable(rets), Expr(wvar), None

    def _trace_edit(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        """Vmap.edit (vmap.py:334-362): `Update(constraint)` edits every element with keys split(key, n)
        and its slice of the constraint (edit_choice_map :236-275); `IndexRequest(idx, request)` applies
        `request` to element idx with the caller's key, everything else is carried over (edit_index
        :277-332).  Any other request raises, as in the reference."""
        from .core.generative import NotSupportedEditRequest
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, _store_site, call_gen_fn
        kind = req.kind if req is not None else "empty"
        if prev is None:
            raise NotImplementedError("editing a plate without its previous trace")
        if not (mode == "update" or kind in ("update", "index", "empty")):
            raise NotSupportedEditRequest(f"Vmap.edit answers Update and IndexRequest (got {kind!r}), vmap.py:342-362")
        axes = self._axes(args)
        n = self._plate_size(args, axes)
        if "vmap" not in prev and "sub" not in prev:
            # a plate of a BARE distribution (`normal.vmap()(locs, scales) @ "a"`): its trace keeps the values and the
            # plate-sum score; the per-element scores an edit needs are recomputed from the old values in the loop
            if kind in ("update", "empty") and (constraint is None or constraint.static_is_empty()) \
                    and not ctx.args_changed(args) and ctx.gate is None:
                # nothing to do: no element is constrained and no argument changed, so every element's new score is its
                # old one and the plate contributes exactly 0 (vmap.py:237-275 computes that 0 element by element; the
                # MH move of 3_speed_gains.ipynb c15 edits `x` and leaves three such plates alone).  The previous
                # values are handed back where they lie — readable: `jnp.sum(a)` loops over the old leaf
                pv = prev["value"]
                out = _SiteRec(self.gen_fn, pv, prev["score"])
                out.escore = prev.get("escore")
                return out, (pv.value if hasattr(pv, "value") else pv), None, None
            return self._trace_edit_loop(ctx, kind if kind != "empty" else "update", key, args, axes, constraint, None,
                                         req, n, req_leaves, addr, bare_prev=prev)
        # (as the ELEMENT of an enclosing plate / scan this plate's trace is held flat: its sites with one more axis)
        inner_prev = prev["vmap"] if "vmap" in prev else prev
        # a small plate whose ELEMENTS ran a counted loop (its previous values are [n, A, T] step leaves): the edit runs
        # the plate as a loop around the elements' loops, as a large one does
        if n > VMAP_UNROLL_MAX or (_has_step_rows(inner_prev) and len(ctx.tr.graph.loop_counts) < 3) or _forced(ctx, n):
            return self._trace_edit_loop(ctx, kind if kind != "empty" else "update", key, args, axes, constraint,
                                         inner_prev, req, n, req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())
        recs, rets = [], []
        weight = Expr(g.const_f32(0.0))
        score = Expr(g.const_f32(0.0))
        for j in range(n):
            args_j = tuple(_tree_take_axes(a, ax, lambda v: _take(v, j)) for a, ax in zip(args, axes))
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
        merged = _merge(recs, self.gen_fn)
        out = _CallRec(self)
        out.sites = merged.sites
        out.retval = _stack(rets)
        out.plate_score = score
        if keep:
            for r in _leaves(out):
                _store_site(ctx, r)
        return out, out.retval, weight, None

    # ------------------------------------------------------------------------------------------------------------
    # A LARGE plate under ONE key: its elements on the LAUNCH axis.
    # The reference's Vmap is jax.vmap (vmap.py:180-218): a plate is as parallel as a particle batch.  A plate inside
    # a site program is a counted loop of ONE thread (one particle runs its whole plate) — right when there are many
    # particles, serial when there is one key and a million elements (`generate_datapoint.vmap()` over a dataset).
    # Called directly with an unbatched key and at least VMAP_LAUNCH_MIN elements, the plate therefore runs as the
    # INNER function over a batch of n: element j's key is split(key, n)[j] (a lazy split: derived in registers from
    # the global index), mapped arguments and the constraint's leaves are the per-"particle" leaves, unmapped arguments
    # are launch-uniform, and the plate's score / weight is gmx_sum_rows of the per-element ones — a fixed tree (the
    # counted loop adds in element order: the two forms agree to rounding, and each is reproducible bit for bit).
    # ------------------------------------------------------------------------------------------------------------
    def _launch_axis(self, key, args, constraint=None, batch_shape=None):
        """(n, inner args) when this direct call takes the launch-axis form, else None"""
        import torch
        if key is not None and tuple(key.shape) != ():
            return None
        if key is None and tuple(batch_shape or ()) != ():
            return None
        try:
            axes = self._axes(args)
            n = self._plate_size(args, axes)
        except NotImplementedError:
            return None
        if n < VMAP_LAUNCH_MIN:
            return None
        inner = []
        for a, ax in zip(args, axes):
            pairs = []
            _tree_leaves_with_axes(a, ax, pairs)
            if any(x is not None and not isinstance(leaf, (torch.Tensor, np.ndarray)) for leaf, x in pairs):
                return None                                    # mapped lists: the loop form reads them as tables
            if any(x is not None and isinstance(leaf, np.ndarray) for leaf, x in pairs):
                # a mapped host array (`jnp.zeros(n)`): one device row per element
                from . import _lib
                dev = _lib.get().device
                a = _tree_take_axes(a, ax, lambda v: torch.from_numpy(np.ascontiguousarray(
                    np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))).to(dev)
                    if isinstance(v, np.ndarray) else v)
            inner.append(_tree_mark_unmapped(a, ax))
        if constraint is not None and not constraint.static_is_empty():
            # per-element constraints only: every leaf leads with the plate (no integer sub-addresses, no masks)
            from .core.mask import Mask
            for adr in constraint.addresses():
                v = constraint[adr] if adr else constraint.get_value()
                if isinstance(v, Mask) or any(isinstance(c, int) for c in (adr or ())) \
                        or tuple(getattr(v, "shape", ()))[:1] != (n,):
                    return None
        return n, tuple(inner)

    # -- a large plate of a call over a SMALL batch of particles: lifted out of the program (static.run_gfi `defer`) ------
    def _defer(self, ctx, mode, key, args, axes, constraint, n):
        """The record of this plate as a DEFERRED site, or None.  With B particles and n elements a loop form keeps B
        lanes busy for n iterations; when B is small and n large (ImportanceK with 50 particles over a model of 1e5
        datapoints) the plate is better run AFTER the program, over B x n launch elements: the program computes the
        plate's arguments per particle, the plate then runs the inner function with the keys split(site key_b, n)[j]
        (exactly the loop form's) and its score / weight is the element-order sum per particle.  Only for a plate that
        is a direct site of the top-level model and its LAST one (the weight / score sums keep the reference's order),
        whose values the model returns but does not compute with."""
        from . import static
        from .engine import Expr, Sym
        st = getattr(ctx, "defer", None)
        if st is None or n < VMAP_LAUNCH_MIN or st["B"] * n < DEFER_MIN_WORK or st["plates"] \
                or len(static._HANDLERS) != st["depth"] or getattr(ctx.tr.graph, "_in_loop", False) or ctx.tr.graph.loop_counts:
            return None
        path = []
        if key is not None:                      # the site key as a chain of fold_ins from the particle's key
            node = key.node
            while node.op == "KDERIVE":
                path.append(int(node.imm))
                node = node.args[0]
            if node.op != "LDKEY":
                return None
            path.reverse()
        tr = ctx.tr

        def origin(v):
            if isinstance(v, Sym):
                return v.origin if v.origin is not None else origin(v.value)
            if isinstance(v, (tuple, list)):
                return ("tuple", [origin(x) for x in v])
            if isinstance(v, dict):
                return ("dict", {k_: origin(x) for k_, x in v.items()})
            if isinstance(v, np.ndarray) and v.dtype == object and v.size > 16:
                raise _NoDefer()                 # a long vector COMPUTED per particle: one output slot per element
            if isinstance(v, Expr) or (isinstance(v, np.ndarray) and v.dtype == object):
                return tr.emit_output(v)
            return ("const", v)                  # a host value: a Python number, a table
        con = None
        if constraint is not None and not constraint.static_is_empty():
            # per-element constraints only (every leaf leads with the plate): integer sub-addresses and masks are the loop form's
            from .core.mask import Mask
            for adr in constraint.addresses():
                v = constraint[adr] if adr else constraint.get_value()
                inner_v = v.value if isinstance(v, Sym) else v
                if isinstance(inner_v, Mask) or any(isinstance(c_, int) for c_ in (adr or ())) \
                        or tuple(getattr(inner_v, "shape", ()))[:1] != (n,):
                    return None
            refs = []

            def ref(v):
                if isinstance(v, Sym) and v.origin is not None and v.origin[0] == "leaf":
                    return ("leafref", v.origin[1])
                refs.append(v)
                return v
            con = constraint.map_values(ref)
            if refs:
                return None                      # a computed / masked constraint: the loop form
        try:
            origins = tuple(origin(a) for a in args)
        except _NoDefer:
            return None
        rec = static._DeferredPlateRec(self, mode, path, origins, tuple(axes), con, n, len(st["plates"]))
        st["plates"].append(rec)
        return rec

    def run_deferred(self, rec, keys, args, constraint):
        """the launch of a deferred plate: (plate trace, weight [B] or None) — or (score [B], retval) for assess"""
        import torch
        from . import _lib
        from .engine import Broadcast, sum_rows_inorder
        from .random import split
        from .static import DistributionTrace, StaticTrace, VmapTrace, run_gfi
        dev = _lib.get().device
        n = rec.n
        B = int(keys.shape[0]) if keys is not None else int(rec.B)

        def per_elem(v, mapped):
            if isinstance(v, np.ndarray) and v.dtype != object:
                v = torch_from_host(v, dev)
            if not isinstance(v, torch.Tensor) or isinstance(v, Broadcast):
                return v
            if mapped:
                if v.ndim >= 2 and tuple(v.shape[:2]) == (B, n):
                    return v
                if v.shape[0] != n:
                    raise ValueError(f"vmap: a mapped argument of leading size {v.shape[0]} for a plate of {n}")
                return v.unsqueeze(0).expand((B,) + tuple(v.shape))
            if v.ndim >= 1 and v.shape[0] == B:
                return v.unsqueeze(1).expand((B, n) + tuple(v.shape[1:]))
            return Broadcast(v) if v.ndim >= 1 else v
        inner = tuple(_tree_take_axes(a, ax, lambda v: per_elem(v, True)) if ax is not None
                      else _tree_map_plain(a, lambda v: per_elem(v, False)) for a, ax in zip(args, rec.axes))
        icon = None
        if constraint is not None:
            def con_elem(v):
                if isinstance(v, np.ndarray) and v.dtype != object:
                    v = torch_from_host(v, dev)
                if isinstance(v, torch.Tensor) and v.ndim >= 1 and v.shape[0] == n and tuple(v.shape[:2]) != (B, n):
                    return v.unsqueeze(0).expand((B,) + tuple(v.shape))
                return v
            icon = constraint.map_values(con_elem)
        ikey = split(keys, n) if keys is not None else None
        if rec.mode == "assess":
            s_, r = run_gfi(self.gen_fn, "assess", None, inner, constraint=icon, batch_shape=(B, n))
            return sum_rows_inorder(s_), r
        if rec.mode == "generate":
            tr, w = run_gfi(self.gen_fn, "generate", ikey, inner, constraint=icon)
            w = sum_rows_inorder(w)
        else:
            tr, w = run_gfi(self.gen_fn, "simulate", ikey, inner), None
        if isinstance(tr, DistributionTrace):
            return DistributionTrace(self, tuple(args), tr.value, sum_rows_inorder(tr.score)), w
        st = StaticTrace(tr.gen_fn, None, tr.retval, tr.subtraces)
        return VmapTrace(self, st, sum_rows_inorder(tr.get_score()), tr.retval, tuple(args)), w

    @staticmethod
    def _plate_constraint(constraint, n):
        """per-element constraints given as HOST arrays ([n, ...] numpy / `jnp.array`): one device row per element"""
        if constraint is None or constraint.static_is_empty():
            return constraint
        from . import _lib

        def dev(v):
            if isinstance(v, np.ndarray) and v.dtype != object and v.ndim >= 1 and v.shape[0] == n:
                return torch_from_host(v, _lib.get().device)
            return v
        return constraint.map_values(dev)

    def _plate_trace(self, tr, args):
        """the inner function's trace over the batch of n elements, as this plate's trace"""
        from .engine import sum_rows
        from .static import DistributionTrace, StaticTrace, VmapTrace
        # (the elements' own scores stay on the trace: an IndexRequest re-sums them with one of them replaced)
        if isinstance(tr, DistributionTrace):       # a bare distribution: one vector-valued site, score = the plate sum
            out = DistributionTrace(self, args, tr.value, sum_rows(tr.score))
            out._elem_scores = tr.score
            return out
        inner = StaticTrace(tr.gen_fn, None, tr.retval, tr.subtraces)
        es = tr.get_score()
        out = VmapTrace(self, inner, sum_rows(es), tr.retval, args)
        out._elem_scores = es
        return out

    # direct use: split(key, n) of the caller's key itself (vmap.py:186)
    def simulate(self, key, args):
        from .random import lazy_split
        from .static import run_gfi
        la = self._launch_axis(key, args)
        if la is not None:
            return self._plate_trace(run_gfi(self.gen_fn, "simulate", lazy_split(key, la[0]), la[1]), tuple(args))
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .engine import sum_rows
        from .random import lazy_split
        from .static import run_gfi
        la = self._launch_axis(key, args, constraint)
        if la is not None:
            tr, w = run_gfi(self.gen_fn, "generate", lazy_split(key, la[0]), la[1],
                            constraint=self._plate_constraint(constraint, la[0]))
            return self._plate_trace(tr, tuple(args)), sum_rows(w)
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        from .engine import sum_rows
        from .static import run_gfi
        if batch_shape is None:
            batch_shape = _plate_batch(sample, lambda: self._plate_size(args, self._axes(args)))
        la = self._launch_axis(None, args, sample, batch_shape)
        if la is not None:
            s_, r = run_gfi(self.gen_fn, "assess", None, la[1], constraint=self._plate_constraint(sample, la[0]),
                            batch_shape=(la[0],))
            return sum_rows(s_), r
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)


def _plate_batch(sample, plate_size):
    """assess has no key to tell particles from plate elements: the choices of a plate carry [*batch, plate, *event],
    so the batch is what comes BEFORE the first axis of the plate's length (None: let the engine infer it)"""
    try:
        n = int(plate_size())
    except Exception:
        return None
    for a in sample.addresses():
        v = sample[a] if a else sample.get_value()
        if isinstance(v, Mask):           # the choices of a masked step / element
            v = v.value
        shp = tuple(getattr(v, "shape", ()))
        if n in shp:
            return shp[:shp.index(n)]
    return None


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
    if isinstance(new, Mask):
        return Mask(_select_tree(c, new.value, old.value), _select_tree(c, new.flag, old.flag))
    if isinstance(new, (tuple, list)):
        return type(new)(_select_tree(c, a, b) for a, b in zip(new, old))
    if new is None:
        return None
    from .engine import StepOutput
    if isinstance(new, StepOutput) or isinstance(old, StepOutput):
        # values that live in memory (a long vector site's, recorded by their origin): the same on both sides of the
        # select when the edit did not touch them — which is all a select between stored outputs can mean
        if isinstance(new, StepOutput) and isinstance(old, StepOutput) and new.origin == old.origin:
            return new
        raise NotImplementedError("an edit at one index that replaces the values of a long vector-valued site: "
                                  "constrain the whole plate / scan axis instead (Update with a [n, m] value)")
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
    if getattr(new, "plate_score", None) is not None and getattr(old, "plate_score", None) is not None:
        out.plate_score = _select_tree(c, new.plate_score, old.plate_score)      # (an unrolled plate / scan as the element)
    return out


def _vmap_edit(self, key, trace, edit_request, argdiffs):
    from .static import run_edit
    la = _vmap_edit_launch_axis(self, key, trace, edit_request, argdiffs)
    if la is not None:
        return la
    la = _vmap_edit_index_one_trace(self, key, trace, edit_request, argdiffs)
    if la is not None:
        return la
    o1 = _vmap_edit_index_o1(self, key, trace, edit_request, argdiffs)
    if o1 is not None:
        return o1
    return run_edit(self, key, trace, edit_request, argdiffs)


def _trace_leaf_map(tr, fn, args=None):
    """a trace of the same kind with `fn` applied to every value / score / return-value leaf"""
    from collections import OrderedDict as OD
    from .static import DistributionTrace, StaticTrace, VmapTrace, _tree_map_leaves
    if isinstance(tr, DistributionTrace):
        return DistributionTrace(tr.gen_fn, args if args is not None else tr.args, fn(tr.value), fn(tr.score))
    if isinstance(tr, StaticTrace):
        subs = OD((a, _trace_leaf_map(st, fn)) for a, st in tr.subtraces.items())
        return StaticTrace(tr.gen_fn, args, _tree_map_leaves(tr.retval, lambda v: fn(v) if _is_leaf(v) else v), subs)
    if isinstance(tr, VmapTrace):
        return VmapTrace(tr.gen_fn, _trace_leaf_map(tr.inner, fn), fn(tr.score), _tree_map_leaves(tr.retval, lambda v: fn(v) if _is_leaf(v) else v), args)
    raise TypeError(type(tr).__name__)


def _trace_leaf_zip(old, new, fn, args=None):
    """`fn(old leaf, new leaf)` leaf by leaf over two traces of the same structure"""
    from collections import OrderedDict as OD
    from .static import DistributionTrace, StaticTrace, VmapTrace
    if isinstance(old, DistributionTrace):
        return DistributionTrace(old.gen_fn, args if args is not None else old.args, fn(old.value, new.value), fn(old.score, new.score))
    if isinstance(old, StaticTrace):
        subs = OD((a, _trace_leaf_zip(st, new.subtraces[a], fn)) for a, st in old.subtraces.items())
        return StaticTrace(old.gen_fn, args, _tree_zip(old.retval, new.retval, fn), subs)
    if isinstance(old, VmapTrace):
        return VmapTrace(old.gen_fn, _trace_leaf_zip(old.inner, new.inner, fn), fn(old.score, new.score),
                         _tree_zip(old.retval, new.retval, fn), args)
    raise TypeError(type(old).__name__)


def _is_leaf(v):
    import torch
    from .engine import Gathered, Patched
    return isinstance(v, (torch.Tensor, Gathered, Patched))


def _tree_zip(a, b, fn):
    import dataclasses
    if dataclasses.is_dataclass(a) and not isinstance(a, type) and not getattr(a, "__gmx_static__", False):
        return dataclasses.replace(a, **{f_.name: _tree_zip(getattr(a, f_.name), getattr(b, f_.name), fn) for f_ in dataclasses.fields(a)})
    if isinstance(a, (tuple, list)):
        return type(a)(_tree_zip(x, y, fn) for x, y in zip(a, b))
    if isinstance(a, dict):
        return {k: _tree_zip(a[k], b[k], fn) for k in a}
    return fn(a, b) if _is_leaf(a) else b


def _vmap_edit_index_o1(self, key, trace, request, argdiffs):
    """`IndexRequest(idx, sub)` on a LONG plate held per particle, in O(1) elements (vmap.py:277-332 `edit_index`: a
    dynamic_slice of the trace at idx, the sub-request on that slice with the caller's key, a dynamic_update_slice back):
    the element is sliced out of every leaf (a row of the [n, B] layout for a Python-int idx; one row per particle for
    an index tensor), `sub` edits that ONE element as a trace of the inner function over the particle batch, and the
    new plate trace shares every other element with the old one (engine.Patched leaves; its score is the in-order sum
    of the patched per-element scores, computed when asked: engine.PlateScore).  The counted-loop form this replaces
    re-scored all n elements of every particle under a mask.  Unchanged arguments only (else: the loop form)."""
    import torch
    from .core.generative import Diff, IndexRequest
    from .engine import Deferred, Patched, PlateScore, materialize, sum_rows_inorder
    from .static import StaticGenerativeFunction, StaticTrace, VmapTrace
    if not isinstance(request, IndexRequest) or not isinstance(trace, VmapTrace) or len(trace.batch_shape) != 1:
        return None
    if argdiffs is not None and not Diff.static_check_no_change(argdiffs):
        return None
    scan_elem = isinstance(self.gen_fn, Scan) and isinstance(request.request, IndexRequest)
    if not isinstance(trace.inner, StaticTrace) or not (isinstance(self.gen_fn, (StaticGenerativeFunction, Vmap)) or scan_elem):
        return None                    # (a bare distribution, a switch as the element: the loop form)
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    try:
        axes = self._axes(args)
        n = self._plate_size(args, axes)
    except (NotImplementedError, ValueError):
        return None                    # (e.g. a sub-trace that does not carry its arguments: the loop form says what is wrong)
    if n <= VMAP_UNROLL_MAX:
        return None
    idx = request.idx
    B = int(trace.batch_shape[0])
    if isinstance(idx, torch.Tensor):
        if tuple(idx.shape) != (B,):
            return None
    elif not (0 <= int(idx) < n):
        raise IndexError(f"IndexRequest: index {idx} out of range for a plate of {n} elements")
    for a, ax in zip(args, axes):           # mapped arguments as tensors (per-element rows to pick from)
        pairs = []
        _tree_leaves_with_axes(a, ax, pairs)
        if any(x is not None and not isinstance(leaf, (torch.Tensor, np.ndarray)) for leaf, x in pairs):
            return None

    def arg_at(v):
        if isinstance(idx, int):
            e = v[idx]
            return e.item() if isinstance(e, np.generic) else (np.asarray(e) if isinstance(v, np.ndarray) else e)
        if isinstance(v, np.ndarray):            # a host table: one row per particle, on the device
            v = torch.as_tensor(np.asarray(v), device=idx.device)
        return v[idx.to(torch.int64)]            # [B, ...]
    args_i = tuple(_tree_take_axes(a, ax, arg_at) for a, ax in zip(args, axes))

    def take(v):
        if isinstance(v, Patched):
            return v.take(idx)
        v = materialize(v)
        if not isinstance(v, torch.Tensor) or v.ndim < 2 or v.shape[0] != B or v.shape[1] != n:
            return v
        return v[:, idx] if isinstance(idx, int) else v[torch.arange(B, device=v.device), idx.to(torch.int64)]
    inner_i = _trace_leaf_map(trace.inner, take, args=args_i)
    if isinstance(self.gen_fn, Vmap):      # a plate of plates: the element is itself a plate trace over the particle batch
        st = StaticTrace(self.gen_fn.gen_fn, None, inner_i.retval, inner_i.subtraces)
        inner_i = VmapTrace(self.gen_fn, st, PlateScore(lambda st=st: st.get_score(), batch=(B,)), inner_i.retval, args_i)
    if scan_elem:
        # a plate of scans (`kernel.scan(n=T).vmap()`): element idx is a scan trace over the particle batch ([B, T]
        # leaves), and the sub-request — an IndexRequest itself — edits ONE step of it in O(1) steps
        # (_scan_edit_index_o1); when that form does not apply the whole nest keeps the loop form
        T_in = None
        for leaf in _trace_value_leaves(inner_i):
            T_in = int(leaf.shape[1]) if getattr(leaf, "ndim", 0) >= 2 else None
            break
        if T_in is None or T_in <= getattr(self.gen_fn, "unroll_max", SCAN_UNROLL_MAX):
            return None
        st = StaticTrace(self.gen_fn.kernel_gen_fn, None, None, inner_i.subtraces)
        step_scores = Deferred(lambda st=st: materialize(st.get_score()), (B, T_in))
        inner_i = VmapTrace(self.gen_fn, st, PlateScore(step_scores), inner_i.retval, args_i)
        inner_i._elem_scores = step_scores
        sub = _scan_edit_index_o1(self.gen_fn, key, inner_i, request.request, Diff.no_change(args_i))
        if sub is None:
            return None
        new_i, w, retdiff, bwd = sub
        new_elem_score = new_i.get_score()
        new_i = StaticTrace(self.gen_fn, None, new_i.retval, new_i.inner.subtraces)      # (flat again, as the plate holds it)
    else:
        new_i, w, retdiff, bwd = request.request.edit(key, inner_i, Diff.no_change(args_i))
        new_elem_score = new_i.get_score()

    def patch(old, new):
        # (a chain of patches is read through element by element; past PATCH_DEPTH_MAX it is folded into one whole leaf,
        #  so a sweep over all n elements of the plate copies each leaf n / PATCH_DEPTH_MAX times, not n)
        base = old if isinstance(old, Patched) and old.depth < PATCH_DEPTH_MAX else materialize(old)
        shp = tuple(base.shape)
        rows = torch.as_tensor(materialize(new), device=base.device).to(base.dtype)     # (a constraint may be launch-uniform)
        return Patched(base, idx, rows.expand(shp[:1] + shp[2:]))
    new_inner = _trace_leaf_zip(trace.inner, new_i, patch, args=None)
    # the per-element scores of the plate (the inner trace's score: [B, n]), patched at idx
    elem_old = getattr(trace, "_elem_scores", None)
    if elem_old is None:               # (deferred: the edit itself does not read them, the new trace's score does)
        def elem_scores(inner=trace.inner):
            es = materialize(inner.get_score())
            while es.ndim > 2:         # a plate of plates: the elements' scores are the inner plates' in-order sums
                lead = tuple(es.shape[:-1])
                es = sum_rows_inorder(es.reshape(-1, es.shape[-1])).reshape(lead)
            return es
        elem_old = Deferred(elem_scores, (B, n))
    if isinstance(elem_old, Patched) and elem_old.depth >= PATCH_DEPTH_MAX:
        elem_old = elem_old.materialize()
    elem_new = Patched(elem_old, idx, materialize(new_elem_score))
    out = VmapTrace(self, new_inner, PlateScore(elem_new), new_inner.retval, args)
    out._elem_scores = elem_new
    return out, w, Diff.unknown_change(out.retval) if not Diff.static_check_no_change(retdiff) else Diff.no_change(out.retval), \
        IndexRequest(idx, bwd)


def _scan_edit_index_o1(self, key, trace, request, argdiffs):
    """`IndexRequest(idx, sub)` on a LONG scan held per particle, in O(1) steps — scan.py:325-416 `edit_index`, literally:
    the slice of the trace at step idx is edited by `sub` with the caller's key, the slice at idx + 1 is visited by an
    empty `Update` against the changed carry (its weight is added), and everything else is carried over.  The reference
    ASSERTS that step idx + 1's return value does not change (`Diff.static_check_no_change(retdiff)`, :366): the edit is
    defined only for kernels whose outputs depend on the incoming carry through their choices alone.  That property is
    checked here once per scan, statically — the retdiff of an empty Update under a changed carry (static._trace_edit's
    change propagation) — and it is also what makes step idx's incoming carry computable from step idx - 1's choices
    alone (the kernel's return value on that slice: no chain from step 0).  Kernels without the property, changed
    arguments, one index per particle: the counted-loop form (which re-runs the chain from the top).  The new trace
    shares every other step with the old one (engine.Patched); its score is the in-order sum of the per-step scores,
    computed when asked."""
    import torch
    from .core.generative import Diff, IndexRequest, Update
    from .engine import Deferred, Patched, PlateScore, elementwise, materialize
    from .static import StaticGenerativeFunction, StaticTrace, VmapTrace
    if not isinstance(request, IndexRequest) or not isinstance(trace, VmapTrace) or len(trace.batch_shape) != 1:
        return None
    if argdiffs is not None and not Diff.static_check_no_change(argdiffs):
        return None
    if not isinstance(trace.inner, StaticTrace) or not isinstance(self.kernel_gen_fn, (StaticGenerativeFunction, _KernelAdapter)):
        return None
    def flat_sites(tr):              # (a plate / scan INSIDE the kernel keeps the loop form: its leaves carry further axes)
        from .static import DistributionTrace
        if isinstance(tr, DistributionTrace):
            return True
        return isinstance(tr, StaticTrace) and type(tr) is StaticTrace and all(flat_sites(st) for st in tr.subtraces.values())
    if isinstance(request.request, IndexRequest):
        return None
    nested_inside = not flat_sites(trace.inner)        # a plate / scan INSIDE the kernel: its leaves carry further axes ([B, T, n])
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    if len(args) != 2:
        return None
    carry0, xs = args
    try:
        T_ = self._length(xs)
    except (ValueError, TypeError):
        return None
    idx = request.idx
    if T_ <= getattr(self, "unroll_max", SCAN_UNROLL_MAX) or self.__dict__.get("_o1_refused"):
        return None
    B = int(trace.batch_shape[0])
    per = isinstance(idx, torch.Tensor)            # one step index per particle (a traced idx under the particle vmap)
    if per:
        if tuple(idx.shape) != (B,):
            return None
        idx = idx.to(torch.int64).clamp(0, T_ - 1)          # (dynamic_slice clamps, scan.py:345-350)
    else:
        idx = int(idx)
        if not 0 <= idx < T_:
            raise IndexError(f"IndexRequest: index {idx} out of range for a scan of {T_} steps")
    kernel = self.kernel_gen_fn
    pairs = []
    _tree_leaves_with_axes(xs, 0, pairs, "scan")
    if any(not isinstance(leaf, (torch.Tensor, np.ndarray)) for leaf, _ in pairs if leaf is not None):
        return None

    def x_at(t):
        def pick(v):
            if isinstance(t, torch.Tensor):          # one step per particle: a row per particle, on the device
                v = torch.as_tensor(np.asarray(v), device=t.device) if isinstance(v, np.ndarray) else v
                return v[t]
            e = v[t]
            return e.item() if isinstance(e, np.generic) else (np.asarray(e) if isinstance(v, np.ndarray) else e)
        return _tree_take_axes(xs, 0, pick) if xs is not None else None

    def slice_at(t, cin):
        def take(v):
            if isinstance(v, Patched):
                return v.take(t)
            v = materialize(v)
            if not isinstance(v, torch.Tensor) or v.ndim < 2 or v.shape[0] != B or v.shape[1] != T_:
                return v
            return v[:, t] if not isinstance(t, torch.Tensor) else v[torch.arange(B, device=v.device), t]
        return _trace_leaf_map(trace.inner, take, args=(cin, x_at(t)))

    def dev_leaf(v, like):
        return v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32) if not hasattr(v, "dtype") else np.asarray(v),
                                                                    device=like.device)

    def where_tree(c, a, b):
        """where(c [B], a, b) leaf by leaf over a carry pytree (either side may be launch-uniform)"""
        def sel(x, y):
            x, y = dev_leaf(materialize(x), c), dev_leaf(materialize(y), c)
            cc = c.reshape(c.shape + (1,) * (max(x.ndim, y.ndim) - 1))
            return torch.where(cc, x, y.to(x.dtype) if x.dtype != y.dtype else y)
        def go(x, y):
            if isinstance(x, (tuple, list)):
                return type(x)(go(p, q) for p, q in zip(x, y))
            if isinstance(x, dict):
                return {k_: go(x[k_], y[k_]) for k_ in x}
            return None if x is None else sel(x, y)
        return go(a, b)
    # the property the reference asserts, checked once per scan on step 0: an empty Update under a changed carry leaves
    # the kernel's return value unchanged
    if "_o1_ok" not in self.__dict__:
        _, _, rd0, _ = Update(ChoiceMap.empty()).edit(key, slice_at(0, carry0), (Diff.unknown_change(carry0), Diff.no_change(x_at(0))))
        self.__dict__["_o1_ok"] = bool(Diff.static_check_no_change(rd0))
    if not self.__dict__["_o1_ok"]:
        self.__dict__["_o1_refused"] = True
        return None
    # step idx's incoming carry: the kernel's return value on slice idx - 1 (its choices decide it, whatever came in)
    if per:
        t_prev, t_next, has_next = (idx - 1).clamp(min=0), (idx + 1).clamp(max=T_ - 1), idx + 1 < T_
        prev_sl = slice_at(t_prev, carry0)
        _, ret_prev = kernel.assess(prev_sl.get_choices(), (carry0, x_at(t_prev)), batch_shape=(B,))
        cin = where_tree(idx == 0, carry0, ret_prev[0])
    elif idx == 0:
        cin = carry0
    else:
        prev_sl = slice_at(idx - 1, carry0)
        _, ret_prev = kernel.assess(prev_sl.get_choices(), (carry0, x_at(idx - 1)), batch_shape=(B,))
        cin = ret_prev[0]
    new_i, w, _rd, bwd = request.request.edit(key, slice_at(idx, cin), Diff.no_change((cin, x_at(idx))))
    new_carry = new_i.get_retval()[0]
    nxt = None
    if per or idx + 1 < T_:
        t_n = t_next if per else idx + 1
        nxt, w2, rd2, _ = Update(ChoiceMap.empty()).edit(key, slice_at(t_n, cin), (Diff.unknown_change(new_carry),
                                                                                 Diff.no_change(x_at(t_n))))
        if not Diff.static_check_no_change(rd2):          # (cannot happen after the check above; the reference asserts it here)
            self.__dict__["_o1_refused"] = True
            return None
        if per:                                            # (a particle edited at its LAST step has no successor to visit)
            w2 = torch.where(has_next, dev_leaf(materialize(w2), idx).expand(B), torch.zeros((), device=idx.device))
        w = elementwise(lambda a_, b_: a_ + b_, w, w2)

    def patch_at(t, keep=None):
        def patch(old, new):
            base = old if isinstance(old, Patched) and old.depth < PATCH_DEPTH_MAX else materialize(old)
            shp = tuple(base.shape)
            rows = torch.as_tensor(materialize(new), device=base.device).to(base.dtype).expand(shp[:1] + shp[2:])
            if keep is not None:                           # rows of the particles in `keep` stay what they are NOW
                now = old.take(t) if isinstance(old, Patched) else _take_rows(materialize(old), t)
                rows = torch.where(keep.reshape(keep.shape + (1,) * (rows.ndim - 1)), now, rows)
            return Patched(base, t, rows)
        return patch
    inner_old = trace.inner
    # (the slices' return values are per-step (carry, y) pairs the scan's inner trace does not keep: only choices and scores
    #  are patched; the scan's own return value is rebuilt below)
    strip = _strip_retvals_deep if nested_inside else _strip_retval
    new_inner = _trace_leaf_zip(strip(inner_old), strip(new_i), patch_at(idx), args=None)
    no_next = ~has_next if per else None
    if nxt is not None:
        new_inner = _trace_leaf_zip(new_inner, strip(nxt), patch_at(t_n, no_next), args=None)
    elem_old = getattr(trace, "_elem_scores", None)
    if elem_old is None:
        elem_old = Deferred(lambda inner=inner_old: materialize(inner.get_score()), (B, T_))
    if isinstance(elem_old, Patched) and elem_old.depth >= PATCH_DEPTH_MAX - 1:
        elem_old = elem_old.materialize()
    elem_new = Patched(elem_old, idx, dev_leaf(materialize(new_i.get_score()), materialize(w) if per else torch.zeros(())).expand(B)
                       if per else materialize(new_i.get_score()))
    if nxt is not None:
        s_n = materialize(nxt.get_score())
        if per:
            s_n = torch.where(has_next, dev_leaf(s_n, idx).expand(B), elem_new.take(t_n))
        elem_new = Patched(elem_new, t_n, s_n)
    old_carry, old_ys = trace.get_retval() if isinstance(trace.retval, tuple) and len(trace.retval) == 2 else (None, None)
    new_y = new_i.get_retval()[1]
    ys = _tree_zip(old_ys, new_y, patch_at(idx)) if old_ys is not None else None
    if per:
        carry_out = where_tree(idx == T_ - 1, new_carry, old_carry) if old_carry is not None else None
    else:
        carry_out = new_carry if idx == T_ - 1 else old_carry
        if idx == T_ - 1 and old_carry is not None:
            # (a launch-uniform new value — `Update(C["x"].set(0.3))` at the last step — as one value per particle, which is
            #  what the scan's return value holds)
            carry_out = _tree_zip(old_carry, carry_out, lambda o_, n_: n_ if isinstance(n_, torch.Tensor) and n_.ndim >= 1 else
                                  torch.as_tensor(materialize(n_), device=o_.device).to(o_.dtype).expand(o_.shape).clone())
    out = VmapTrace(self, new_inner, PlateScore(elem_new), (carry_out, ys), args)
    out._elem_scores = elem_new
    return out, w, Diff.unknown_change(out.retval), IndexRequest(idx, bwd)


def _strip_retvals_deep(tr):
    """the same trace without ANY return value, nested ones included: a scan's inner trace keeps choices and scores per
    step, never the steps' (or their callees') return values"""
    from collections import OrderedDict as OD
    from .static import DistributionTrace, StaticTrace, VmapTrace
    if isinstance(tr, DistributionTrace):
        return tr
    if isinstance(tr, VmapTrace):
        out = VmapTrace(tr.gen_fn, _strip_retvals_deep(tr.inner), tr.score, None, tr.args)
        if getattr(tr, "_elem_scores", None) is not None:
            out._elem_scores = tr._elem_scores
        return out
    if isinstance(tr, StaticTrace):
        return StaticTrace(tr.gen_fn, tr.args, None, OD((a, _strip_retvals_deep(st)) for a, st in tr.subtraces.items()))
    return tr


def _take_rows(v, t):
    """v[:, t] for an int t, one row per particle for a [B] index tensor"""
    import torch
    return v[:, t] if not isinstance(t, torch.Tensor) else v[torch.arange(v.shape[0], device=v.device), t]


def _trace_value_leaves(tr):
    from .static import DistributionTrace
    if isinstance(tr, DistributionTrace):
        yield tr.value
        return
    for st in getattr(tr, "subtraces", {}).values():
        yield from _trace_value_leaves(st)
    if hasattr(tr, "inner"):
        yield from _trace_value_leaves(tr.inner)


def _strip_retval(tr):
    from .static import StaticTrace
    return StaticTrace(tr.gen_fn, tr.args, None, tr.subtraces) if isinstance(tr, StaticTrace) else tr


def _vmap_edit_launch_axis(self, key, trace, request, argdiffs):
    """`Update` of a large plate held under ONE key (vmap.py:236-275 `edit_choice_map`): the inner function's Update over
    the batch of n elements, keys split(key, n); weight = the plate sum of the elements' weights."""
    from .core.generative import Diff, Update
    from .engine import sum_rows
    from .random import lazy_split
    from .static import DistributionTrace, StaticTrace, VmapTrace, run_edit
    bare = isinstance(trace, DistributionTrace) and getattr(trace, "_elem_scores", None) is not None
    if not isinstance(request, Update) or not (isinstance(trace, VmapTrace) or bare) or tuple(trace.batch_shape) != ():
        return None
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    la = self._launch_axis(key, args, request.constraint)
    if la is None:
        return None
    n, inner_args = la
    tang = Diff.tree_tangent(argdiffs) if argdiffs is not None else None
    changed = tang is not None and not Diff.static_check_no_change(argdiffs)
    inner_diffs = Diff.unknown_change(inner_args) if changed else Diff.no_change(inner_args)
    if bare:            # a bare distribution under vmap: the distribution's own Update over the batch of n elements
        inner_tr = DistributionTrace(self.gen_fn, inner_args, trace.value, trace._elem_scores)
    else:
        inner_tr = StaticTrace(trace.inner.gen_fn, inner_args, trace.inner.retval, trace.inner.subtraces)
    new_tr, w, retdiff, bwd = run_edit(self.gen_fn, lazy_split(key, n) if key is not None else None, inner_tr,
                                       Update(self._plate_constraint(request.constraint, n)), inner_diffs)
    return self._plate_trace(new_tr, args), sum_rows(w), retdiff, bwd


def _vmap_edit_index_one_trace(self, key, trace, request, argdiffs):
    """`IndexRequest(idx, sub)` on a large plate held under ONE key (vmap.py:277-332 `edit_index`): element idx is sliced
    out of the trace, `sub` edits it with the caller's key (one-element launches), and the new trace's leaves are copies
    with row idx replaced — 12 bytes of memory traffic per element and leaf, no density of any other element evaluated;
    the plate's score is the fixed tree over the elements' scores with one of them replaced (what scoring every element
    again would sum).  Unchanged arguments only."""
    import torch
    from .core.generative import Diff, IndexRequest
    from .engine import materialize, sum_rows
    from .static import DistributionTrace, StaticGenerativeFunction, StaticTrace, VmapTrace
    if not isinstance(request, IndexRequest) or tuple(trace.batch_shape) != ():
        return None
    es = getattr(trace, "_elem_scores", None)
    bare = isinstance(trace, DistributionTrace)
    if es is None or not (bare or (isinstance(trace, VmapTrace) and isinstance(self.gen_fn, StaticGenerativeFunction))):
        return None
    if argdiffs is not None and not Diff.static_check_no_change(argdiffs):
        return None
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    try:
        axes = self._axes(args)
        n = self._plate_size(args, axes)
    except (NotImplementedError, ValueError):
        return None
    idx = request.idx
    if isinstance(idx, torch.Tensor):
        if idx.numel() != 1:
            return None
        idx = int(idx.item())
    if not 0 <= idx < n:
        raise IndexError(f"IndexRequest: index {idx} out of range for a plate of {n} elements")
    es = materialize(es)

    def arg_at(v):
        e = v[idx]
        return e.item() if isinstance(e, np.generic) else e
    args_i = tuple(_tree_take_axes(a, ax, arg_at) for a, ax in zip(args, axes))

    def take(v):
        v = materialize(v)
        return v[idx] if isinstance(v, torch.Tensor) and v.ndim >= 1 and v.shape[0] == n else v

    def put(old, new):
        old = materialize(old)
        if not (isinstance(old, torch.Tensor) and old.ndim >= 1 and old.shape[0] == n):
            return new
        out = old.clone()
        out[idx] = torch.as_tensor(materialize(new), device=old.device).to(old.dtype)
        return out
    if bare:
        elem = DistributionTrace(self.gen_fn, args_i, take(trace.value), es[idx])
    else:
        elem = _trace_leaf_map(trace.inner, take, args=args_i)
    new_e, w, retdiff, bwd = request.request.edit(key, elem, Diff.no_change(args_i))
    es_new = es.clone()
    es_new[idx] = materialize(new_e.get_score()).to(es.dtype)
    if bare:
        out = DistributionTrace(self, args, put(trace.value, new_e.value), sum_rows(es_new))
    else:
        inner = _trace_leaf_zip(trace.inner, new_e, put, args=None)
        out = VmapTrace(self, inner, sum_rows(es_new), inner.retval, args)
    out._elem_scores = es_new
    same = Diff.static_check_no_change(retdiff)
    return out, w, (Diff.no_change(out.get_retval()) if same else Diff.unknown_change(out.get_retval())), IndexRequest(idx, bwd)


Vmap.edit = _vmap_edit


def _index_prev(prev, j):
    """Element j of a symbolic previous plate trace (values / scores carry the plate axis first)."""
    from .engine import Sym

    def pick(v):
        if isinstance(v, Sym):
            inner = v.value
            if isinstance(inner, np.ndarray) and inner.dtype == object and inner.ndim >= 1:
                return Sym(_take(inner, j), None)
            if type(inner).__name__ == "StepInput2" and inner.ndim >= 2:
                return Sym(inner[int(j)], None)      # row j of a per-particle leaf with a long last axis (a short scan /
            return v                                 # plate of long vector sites: the row is read inside the site's loop)
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
    from .engine import StepInput, StepInput2, Sym
    from .numpy import RuntimeTable, TableArray
    if chm is None or chm.static_is_empty():
        return ChoiceMap.empty()
    explicit = {a: c for a, c in chm._children.items() if isinstance(a, int)}
    rest = ChoiceMap(chm._value, {a: c for a, c in chm._children.items() if not isinstance(a, int)}) if explicit else chm

    def pick(v):
        from .core.mask import Indexed
        if isinstance(v, Indexed):         # a run-time index: this iteration's element is constrained where idx == t
            idx = v.idx.value if isinstance(v.idx, Sym) else v.idx
            val = v.value.value if isinstance(v.value, Sym) else v.value
            return Mask(val, idx == t)
        inner = v.value if isinstance(v, Sym) else v
        if isinstance(inner, Mask):
            # a masked constraint over the whole axis: this iteration's element of its value — and of its flag when the
            # flag carries the axis too (the choices of a masked plate / masked scan given back as constraints); the same
            # flag otherwise (an enclosing loop's explicit index, one flag per particle)
            fl = inner.flag.value if isinstance(inner.flag, Sym) else inner.flag
            if (isinstance(fl, (RuntimeTable, TableArray, StepInput, StepInput2)) or isinstance(fl, np.ndarray)) \
                    and getattr(fl, "ndim", 0) >= 1 and fl.shape[0] == n:
                fl = pick(fl)
            return Mask(pick(inner.value), fl)
        if isinstance(inner, (RuntimeTable, TableArray, StepInput, StepInput2)) and inner.shape[0] == n:
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
            if isinstance(v, Mask):            # an already masked constraint at step i: both conditions must hold
                mflag = v.flag.value if isinstance(v.flag, Sym) else v.flag
                here = here & mflag
                v = v.value.value if isinstance(v.value, Sym) else v.value
            val = v if val is None else T.where(here, v, val)
            flag = here if flag is None else (flag | here)
        out = out.set(a, Mask(val, flag))
    return out


def _promote_weak_carry(ctx, kernel_gen_fn, key, leaves0, rebuild_carry, x0, req_leaves, addr):
    """The element types a counted loop's carry settles on.  A loop-carried register has ONE type; `jax.lax.scan`
    promotes a weakly typed initial carry to what the body returns (`add.accumulate()` from the Python int 0 over float
    inputs carries floats: scan.py:1050-1103 and its tests).  When some initial leaf is not a float, the kernel is
    traced once on the initial carry just for its types (the probe's effects become dead code, as in Vmap.trace_call)
    and the initial leaves are converted to the types it returned."""
    from . import tracer as T
    from .program import EFFECT
    from .static import call_gen_fn
    if all(e.dtype in ("f32", "key") for e in leaves0):
        return leaves0
    g, tr = ctx.tr.graph, ctx.tr
    probe = (len(g.nodes), g.n_out, len(tr.outputs))
    with T.tracing(g):
        _, ret, _, _ = call_gen_fn(ctx, "simulate", kernel_gen_fn, key, (rebuild_carry(leaves0), x0), None, None, None,
                                   req_leaves, addr)
    for nd in g.nodes[probe[0]:]:
        if nd.op in EFFECT:
            nd.op, nd.args = "DEAD", ()
    g.n_out = probe[1]
    del tr.outputs[probe[2]:]
    g._cse.clear()
    if not (isinstance(ret, tuple) and len(ret) == 2):
        raise TypeError("scan: the kernel must return (carry, output)")
    outs = []
    _flat_any(ret[0], outs)
    if len(outs) != len(leaves0):
        raise TypeError("scan: the kernel must return a carry of the same structure as it received")
    rank = {"bool": 0, "i32": 1, "f32": 2}
    conv = {"f32": T.as_float, "i32": T.as_int}
    return [conv[o.dtype](e) if rank.get(o.dtype, -1) > rank.get(e.dtype, 3) else e for e, o in zip(leaves0, outs)]


def _flat_any(v, out):
    from . import tracer as T
    from .engine import Sym
    if isinstance(v, Sym):
        v = v.value
    if v is None:
        return
    if isinstance(v, (tuple, list)):
        for x in v:
            _flat_any(x, out)
    elif isinstance(v, dict):
        for x in v.values():
            _flat_any(x, out)
    elif isinstance(v, np.ndarray) and v.dtype == object:
        for x in v.reshape(-1):
            _flat_any(x, out)
    else:
        out.append(T.lift(v))


class Scan(GenerativeFunction):
    """scan.py:140-294: kernel (carry, x) -> (carry, y), repeated `length` times."""

    project = _plate_project

    def __init__(self, kernel_gen_fn, length=None):
        self.kernel_gen_fn, self.length = kernel_gen_fn, length

    def _length(self, scanned_in):
        if self.length is not None:
            return int(self.length)
        pairs = []
        _tree_leaves_with_axes(scanned_in, 0, pairs, "scan")
        lens = []
        for leaf, _ in pairs:
            shp = getattr(leaf, "shape", None)
            n = _axis_len(leaf) if shp is None else (shp[0] if len(shp) else None)
            if n is not None and n not in lens:
                lens.append(int(n))
        if len(lens) > 1:
            raise ValueError("scan got values with different leading axis sizes: " + ", ".join(str(n) for n in lens) + ".")
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
        if n == 0:                 # GEN-333: a zero-length scan has no choices; the carry passes through
            out = _CallRec(self)
            out.retval = (carry, None)
            out.plate_score = 0.0
            if mode in ("simulate", "assess"):
                return out, out.retval, None, 0.0
            return out, out.retval, 0.0, None
        if n > getattr(self, "unroll_max", SCAN_UNROLL_MAX) or _forced(ctx, n):
            return self._trace_loop(ctx, mode, key, carry, scanned_in, constraint, n, req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
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
        from .engine import StepInput, StepInput2, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted

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
            if hasattr(v, "shape") and tuple(v.shape) != () and not isinstance(v, Expr):
                # a CONCRETE array as (part of) the initial carry — `step.scan(n=20)(jnp.zeros(2), None)`, scan.py:200-294
                # takes any pytree: its elements are constants of the program, the carry an array of that shape
                a = np.asarray(v.cpu() if hasattr(v, "cpu") else v)
                a = a.astype(np.float32) if a.dtype.kind == "f" else (a.astype(np.bool_) if a.dtype.kind == "b" else a.astype(np.int32))
                return ("array", a.shape, [flat_carry(x.item(), out) for x in a.reshape(-1)])
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
            if isinstance(v, (RuntimeTable, TableArray, StepInput, StepInput2)):
                return v[t]
            if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
                return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
            if isinstance(v, np.ndarray) and v.dtype == object and 1 <= v.shape[0] <= SCAN_UNROLL_MAX:
                return _dyn_take(v, t)       # a short vector held in registers (a masked scan of a few steps): selects
            raise NotImplementedError("scan of more than 16 steps: scanned inputs and per-step constraints must be "
                                      f"launch-uniform vectors (tables) or per-particle [n, T] arrays (got {type(v).__name__})")

        def step_constraint(chm, t):
            return _loop_step_constraint(chm, t, n, at_step)

        leaves0 = []
        ctree = flat_carry(carry, leaves0)
        with T.tracing(g):
            leaves0 = _promote_weak_carry(ctx, self.kernel_gen_fn, key, leaves0, lambda lv: rebuild(ctree, lv),
                                          at_step(scanned_in, T.lift(0)), req_leaves, addr)
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
            for sub_ in (rec.sites.values() if not isinstance(rec, _SiteRec) else ()):
                _store_inner_plate_scores(sub_, tr, n, wanted)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                if wanted and not isinstance(val, StepOutput):
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n), None)
                    r.value = _so(tr, r.origins[0], n)
                    r.score = _so(tr, r.origins[1], n)
                elif wanted and not isinstance(sc, StepOutput):
                    _store_score_of_stacked_value(tr, r, sc, None, n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Mask):
                    return Mask(stack_out(v.value), stack_out(v.flag))
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                if isinstance(v, StepOutput):          # stacked by a loop inside this one: already [T0, T1, n]
                    return v
                if hasattr(v, "passthrough") and v.passthrough() is not None:
                    return v.passthrough()             # a long row of a per-particle leaf returned as it was given
                return _so(tr, tr.store_step(v, n), n)
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred

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
        retval = (retval[0], _readable(tr, ys, n))      # the stacked outputs as the MODEL sees them: readable (scan.py:221-233)
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
        if prev is None or ("vmap" not in prev and "sub" not in prev):
            raise NotImplementedError("editing a scan of bare distributions")
        # (as the ELEMENT of an enclosing plate / scan — `kernel.scan(n=T).vmap()` — this scan's trace is held flat: the
        #  kernel's sites with one more axis)
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
        inner_prev = prev["vmap"] if "vmap" in prev else prev
        if n > getattr(self, "unroll_max", SCAN_UNROLL_MAX) or _forced(ctx, n):
            return self._trace_edit_loop(ctx, sub_mode, key, carry, scanned_in, constraint, inner_prev, req, kind, n,
                                         req_leaves, addr)
        g = ctx.tr.graph
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
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
            # (the scan's own return value — final carry and stacked outputs — is not a per-step leaf: a carry that is an
            #  ARRAY has its own leading axis, which is not the step axis)
            prev_t = _index_prev({k_: (None if k_ == "retval" else v_) for k_, v_ in inner_prev.items()}
                                 if isinstance(inner_prev, dict) else inner_prev, t)
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
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
        from .engine import StepInput, StepInput2, StepOutput, Sym
        from .numpy import RuntimeTable, TableArray
        from .static import _CallRec, _ReqSpec, _SiteRec, _rec_score, call_gen_fn
        g, tr = ctx.tr.graph, ctx.tr
        keep = ctx.store_sites
        was_deferred = getattr(ctx, "sites_deferred", False)
        wanted = keep or was_deferred      # a parent plate deferred the stores: its elements' LOOP outputs are still wanted
        ctx.store_sites, ctx.sites_deferred = False, wanted
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
            if hasattr(v, "shape") and tuple(v.shape) != () and not isinstance(v, Expr):
                # a CONCRETE array as (part of) the initial carry — `step.scan(n=20)(jnp.zeros(2), None)`, scan.py:200-294
                # takes any pytree: its elements are constants of the program, the carry an array of that shape
                a = np.asarray(v.cpu() if hasattr(v, "cpu") else v)
                a = a.astype(np.float32) if a.dtype.kind == "f" else (a.astype(np.bool_) if a.dtype.kind == "b" else a.astype(np.int32))
                return ("array", a.shape, [flat_carry(x.item(), out) for x in a.reshape(-1)])
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
            if isinstance(v, (RuntimeTable, TableArray, StepInput, StepInput2)):
                return v[t]
            if isinstance(v, (list, np.ndarray)) and not (isinstance(v, np.ndarray) and v.dtype == object):
                return TableArray(np.asarray(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.int32))[t]
            if isinstance(v, np.ndarray) and v.dtype == object and 1 <= v.shape[0] <= SCAN_UNROLL_MAX:
                return _dyn_take(v, t)       # a short vector held in registers (a masked scan of a few steps): selects
            raise NotImplementedError("editing a scan of more than 16 steps: scanned inputs, constraints and the previous "
                                      "trace must be tables or per-particle [n, T] arrays")

        def prev_at(v, t):
            if isinstance(v, Sym):
                inner = v.value
                if isinstance(inner, (StepInput, StepInput2, RuntimeTable, TableArray)):
                    return Sym(inner[t], None)
                if isinstance(inner, np.ndarray) and inner.dtype == object and inner.ndim >= 1 and inner.shape[0] == n:
                    return Sym(_dyn_take(inner, t), None)     # (a short scan run as a loop: its leaves sit in registers)
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
        with T.tracing(g):
            leaves0 = _promote_weak_carry(ctx, self.kernel_gen_fn, key, leaves0, lambda lv: rebuild(ctree, lv),
                                          at_step(scanned_in, T.lift(0)), req_leaves, addr)
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
            if sub_mode == "index" and (_has_step_rows(prev_t) or ctx.gate is not None):
                # the step runs a counted loop itself (a scan of plates / scans), or this scan is an element of a plate
                # edited at ONE index: traced once, with the request, under the gate idx == t (static._gate_site)
                sub = req.sub
                m_ = {"update": "update", "regen": "regen"}.get(sub.kind, "static_edit")
                con_ = sub.constraint if sub.kind == "update" else ChoiceMap.empty()
                here = t == req.idx
                outer = ctx.gate
                ctx.gate = here if outer is None else (outer & here)
                try:
                    rec, ret, w, _ = call_gen_fn(ctx, m_, self.kernel_gen_fn, Expr(key0.node) if key0 is not None else None,
                                                 (carry_in, x_t), con_, prev_t, sub, req_leaves, addr)
                finally:
                    ctx.gate = outer
            elif sub_mode == "index":
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
            for sub_ in (rec.sites.values() if not isinstance(rec, _SiteRec) else ()):
                _store_inner_plate_scores(sub_, tr, n, wanted)
            for r in _leaves(rec):
                val = r.value.value if isinstance(r.value, Sym) else r.value
                sc = r.score.value if isinstance(r.score, Sym) else r.score
                dis = r.discard.value if isinstance(r.discard, Sym) else r.discard
                if wanted and not isinstance(val, StepOutput):
                    r.origins = (tr.store_step(val, n), tr.store_step(sc, n),
                                 (dis.origin if isinstance(dis, StepOutput) else tr.store_step(dis, n)) if dis is not None else None)
                    r.value = _so(tr, r.origins[0], n)
                    r.score = _so(tr, r.origins[1], n)
                    r.discard = (dis if isinstance(dis, StepOutput) else _so(tr, r.origins[2], n)) if dis is not None else None
                elif wanted and not isinstance(sc, StepOutput):
                    _store_score_of_stacked_value(tr, r, sc, dis, n)

            def stack_out(v):
                if v is None:
                    return None
                if isinstance(v, Mask):
                    return Mask(stack_out(v.value), stack_out(v.flag))
                if isinstance(v, (tuple, list)):
                    return type(v)(stack_out(x) for x in v)
                if isinstance(v, StepOutput):          # stacked by a loop inside this one: already [T0, T1, n]
                    return v
                if hasattr(v, "passthrough") and v.passthrough() is not None:
                    return v.passthrough()             # a long row of a per-particle leaf returned as it was given
                return _so(tr, tr.store_step(v, n), n)
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
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred

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
        ys_r = _readable(tr, ys, n)
        if ys_r is not ys:          # (an edit may have changed what the steps put out: their reads are marked so)
            ctx.mark_changed(_mask_flat(ys_r))
        return out, (retval[0], ys_r), Expr(wvar), None

    def edit(self, key, trace, edit_request, argdiffs):
        from .static import run_edit
        o1 = _scan_edit_index_o1(self, key, trace, edit_request, argdiffs)
        if o1 is not None:
            return o1
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
        if batch_shape is None and len(args) == 2:
            batch_shape = _plate_batch(sample, lambda: self._length(args[1]))
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)


def scan(*, n=None):
    def decorator(gen_fn):
        return Scan(gen_fn, n)
    return decorator


def _tree_take(v, t):
    if v is None:
        return None
    return _tree_take_axes(v, 0, lambda x: _take(x, t))


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
    if getattr(first, "plate_score", None) is not None:     # a plate / scan INSIDE the elements: its score per element
        out.plate_score = _stack([r.plate_score for r in recs])
    return out


def _store_inner_plate_scores(rec, tr, n, wanted):
    """Inside a counted loop: a plate / scan called by this iteration's element keeps its own total score (a register
    of THIS iteration); as element t of a [T, n] output it becomes the [n, T] per-element score the sub-trace reports."""
    from .engine import StepOutput, Sym
    from .static import _CallRec
    if not isinstance(rec, _CallRec):
        return
    ps = getattr(rec, "plate_score", None)
    if ps is not None and wanted:
        ps = ps.value if isinstance(ps, Sym) else ps
        if not isinstance(ps, StepOutput):
            rec.plate_score = _so(tr, tr.store_step(ps, n), n)
    for r in rec.sites.values():
        _store_inner_plate_scores(r, tr, n, wanted)


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


def RepeatCombinator(gen_fn, /, *, n: int):
    """repeat.py:28-42: `gen_fn` run n times on the same arguments (keys split(key, n)[j], addresses [j, ...]),
    returning the n results stacked."""
    return _Repeat(gen_fn, n)


def repeat(*, n: int):
    """repeat.py:45-81: the decorator form of RepeatCombinator"""
    def decorator(gen_fn):
        return RepeatCombinator(gen_fn, n=n)
    return decorator


# ---------------------------------------------------------------------------
# scan sugar (scan.py:762-1150): iterate / iterate_final / accumulate / reduce / masked_iterate(_final)
#
# The reference builds these from `dimap` + `scan`; here each is a Scan over a thin kernel adapter plus an
# argument / return-value adapter around the Scan — the traced choices, their addresses ([t, ...]), the chained
# keys and every edit are the underlying Scan's.
# ---------------------------------------------------------------------------
# ---------------------------------------------------------------------------
# MaskCombinator (mask.py:96-262): `gen_fn.mask()` takes a boolean first argument; the call always runs, its score and
# weight count only where the flag holds, its return value is `Mask(value, flag)` and its choices are masked by it.
# Needed by `masked_iterate` / `masked_iterate_final` (scan.py:1050-1150: variable-length chains).
# ---------------------------------------------------------------------------
def _prev_score(prev):
    """the score of a symbolic previous trace (static._trace_tree / _rec_to_prev)"""
    from .engine import Sym
    from .static import MASK_FLAG, _masked_score
    val = lambda v: v.value if isinstance(v, Sym) else v
    if "score" in prev:
        return val(prev["score"])
    subs = prev["sub"]
    if MASK_FLAG in subs:
        return _masked_score(val(subs[MASK_FLAG]["value"]), _prev_score(subs[()]))
    acc = None
    for p_ in subs.values():
        s_ = _prev_score(p_)
        acc = s_ if acc is None else acc + s_
    return acc if acc is not None else 0.0


class MaskCombinator(GenerativeFunction):
    """mask.py:96-262.  The record of a call is a call record with two entries: the inner record under `()` and the
    flag as a pseudo-site under static.MASK_FLAG (a leaf like any other: stacked by plates, stored per step by loops,
    selected by per-particle index edits, gathered by resampling)."""

    def __init__(self, gen_fn):
        self.gen_fn = gen_fn

    @staticmethod
    def _flag(check):
        from .engine import Sym
        check = check.value if isinstance(check, Sym) else check
        if isinstance(check, np.ndarray) and check.dtype != object and check.shape == ():
            check = check.item()
        if isinstance(check, (bool, np.bool_)):
            return bool(check)
        if isinstance(check, Expr):
            return T.as_bool(check)
        # mask.py types the flag `ScalarFlag`: a vector of flags belongs under `.vmap()` (test_mask_combinator.py:227-244)
        raise TypeError(f"mask: the flag must be a scalar boolean (got {type(check).__name__}"
                        f"{' of shape ' + str(tuple(check.shape)) if hasattr(check, 'shape') else ''}); map a vector of "
                        "flags with .vmap()")

    def _record(self, ctx, inner_rec, check, ret):
        from .static import MASK_FLAG, _MASK_FLAG_SITE, _CallRec, _SiteRec, _store_site
        g = ctx.tr.graph
        out = _CallRec(self)
        out.sites[()] = inner_rec
        flag = _SiteRec(_MASK_FLAG_SITE, T.lift(check), Expr(g.const_f32(0.0)))
        out.sites[MASK_FLAG] = _store_site(ctx, flag)
        out.retval = Mask(ret, check)
        return out

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        from .core.generative import NotSupportedEditRequest
        from .engine import Sym
        from .static import MASK_FLAG, _ReqSpec, _masked_score, _rec_score, call_gen_fn
        if not args:
            raise TypeError("mask: the first argument is the flag")
        check, inner_args = self._flag(args[0]), tuple(args[1:])
        if mode in ("simulate", "generate", "assess"):
            rec, ret, w, s_ = call_gen_fn(ctx, mode, self.gen_fn, key, inner_args, constraint, None, None, req_leaves, addr)
            out = self._record(ctx, rec, check, ret)
            if mode == "simulate":
                return out, out.retval, None, _masked_score(check, _rec_score(rec))
            if mode == "assess":
                return out, out.retval, None, _masked_score(check, s_)
            return out, out.retval, (w * T.as_float(check) if w is not None else None), None
        # edit (mask.py:168-223): Update only
        kind = req.kind if req is not None else "empty"
        if not (mode == "update" or kind in ("update", "empty")):
            raise NotSupportedEditRequest(f"MaskCombinator.edit answers Update (got {kind!r}), mask.py:176")
        if prev is None:
            raise NotImplementedError("editing a masked call without its previous trace")
        inner_prev = prev["sub"][()]
        pre = prev["sub"][MASK_FLAG]["value"]
        pre = T.as_bool(T.lift(pre.value if isinstance(pre, Sym) else pre))
        carry_over = _ReqSpec("update", tree=None, constraint=ChoiceMap.empty())
        rec, ret, w, _ = call_gen_fn(ctx, "update", self.gen_fn, key, inner_args, constraint, inner_prev,
                                     req if kind == "update" else carry_over, req_leaves, addr)
        post = T.as_bool(T.lift(check))
        new_score, old_score = _rec_score(rec), _prev_score(inner_prev)
        final_score = T.where(post, new_score, old_score)      # the score of where(post, premasked trace, original trace)
        f = T.as_float
        t_to_t, t_to_f = pre & post, pre & ~post
        f_to_f, f_to_t = ~pre & ~post, ~pre & post
        weight = ((f(f_to_t) * final_score + f(t_to_f) * (-T.lift(old_score))) + f(f_to_f) * 0.0) \
            + f(t_to_t) * (w if w is not None else 0.0)
        out = self._record(ctx, rec, check, ret)
        ctx.mark_changed(out.retval.flag if isinstance(out.retval.flag, Expr) else [])
        return out, out.retval, weight, None

    def simulate(self, key, args):
        from .static import run_gfi
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .static import run_gfi
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        from .static import run_gfi
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)

    def edit(self, key, trace, edit_request, argdiffs):
        from .static import run_edit
        return run_edit(self, key, trace, edit_request, argdiffs)

    def __repr__(self):
        return f"genjax.mask({self.gen_fn!r})"


def mask(f):
    """mask.py:265-322: the decorator form of MaskCombinator"""
    return MaskCombinator(f)


class _KernelAdapter(GenerativeFunction):
    """kernel(carry, x) built from a user function: `call(carry, x)` gives the user function's arguments,
    `ret(user retval)` the (carry, output) pair."""

    def __init__(self, gen_fn, call, ret):
        self.gen_fn, self._call, self._ret = gen_fn, call, ret

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        from .static import call_gen_fn
        carry, x = args
        rec, retval, w, s_ = call_gen_fn(ctx, mode, self.gen_fn, key, self._call(carry, x), constraint, prev, req,
                                         req_leaves, addr)
        return rec, self._ret(retval), w, s_


class _ScanAdapter(GenerativeFunction):
    """`pre(*args)` -> the Scan's (carry, xs); `post(args, (carry, ys))` -> the return value.  Traces, choices and
    edits are the Scan's own (this object is the trace's generative function, so an edit re-applies `pre`)."""

    project = _plate_project

    def __init__(self, scan_gf, pre, post, name, post_ctx=False):
        self.scan_gf, self._pre, self._post, self.name = scan_gf, pre, post, name
        self._post_ctx = post_ctx          # `post` also takes the tracing context (it records outputs of its own)

    @property
    def gen_fn(self):
        return self.scan_gf.kernel_gen_fn

    def trace_call(self, ctx, mode, key, args, constraint, prev, req, req_leaves, addr):
        rec, retval, w, s_ = self.scan_gf.trace_call(ctx, mode, key, self._pre(*args), constraint, prev, req, req_leaves,
                                                     addr)
        out = self._post(args, retval, ctx) if self._post_ctx else self._post(args, retval)
        if hasattr(rec, "sites"):
            rec.gen_fn = self
            rec.retval = out
        return rec, out, w, s_

    def simulate(self, key, args):
        from .static import run_gfi
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .static import run_gfi
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        from .static import run_gfi
        if batch_shape is None:
            # the Scan's own inference (Scan.assess): without it the step axis of the choices reads as a particle batch
            batch_shape = _plate_batch(sample, lambda: self.scan_gf._length(self._pre(*args)[1]))
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)

    def edit(self, key, trace, edit_request, argdiffs):
        from .static import run_edit
        return run_edit(self, key, trace, edit_request, argdiffs)

    def __repr__(self):
        return f"genjax.{self.name}({self.scan_gf.kernel_gen_fn.gen_fn!r})"


def _prepend(init, ys, ctx=None):
    """[init, ys[0], ys[1], ...] leaf by leaf (scan.py:762-788 `prepend_initial_acc`).  The stacked outputs of a
    counted loop live in memory only ([*batch, T, *event] after the launch): the initial value is then recorded as an
    output of its own and put in front when the launch's results are resolved (engine.resolve "prepend")."""
    from .engine import StepOutput, Sym
    init = init.value if isinstance(init, Sym) else init
    if isinstance(init, (tuple, list)):
        return type(init)(_prepend(a, b, ctx) for a, b in zip(init, ys))
    if isinstance(init, dict):
        return {k: _prepend(init[k], ys[k], ctx) for k in init}
    if isinstance(ys, StepOutput):
        return StepOutput(("prepend", ctx.tr.emit_output(init), ys.origin, ys.trailing), ys.T + 1, ys.trailing)
    head = np.asarray(init, dtype=object) if not (isinstance(init, np.ndarray) and init.dtype == object) else init
    ys = np.asarray(ys, dtype=object) if not (isinstance(ys, np.ndarray) and ys.dtype == object) else ys
    return np.concatenate([head[None], ys.reshape((ys.shape[0],) + head.shape)], axis=0)


def _sugar_scan(kernel, n, every):
    return Scan(kernel, n)


def iterate(*, n: int):
    """f: a -> a  =>  a -> [a, f(a), f(f(a)), ...] (n + 1 values; scan.py:916-977)"""
    def decorator(f):
        k = _KernelAdapter(f, lambda carry, _x: (carry,), lambda r: (r, r))
        return _ScanAdapter(_sugar_scan(k, n, True), lambda init: (init, None), lambda args, ret, ctx: _prepend(args[0], ret[1], ctx),
                            "iterate", post_ctx=True)
    return decorator


def iterate_final(*, n: int):
    """f: a -> a  =>  a -> f^n(a) (scan.py:980-1047)"""
    def decorator(f):
        k = _KernelAdapter(f, lambda carry, _x: (carry,), lambda r: (r, None))
        return _ScanAdapter(_sugar_scan(k, n, False), lambda init: (init, None), lambda args, ret: ret[0], "iterate_final")
    return decorator


def _masked_scan(kernel):
    """a masked step keeps its flags (and their plates') beside its choices: unrolled, that is dozens of launch slots
    per step — it always runs as the counted loop (one slot per leaf whatever the length), its edits reading the
    previous trace's per-particle vectors step by step (step_leaf_min)"""
    sc = Scan(kernel, None)
    sc.unroll_max = 0
    return sc


def masked_iterate_final():
    """f: a -> a  =>  (a, [flag]) -> the carry after the last step (scan.py:1050-1097): every step runs and threads its
    return value on (`masked_retval.value`, flag or not: scan.py:1089); only the steps whose flag holds count in the
    score and weights, and only their choices are visible"""
    def decorator(f):
        k = _KernelAdapter(MaskCombinator(f), lambda carry, flag: (flag, carry), lambda m: (m.value, None))
        out = _ScanAdapter(_masked_scan(k), lambda init, flags: (init, flags), lambda args, ret: ret[0],
                           "masked_iterate_final")
        out.step_leaf_min = 0          # its edits read the previous [n, T] leaves step by step whatever T
        return out
    return decorator


def masked_iterate():
    """f: a -> a  =>  (a, [flag]) -> [a, f(a), f(f(a)), ...] (scan.py:1100-1150)"""
    def decorator(f):
        k = _KernelAdapter(MaskCombinator(f), lambda carry, flag: (flag, carry), lambda m: (m.value, m.value))
        out = _ScanAdapter(_masked_scan(k), lambda init, flags: (init, flags),
                           lambda args, ret, ctx: _prepend(args[0], ret[1], ctx), "masked_iterate", post_ctx=True)
        out.step_leaf_min = 0
        return out
    return decorator


def accumulate():
    """f: (c, a) -> c  =>  (c, [a]) -> [c0, c1, ..., cN] (scan.py:791-851)"""
    def decorator(f):
        k = _KernelAdapter(f, lambda carry, x: (carry, x), lambda r: (r, r))
        return _ScanAdapter(_sugar_scan(k, None, True), lambda init, xs: (init, xs),
                            lambda args, ret, ctx: _prepend(args[0], ret[1], ctx), "accumulate", post_ctx=True)
    return decorator


def reduce():
    """f: (c, a) -> c  =>  (c, [a]) -> cN (scan.py:854-913)"""
    def decorator(f):
        k = _KernelAdapter(f, lambda carry, x: (carry, x), lambda r: (r, None))
        return _ScanAdapter(_sugar_scan(k, None, False), lambda init, xs: (init, xs), lambda args, ret: ret[0], "reduce")
    return decorator
