"""Site-by-site execution of a `@gen` function called for ONE trace (no particle batch) that contains a LARGE plate.

The traced form (static.run_gfi) compiles the whole model into one site program and runs one thread per particle:
right for 1e6 particles of a small model, wrong for the other shape the reference serves just as well — ONE trace of a
model whose plates hold 1e4 .. 1e8 elements (`7_application_dirichlet_mixture_model.ipynb` c6 / c10: hyper-parameters,
then `generate_datapoint.vmap(...)` over the data; `4_index_request.ipynb` c3: three 1e4 .. 1e8-element plates and one
observation of their sums).  In one thread such a plate is a loop of n iterations on one lane.

Here the model's source runs ON THE HOST with concrete device values, the way the reference's handlers run it under
`jit` tracing (static.py:254-673) — site by site, in program order, each site through the callee's own GFI method
with the key `fold_in(key, counter)` (static.py:261): a distribution is a one-element launch, a nested `@gen` function
is traced (or, if it holds a large plate itself, executed this way in turn), and a plate of >= VMAP_LAUNCH_MIN elements
takes its launch-axis form (combinators.Vmap._launch_axis: one GPU thread per ELEMENT).  Weights and scores are added
in program order on the device (engine.elementwise), so every number is the one the traced form computes.

Edits (static.py:827-981) run the same way: a site whose arguments did not change and that the request does not
address keeps its sub-trace untouched — which is what makes `StaticRequest({"a": IndexRequest(i, Update(v))})` on a
1e8-element plate cost one element's densities (combinators._vmap_edit_index_one_trace: a copy of the plate's leaves
with one row replaced) plus the sites downstream of it.

static.run_gfi / run_edit come here when Vmap.trace_call raises NeedsSiteBySite (an un-batched call reached a large
plate) and remember that per call signature."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .core.choice_map import ChoiceMap
from .core.generative import Diff, EditRequest, IndexRequest, Regenerate, Update


class NeedsSiteBySite(Exception):
    """raised while TRACING an un-batched call that reaches a plate of >= VMAP_LAUNCH_MIN elements"""


def _add(a, b):
    """a + b of two scores / weights in f32 on the device (0.0 + w is w: the first term costs nothing)"""
    from .engine import elementwise
    if isinstance(a, float) and a == 0.0:
        return b
    if isinstance(b, float) and b == 0.0:
        return a
    return elementwise(lambda x, y: x + y, a, b)


def _resolve(gen_fn, args):
    from .core.generative import GenerativeFunctionClosure
    while isinstance(gen_fn, GenerativeFunctionClosure):
        gen_fn, args = GenerativeFunctionClosure(gen_fn.gen_fn, gen_fn.args + tuple(args), gen_fn.kwargs)._target()
    return gen_fn, tuple(args)


class _Handler:
    """what `callee(*args) @ addr` reaches while a model runs site by site (static.trace -> _HANDLERS[-1].handle)"""

    def __init__(self, mode, key, constraint=None, prev=None, request=None):
        self.mode, self.key = mode, key
        self.constraint = constraint if constraint is not None else ChoiceMap.empty()
        self.prev, self.request = prev, request
        self.counter = 1
        self.subtraces: "OrderedDict" = OrderedDict()
        self.weight, self.score = 0.0, 0.0
        self.backward = OrderedDict()          # addr -> the sub-edit's backward request

    def _key(self):
        from .random import fold_in
        c = self.counter
        self.counter += 1
        return fold_in(self.key, c) if self.key is not None else None

    def handle(self, addr, gen_fn, args):
        from .static import AddressReuse, MissingAddress
        gen_fn, args = _resolve(gen_fn, args)
        if addr in self.subtraces:
            raise AddressReuse(addr)
        k = self._key()
        sub = self.constraint.get_submap(addr)
        big = vector_site_size(gen_fn, args)
        try:
            if wide_dirichlet(gen_fn, args) and self.mode in ("simulate", "generate", "assess"):
                out = dirichlet_site(gen_fn, self.mode, k, args, sub)
                if self.mode == "assess":
                    self.score = _add(self.score, out[0])
                    self.subtraces[addr] = None
                    return out[1]
                tr = out[0]
                if self.mode == "generate":
                    self.weight = _add(self.weight, out[1])
            elif big is not None and self.mode in ("simulate", "generate", "assess"):
                out = vector_site(gen_fn, self.mode, k, args, sub, big)
                if self.mode == "assess":
                    self.score = _add(self.score, out[0])
                    self.subtraces[addr] = None
                    return out[1]
                tr = out[0]
                if self.mode == "generate":
                    self.weight = _add(self.weight, out[1])
            elif self.mode == "simulate":
                tr = gen_fn.simulate(k, args)
            elif self.mode == "generate":
                tr, w = gen_fn.generate(k, sub, args)
                self.weight = _add(self.weight, w)
            elif self.mode == "assess":
                try:            # ONE trace: the callee must not read a plate's length as a particle batch
                    s, retval = gen_fn.assess(sub, args, batch_shape=())
                except TypeError:
                    s, retval = gen_fn.assess(sub, args)
                self.score = _add(self.score, s)
                self.subtraces[addr] = None
                return retval
            else:
                tr = self._edit(addr, gen_fn, k, args, sub)
        except MissingAddress as e:
            inner = tuple(a for a in e.args if a != ())
            raise MissingAddress(*((addr,) + inner)) from None
        self.subtraces[addr] = tr
        return tr.get_retval()

    # -- edits ---------------------------------------------------------------------------------------------------------
    def _subrequest(self, addr, sub_constraint):
        from .static import StaticRequest, _norm
        r = self.request
        if isinstance(r, Update):
            return Update(sub_constraint) if not sub_constraint.static_is_empty() else None
        if isinstance(r, Regenerate):
            sel = r.selection(addr)
            from .core import choice_map as cm
            return None if isinstance(sel, cm._None) else Regenerate(sel)
        if isinstance(r, StaticRequest):
            for a, q in r.addressed.items():
                if a != () and _norm(a) == _norm(addr):
                    return q
            return None
        raise NotImplementedError(f"site-by-site edit with a {type(r).__name__}")

    def _edit(self, addr, gen_fn, k, args, sub_constraint):
        try:
            prev = self.prev.subtraces[addr]
        except KeyError:
            raise KeyError(f"address {addr!r} is not in the previous trace") from None
        req = self._subrequest(addr, sub_constraint)
        changed = not same_args(args, prev.get_args())
        if req is None and not changed:
            return prev                        # untouched: no launch, the sub-trace is shared
        if req is None:
            req = Update(ChoiceMap.empty())    # re-scored against its new arguments
        ad = Diff.unknown_change(args) if changed else Diff.no_change(args)
        big = vector_site_size(gen_fn, args)
        if wide_dirichlet(gen_fn, args):
            new, w, bwd = dirichlet_site_update(gen_fn, prev, req, args)
        elif big is not None and getattr(prev, "_elem_scores", None) is not None:
            new, w, bwd = vector_site_update(gen_fn, k, prev, req, args, changed, big)
        else:
            new, w, _retdiff, bwd = req.edit(k, prev, ad)
        self.weight = _add(self.weight, w)
        self.backward[addr] = bwd
        return new


VECTOR_SITE_MIN = 65          # elements from which a vector-valued site of ONE trace runs on the launch axis (an unrolled
                              # site stores one output slot per element: 64 of them)


def sum_defined(x):
    """the sum of a site's per-element scores / weights in the build's defined order: element order below VMAP_LAUNCH_MIN
    elements, the plate score's fixed tree from there on (oracle: sum_vector)"""
    from .combinators import VMAP_LAUNCH_MIN
    from .engine import sum_rows, sum_rows_inorder
    if x.shape[-1] >= VMAP_LAUNCH_MIN:
        return sum_rows(x)
    return sum_rows_inorder(x.reshape(1, -1)).reshape(())


def vector_site_size(gen_fn, args):
    """n when `gen_fn(*args)` is a DISTRIBUTION site whose value is a vector of n >= VECTOR_SITE_MIN elements — a vector
    parameter (`normal(clusters[idx], 1.0)` with idx of n elements) or `sample_shape=n` — else None"""
    VMAP_LAUNCH_MIN = VECTOR_SITE_MIN
    from .distributions import Distribution, _Categorical
    if not isinstance(gen_fn, Distribution) or gen_fn.sample_op is None and not isinstance(gen_fn, _Categorical):
        return None
    pos, kw = (args[0], args[1]) if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple) else (args, {})
    ss = kw.get("sample_shape")
    if ss is not None:
        ss = ss.unwrap() if hasattr(ss, "unwrap") else ss
        ss = (int(ss),) if isinstance(ss, (int, np.integer)) else tuple(int(x) for x in ss)
        return ss[0] if len(ss) == 1 and ss[0] >= VMAP_LAUNCH_MIN else None
    if isinstance(gen_fn, _Categorical):
        return None                     # (its vector parameter is the logits of ONE draw)
    sizes = {int(v.shape[0]) for v in list(pos) + list(kw.values())
             if isinstance(v, (torch.Tensor, np.ndarray)) and getattr(v, "dtype", None) != object and v.ndim == 1}
    if any(isinstance(v, (torch.Tensor, np.ndarray)) and v.ndim > 1 for v in list(pos) + list(kw.values())):
        return None
    if len(sizes) == 1 and max(sizes) >= VMAP_LAUNCH_MIN:
        return max(sizes)
    return None


def symbolic_vector_site_size(args):
    """the same question asked while TRACING (arguments are symbolic: vectors are object arrays or tables)"""
    VMAP_LAUNCH_MIN = VECTOR_SITE_MIN
    pos, kw = (args[0], args[1]) if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple) else (args, {})
    ss = kw.get("sample_shape")
    if ss is not None:
        ss = ss.unwrap() if hasattr(ss, "unwrap") else ss
        ss = (int(ss),) if isinstance(ss, (int, np.integer)) else tuple(int(x) for x in ss)
        return ss[0] if len(ss) == 1 and ss[0] >= VMAP_LAUNCH_MIN else None
    big = [int(v.shape[0]) for v in list(pos) + [x for k_, x in kw.items() if k_ != "sample_shape"]
           if hasattr(v, "shape") and len(getattr(v, "shape", ())) == 1 and int(v.shape[0]) >= VMAP_LAUNCH_MIN]
    return max(big) if big else None


def vector_site(dist, mode, key, args, constraint, n):
    """A distribution site whose value has n >= VECTOR_SITE_MIN elements, for ONE trace: the scalar distribution over a
    launch of n elements — element i draws with counter i from the ONE site key (what `tfd.X(...).sample(seed=key)` of
    that shape does; unrolled vector-valued sites give element i the immediate i) — and the site's
    score is the defined sum of the elements' log-densities (sum_defined; oracle: sum_vector)."""
    from . import _lib
    from .combinators import torch_from_host
    from .engine import Broadcast
    from .static import DistributionTrace, run_gfi
    dev = _lib.get().device
    pos, kw = (args[0], dict(args[1])) if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple) else (args, {})
    kw.pop("sample_shape", None)

    def elem(v):            # a vector parameter is one value per element, anything else is shared by all of them
        if isinstance(v, np.ndarray) and v.dtype != object and v.ndim == 1 and v.shape[0] == n:
            return torch_from_host(v, dev)
        if isinstance(v, torch.Tensor) and v.ndim == 1 and v.shape[0] == n:
            return v
        if isinstance(v, torch.Tensor) and v.ndim >= 1:
            return Broadcast(v)         # e.g. the logits of every draw of a categorical
        return v
    eargs = tuple(elem(v) for v in pos)
    if kw:
        eargs = (eargs, {k_: elem(v) for k_, v in kw.items()})
    value = constraint.get_value() if constraint is not None and not constraint.static_is_empty() else None
    if value is not None:
        if isinstance(value, np.ndarray):
            value = torch_from_host(value, dev)
        if tuple(getattr(value, "shape", ())) != (n,):
            raise ValueError(f"{dist.name}: the constraint of a {n}-element site must have {n} elements")
        con = ChoiceMap.choice(value)
    if mode == "assess":
        if value is None:
            from .static import MissingAddress
            raise MissingAddress(())
        s, v = run_gfi(dist, "assess", None, eargs, constraint=con, batch_shape=(n,))
        return sum_defined(s), v
    if mode == "generate" and value is not None:
        tr, w = run_gfi(dist, "generate", key, eargs, constraint=con, batch_shape=(n,), elem_index=True)
        out = DistributionTrace(dist, tuple(args), tr.value, sum_defined(tr.score))
        out._elem_scores = tr.score
        return out, sum_defined(w)
    tr = run_gfi(dist, "simulate", key, eargs, batch_shape=(n,), elem_index=True)
    out = DistributionTrace(dist, tuple(args), tr.value, sum_defined(tr.score))
    out._elem_scores = tr.score
    if mode == "generate":
        return out, _as_score(0.0)
    return (out,)


def vector_site_update(dist, key, prev, req, args, changed, n):
    """`Update` (a new value and / or new arguments) of a distribution site of n >= VMAP_LAUNCH_MIN elements held by ONE
    trace: the scalar distribution's Update over the launch of n elements (distribution.py:189-242), weight and score
    summed in the fixed tree"""
    from . import _lib
    from .combinators import torch_from_host
    from .engine import Broadcast
    from .static import DistributionTrace, run_edit
    if isinstance(req, Regenerate):
        # edit_regenerate (distribution.py:258-300): the selected site is drawn again (weight = new score - old score,
        # backward = Update(old value)); an unselected one is re-scored against its arguments
        from .engine import elementwise, materialize
        if req.selection.check():
            new, = vector_site(dist, "simulate", key, args, None, n)
            w = elementwise(lambda a_, b_: a_ - b_, materialize(new.get_score()), materialize(prev.get_score()))
            return new, w, Update(ChoiceMap.choice(prev.value))
        req = Update(ChoiceMap.empty())
    from .static import Rejuvenate as _Rejuvenate
    if isinstance(req, _Rejuvenate):
        # Rejuvenate.edit (rejuvenate.py:70-94), literally, on the launch axis: key, sub_key = split(key); the proposal
        # simulated from sub_key at the arguments the mapping builds from the OLD value (a vector-valued site itself: its n
        # draws and their densities are one launch), Update(proposed) on this site, the proposal assessed at the old value;
        # weight = (w + bwd) - fwd
        from .distributions import Distribution as _Dist
        from .engine import elementwise, materialize
        from .random import split as _split
        if not isinstance(req.proposal, _Dist):
            raise NotImplementedError(f"Rejuvenate on a {n}-element site of one trace: the proposal must be a distribution")
        old = ChoiceMap.choice(prev.value)
        margs = req.argument_mapping(old)
        margs = margs if isinstance(margs, tuple) else (margs,)
        sub = _split(key)[1]
        prop, = vector_site(req.proposal, "simulate", sub, margs, None, n)
        fwd = materialize(prop.get_score())
        new, w, _ = vector_site_update(dist, key, prev, Update(ChoiceMap.choice(prop.value)), args, changed, n)
        bwd, _v = vector_site(req.proposal, "assess", None, margs, old, n)
        final = elementwise(lambda w_, b_, f_: (w_ + b_) - f_, materialize(w), materialize(bwd), fwd)
        return new, final, req
    if not isinstance(req, Update):
        raise NotImplementedError(f"{type(req).__name__} on a {n}-element site of one trace (Update / Regenerate / Rejuvenate; "
                                  "write the plate with vmap for per-element requests)")
    dev = _lib.get().device
    pos, kw = (args[0], dict(args[1])) if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple) else (args, {})
    kw.pop("sample_shape", None)

    def elem(v):
        if isinstance(v, np.ndarray) and v.dtype != object and v.ndim == 1 and v.shape[0] == n:
            return torch_from_host(v, dev)
        if isinstance(v, torch.Tensor) and v.ndim == 1 and v.shape[0] == n:
            return v
        if isinstance(v, torch.Tensor) and v.ndim >= 1:
            return Broadcast(v)
        return v
    eargs = tuple(elem(v) for v in pos)
    if kw:
        eargs = (eargs, {k_: elem(v) for k_, v in kw.items()})
    value = req.constraint.get_value() if not req.constraint.static_is_empty() else None
    con = ChoiceMap.empty()
    if value is not None:
        if isinstance(value, np.ndarray):
            value = torch_from_host(value, dev)
        if tuple(getattr(value, "shape", ())) != (n,):
            raise ValueError(f"{dist.name}: the new value of a {n}-element site must have {n} elements")
        con = ChoiceMap.choice(value)
    elem_tr = DistributionTrace(dist, eargs, prev.value, prev._elem_scores)
    ad = Diff.unknown_change(eargs) if changed else Diff.no_change(eargs)
    new_e, _w, _rd, bwd = run_edit(dist, key, elem_tr, Update(con), ad)
    total = sum_defined(new_e.score)
    out = DistributionTrace(dist, tuple(args), new_e.value, total)
    out._elem_scores = new_e.score
    # the site's weight is the difference of its SUMMED scores (distribution.py:205-224: new score - old score), not the
    # sum of the elements' differences
    from .engine import elementwise, materialize
    return out, elementwise(lambda a_, b_: a_ - b_, total, materialize(prev.get_score())), bwd


DIRICHLET_PROGRAM_MAX = 20        # components a Dirichlet site holds as ONE program (2 K live values of 64 registers)


def _concentration(dist, args):
    a = dist.canon(tuple(args))[0]
    if isinstance(a, torch.Tensor):
        return a
    a = np.asarray(a)
    return a if a.dtype != object else None


def wide_dirichlet(gen_fn, args) -> bool:
    from .distributions import _Dirichlet
    if not isinstance(gen_fn, _Dirichlet):
        return False
    a = _concentration(gen_fn, args)
    return a is not None and a.ndim == 1 and a.shape[0] > DIRICHLET_PROGRAM_MAX


def symbolic_wide_dirichlet(dist, args) -> bool:
    """the same question asked while TRACING (the concentration is an object array or a table)"""
    from .distributions import _Dirichlet
    if not isinstance(dist, _Dirichlet):
        return False
    pos, kw = (args[0], args[1]) if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple) else (args, {})
    a = kw.get("concentration", pos[0] if pos else None)
    shp = tuple(getattr(a, "shape", ()))
    return len(shp) == 1 and int(shp[0]) > DIRICHLET_PROGRAM_MAX


def _dirichlet_logpdf(x, a):
    """log Dirichlet(x; a) in the site program's operation order (distributions._Dirichlet.sym_logpdf): sum of xlogy(a - 1,
    x) in element order, minus (sum lgamma(a) in element order - lgamma(sum a in element order))"""
    from . import numpy as jnp, tracer as T
    from .engine import elementwise, sum_rows_inorder
    row = lambda v: sum_rows_inorder(v.reshape(1, -1)).reshape(())
    terms = elementwise(lambda a_, x_: T.where((a_ - 1.0) == 0.0, 0.0, (a_ - 1.0) * jnp.log(x_)), a, x)
    lbeta = elementwise(lambda lg_, sa_: lg_ - jnp.lgamma(sa_), row(elementwise(lambda a_: jnp.lgamma(a_), a)), row(a))
    return elementwise(lambda t_, lb_: t_ - lb_, row(terms), lbeta)


def dirichlet_site(dist, mode, key, args, constraint):
    """A Dirichlet site of more than DIRICHLET_PROGRAM_MAX components, for ONE trace: the operations of the site's
    program (tfp/__init__.py:125: log-space Gammas from the keys split(site key)[k], x = exp(lg - logsumexp(lg)), the
    density) as a handful of launches over the K components instead of one program that would need 2 K registers —
    the same device functions in the same order, so the same bits."""
    from . import _lib, numpy as jnp
    from .combinators import torch_from_host
    from .engine import elementwise, sum_rows_inorder
    from .program import ELEM_INDEX
    from .static import DistributionTrace, MissingAddress
    from .tracer import Expr, current_graph
    dev = _lib.get().device
    a = _concentration(dist, args)
    a = (a if isinstance(a, torch.Tensor) else torch_from_host(a, dev)).float()
    value = constraint.get_value() if constraint is not None and not constraint.static_is_empty() else None
    if value is not None:
        value = (value if isinstance(value, torch.Tensor) else torch_from_host(np.asarray(value), dev)).float()
        if tuple(value.shape) != tuple(a.shape):
            raise ValueError(f"dirichlet: a constraint of {tuple(value.shape)} for a concentration of {tuple(a.shape)}")
    if mode == "assess":
        if value is None:
            raise MissingAddress(())
        return _dirichlet_logpdf(value, a), value
    if value is None:
        def draw(k_, a_):            # element k: log Gamma(a_k) from the key split(site key)[k]
            return Expr(current_graph().add("S_LOGGAMMA", (k_.node, a_.node), imm=ELEM_INDEX, dtype="f32"))
        lg = elementwise(draw, a, key=key)
        m = torch.max(lg)                                       # (a maximum does not depend on the order)
        s = sum_rows_inorder(elementwise(lambda l_, m_: jnp.exp(l_ - m_), lg, m).reshape(1, -1)).reshape(())
        lse = elementwise(lambda s_, m_: jnp.log(s_) + m_, s, m)
        x = elementwise(lambda l_, z_: jnp.exp(l_ - z_), lg, lse)
    else:
        x = value
    score = _dirichlet_logpdf(x, a)
    tr = DistributionTrace(dist, tuple(args), x, score)
    if mode == "generate":
        return tr, (score if value is not None else _as_score(0.0))
    return (tr,)


def dirichlet_site_update(dist, prev, req, args):
    """`Update` of such a site (a new value and / or a new concentration): the density again; weight = new - old score"""
    from .engine import elementwise, materialize
    from .static import DistributionTrace
    if not isinstance(req, Update):
        raise NotImplementedError(f"{type(req).__name__} on a Dirichlet site of more than {DIRICHLET_PROGRAM_MAX} components")
    value = req.constraint.get_value() if not req.constraint.static_is_empty() else None
    tr, = dirichlet_site(dist, "simulate", None, args, ChoiceMap.choice(value if value is not None else prev.value))
    w = elementwise(lambda a_, b_: a_ - b_, tr.get_score(), materialize(prev.get_score()))
    discard = ChoiceMap.choice(prev.value) if value is not None else ChoiceMap.empty()
    return tr, w, Update(discard)


def same_args(a, b) -> bool:
    """are the arguments a site is called with now the ones its sub-trace was made with? (values, not identities:
    `jnp.zeros(n)` builds a new array at every run of the model's source)"""
    if a is b:
        return True
    if b is None:
        return False
    if isinstance(a, (tuple, list)):
        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(same_args(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(same_args(a[k], b[k]) for k in a)
    from .engine import materialize
    a, b = materialize(a), materialize(b)
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
            try:
                a_, b_ = torch.as_tensor(np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a)), \
                    torch.as_tensor(np.asarray(b.cpu() if isinstance(b, torch.Tensor) else b))
            except Exception:      # noqa: BLE001
                return False
            return a_.shape == b_.shape and bool(torch.equal(a_.to(torch.float64), b_.to(torch.float64)))
        return a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and bool(torch.equal(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        return a.shape == b.shape and bool(np.array_equal(a, b))
    import dataclasses
    if dataclasses.is_dataclass(a) and not isinstance(a, type):
        return type(a) is type(b) and all(same_args(getattr(a, f.name), getattr(b, f.name)) for f in dataclasses.fields(a))
    try:
        return bool(a == b)
    except Exception:      # noqa: BLE001
        return False


def _f32_args(v):
    """Python floats among the arguments become float32 scalars: the model's own arithmetic on them (`a * 0.5 + 0.1`)
    then runs in f32 on the host, as it does inside a site program and under the reference's jit (weak-typed Python
    scalars meet f32 arrays) — in f64 it would differ from both in the last bit"""
    if isinstance(v, float):
        return np.float32(v)
    if isinstance(v, tuple):
        return tuple(_f32_args(x) for x in v)
    if isinstance(v, list):
        return [_f32_args(x) for x in v]
    if isinstance(v, dict):
        return {k: _f32_args(x) for k, x in v.items()}
    return v


def _run_source(gen_fn, handler, args):
    from . import static
    static._HANDLERS.append(handler)
    try:
        return gen_fn.source(*_f32_args(tuple(args)))
    finally:
        static._HANDLERS.pop()


def run_gfi(gen_fn, mode, key, args, constraint=None):
    """simulate / generate / assess of `gen_fn` for one trace, site by site (see the module docstring)"""
    from .static import StaticTrace
    if key is not None and tuple(key.shape) != ():
        raise ValueError("site-by-site execution is for ONE trace (an un-batched key)")
    args = tuple(args)
    h = _Handler(mode, key, constraint)
    retval = _run_source(gen_fn, h, args)
    if mode == "assess":
        return _as_score(h.score), retval
    tr = StaticTrace(gen_fn, args, retval, h.subtraces)
    tr._site_by_site = True
    if mode == "simulate":
        return tr
    return tr, _as_score(h.weight)


def _as_score(v):
    from . import _lib
    if isinstance(v, torch.Tensor):
        return v
    return torch.full((), float(v), dtype=torch.float32, device=_lib.get().device)


def run_edit(gen_fn, key, trace, request: EditRequest, argdiffs):
    """edit(key, trace, request, argdiffs) of a trace made site by site: (new trace, weight, retdiff, backward request)"""
    from .static import StaticRequest, StaticTrace
    if not isinstance(request, (Update, Regenerate, StaticRequest)):
        raise NotImplementedError(f"site-by-site edit with a {type(request).__name__}")
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    constraint = request.constraint if isinstance(request, Update) else None
    h = _Handler("edit", key, constraint, prev=trace, request=request)
    retval = _run_source(gen_fn, h, args)
    missing = [a for a in trace.subtraces if a not in h.subtraces]
    if missing:
        raise NotImplementedError(f"an edit that removes addresses ({missing[0]!r} ...) from the trace")
    new = StaticTrace(gen_fn, args, retval, h.subtraces)
    new._site_by_site = True
    if isinstance(request, StaticRequest):
        bwd = StaticRequest(dict(h.backward))
    else:                                           # Update / Regenerate: Update(the discarded values)
        discard = ChoiceMap.empty()
        for addr, b in h.backward.items():
            sub = _discard_of(b)
            if sub is not None and not sub.static_is_empty():
                discard = discard.set(addr, sub) if not isinstance(addr, tuple) else discard.set(addr, sub)
        bwd = Update(discard)
    retdiff = Diff.no_change(retval) if all(new.subtraces[a] is trace.subtraces[a] for a in new.subtraces) \
        else Diff.unknown_change(retval)
    return new, _as_score(h.weight), retdiff, bwd


def _discard_of(bwd):
    """the choice map a backward request puts back (Update(discard); an IndexRequest's is addressed by its index)"""
    from .static import StaticRequest
    if isinstance(bwd, Update):
        return bwd.constraint
    if isinstance(bwd, IndexRequest):
        inner = _discard_of(bwd.request)
        if inner is None or not isinstance(bwd.idx, int):
            return None
        return ChoiceMap.empty().set(bwd.idx, inner)
    if isinstance(bwd, StaticRequest):
        out = ChoiceMap.empty()
        for a, r in bwd.addressed.items():
            sub = _discard_of(r)
            if sub is not None and not sub.static_is_empty():
                out = out.set(a, sub) if a != () else sub
        return out
    return None
