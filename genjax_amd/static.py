"""The `@gen` static modelling language and the engine that answers every
generative-function-interface call with ONE fused site-program launch.

Reference semantics restated here (paths under src/genjax/_src/):
  generative_functions/static.py
    StaticTrace :80-119; StaticRequest :126-128; AddressReuse :139; MissingAddress :147
    SimulateHandler :254-278, AssessHandler :297-321, GenerateHandler :341-380,
    UpdateHandler :407-466, StaticEditRequestHandler :512-566,
    RegenerateRequestHandler :616-673; StaticGenerativeFunction :725-1036; gen :1044
    per-site key = fold_in(key, counter), counter from 1 in program order for
    EVERY method (:260-263, 349-352, 419-422, 524-527, 633-636)
  generative_functions/distributions/distribution.py
    simulate :108-115, generate_choice_map :117-147,
    edit_update_with_constraint :179-244, edit_regenerate :258-300, assess :398-419
  inference/requests/rejuvenate.py :70-94 (Rejuvenate.edit)
  core/generative/requests.py :48-60 (EmptyRequest)

Mechanism (new): the model's Python source runs ONCE per call signature with
symbolic values (tracer.Expr); the handler below records sites and weight
algebra into a program graph; the compiled program is cached and each call is
a single gmx_program_run over all particles.
"""
from __future__ import annotations

import functools
import itertools
import types
import zlib
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import tracer as T
from . import sitewise
from .core.choice_map import ChoiceMap, Selection, _norm
from .core.mask import Mask
from .core.generative import (Diff, EditRequest, EmptyRequest, GenerativeFunction, IndexRequest, NoChange,
                              NotSupportedEditRequest, Regenerate, Trace, Update)
from .engine import Broadcast, Compiled, Flat, Gathered, Sym, Tracing, leaf_spec, materialize, resolve, unflatten
from .random import Key
from .tracer import Expr
from .engine import new_cache as _new_program_cache


class AddressReuse(Exception):
    """Attempt to re-write an address in a trace (static.py:139)."""


class MissingAddress(Exception):
    """Assess without a value for a sampled address (static.py:147)."""


# ---------------------------------------------------------------------------
# traces
# ---------------------------------------------------------------------------
class DistributionTrace(Trace):
    """distribution.py:59-82.  `args` are kept only when cheap (top-level call)."""

    def __init__(self, gen_fn, args, value, score):
        self.gen_fn, self.args, self.value, self.score = gen_fn, args, value, score

    def get_args(self): return self.args
    def get_retval(self): return materialize(self.value)
    def get_gen_fn(self): return self.gen_fn
    def get_score(self): return materialize(self.score)

    def get_choices(self):
        v = materialize(self.value)
        nb = len(self.batch_shape)
        if nb and not T.is_tracing():
            from .engine import expand_over_batch
            v = expand_over_batch(v, self.batch_shape)       # a launch-uniform constraint: shown with the batch axis
        cm = ChoiceMap(value=v)            # (not ChoiceMap.choice: a batch of zero particles still has its address)
        return cm.with_plate(nb) if len(getattr(v, "shape", ())) > nb else cm      # a vector-valued site: chm[j]

    @property
    def batch_shape(self):
        return tuple(getattr(self.score, "shape", ()))


MASK_FLAG = "\x00mask"      # address of the pseudo-site that carries a MaskCombinator's flag through records and traces


class StaticTrace(Trace):
    """static.py:80-119"""

    def __new__(cls, gen_fn=None, args=None, retval=None, subtraces=None):
        # whoever rebuilds a trace from its parts (gathers, stacks, slices, zero traces) gets a MaskTrace back for a
        # masked call: the flag pseudo-site among the sub-traces says so
        if cls is StaticTrace and subtraces is not None and MASK_FLAG in subtraces:
            cls = MaskTrace
        return object.__new__(cls)

    def __init__(self, gen_fn, args, retval, subtraces: "OrderedDict"):
        self.gen_fn, self.args, self.retval, self.subtraces = gen_fn, args, retval, subtraces

    def get_args(self): return self.args
    def get_retval(self): return _tree_materialize(self.retval)
    def get_gen_fn(self): return self.gen_fn

    def get_choices(self) -> ChoiceMap:
        cm = ChoiceMap.empty()
        nb = len(self.batch_shape)
        from .engine import materialize_together
        materialize_together([st.value for st in self.subtraces.values() if isinstance(st, DistributionTrace)])
        for addr, st in self.subtraces.items():
            sub = st.get_choices()
            if isinstance(st, DistributionTrace) and sub.has_value() and len(getattr(sub.get_value(), "shape", ())) > nb:
                sub = sub.with_plate(nb)      # a vector-valued site (a bare distribution under vmap): chm[addr, j]
            cm = cm.set(addr, sub)
        return cm

    def get_score(self):
        """sum of sub-trace scores in program order (static.py:102-105)."""
        scores = [st.get_score() for st in self.subtraces.values()]
        if not scores:
            return 0.0
        if len(scores) == 1:
            return scores[0]
        if T.is_tracing() and T.is_symbolic(scores):       # a trace handed to a `@gen` function as an argument
            return _sum_in_order(*scores)
        from .engine import elementwise
        return elementwise(_sum_in_order, *scores)

    @property
    def batch_shape(self):
        for st in self.subtraces.values():
            return st.batch_shape
        return ()

    def get_subtrace(self, *addr):
        if len(addr) == 1 and isinstance(addr[0], tuple):
            import warnings
            warnings.warn("get_subtrace(('a', 'b')) is deprecated: pass the components, get_subtrace('a', 'b') "
                          "(generative_function.py `get_subtrace`)", DeprecationWarning, stacklevel=2)
        addr = _norm(addr)
        tr = self
        i = 0
        while i < len(addr):
            # addresses may be registered as tuples ("a", "b") in one component
            for j in range(len(addr), i, -1):
                k = addr[i] if j == i + 1 else tuple(addr[i:j])
                if isinstance(tr, StaticTrace) and k in tr.subtraces:
                    tr = tr.subtraces[k]
                    i = j
                    break
            else:
                raise KeyError(addr)
        return tr

    get_inner_trace = get_subtrace


class _MaskFlagSite:
    """the "generative function" of the flag pseudo-site (never called: the site is written by MaskCombinator.trace_call)"""
    name = "mask flag"

    def __repr__(self):
        return "genjax.mask flag"


_MASK_FLAG_SITE = _MaskFlagSite()


def _masked_score(check, score):
    """`check * score` (mask.py:71): f32(flag) times the inner score"""
    from . import tracer as Tm
    return Tm.as_float(check) * score


class MaskTrace(StaticTrace):
    """mask.py:33-88 `MaskTrace`: the inner trace under `()` and the flag under MASK_FLAG — held as a site so that it
    travels through plates, scans, gathers and edits like any other leaf.  choices = inner choices masked by the flag,
    score = flag * inner score, return value = Mask(inner return value, flag)."""

    def __init__(self, gen_fn, args, retval, subtraces):
        super().__init__(gen_fn, args, retval, subtraces)
        if args is not None and getattr(self.inner, "args", None) is None:
            self.inner.args = tuple(args[1:])          # (`tr.inner.update(key, C.n())`: the inner call's own arguments)

    @property
    def inner(self):
        return self.subtraces[()]

    @property
    def check(self):
        return materialize(self.subtraces[MASK_FLAG].value)

    def get_choices(self) -> ChoiceMap:
        return self.inner.get_choices().mask(_host_flag(self.check))

    def get_score(self):
        c, sc = self.check, self.inner.get_score()
        if isinstance(c, (bool, np.bool_)):
            if isinstance(sc, torch.Tensor):
                return sc * float(c)
            return float(c) * sc
        from .engine import elementwise
        return elementwise(_masked_score, c, sc)

    @property
    def batch_shape(self):
        return self.inner.batch_shape

    def get_subtrace(self, *addr):
        return self.inner.get_subtrace(*addr)

    get_inner_trace = get_subtrace


def _host_flag(c):
    """a flag fixed while the program was traced stays a VALUE in choice maps (`np.bool_`, as under the reference's jit:
    the masked choices keep their addresses), instead of resolving the mask on the spot (ChoiceMap.mask: bool)"""
    return np.bool_(c) if isinstance(c, (bool, np.bool_)) else c


def _sum_in_order(*terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


class VmapTrace(Trace):
    """Trace of a `Vmap` call: the inner trace's leaves carry a trailing plate
    axis; the score is the plate sum (vmap.py VmapTrace)."""

    def __init__(self, gen_fn, inner: "StaticTrace", score, retval, args=None):
        self.gen_fn, self.inner, self.score, self.retval, self.args = gen_fn, inner, score, retval, args
        self.subtraces = inner.subtraces

    def get_args(self): return self.args
    def get_retval(self): return _tree_materialize(self.retval)
    def get_gen_fn(self): return self.gen_fn
    def get_score(self): return materialize(self.score)

    def get_choices(self):
        """the inner choices; an integer address component reads one element of the plate (`chm[j, "x"]`)"""
        # (a plate whose element is itself a plate / scan — `f.vmap().vmap()` — holds its trace flat, the sites with one
        #  more axis each: `chm[i, j, "x"]` peels them off one by one)
        depth, g = 1, getattr(self.gen_fn, "gen_fn", None)
        while g is not None and type(g).__name__ in ("Vmap", "_Repeat", "Scan") and depth < 4:
            depth, g = depth + 1, getattr(g, "gen_fn", None)
        return self.inner.get_choices().with_plate(len(self.batch_shape), depth)

    def get_subtrace(self, *addr): return self.inner.get_subtrace(*addr)

    @property
    def batch_shape(self):
        return tuple(getattr(self.score, "shape", ()))


def _zero_like_trace(tr):
    z = lambda v: _tree_map_leaves(v, lambda t: torch.zeros_like(t) if isinstance(t, torch.Tensor) else
                                   (type(t)(0) if isinstance(t, (int, float, bool)) else t))
    if isinstance(tr, DistributionTrace):
        return DistributionTrace(tr.gen_fn, tr.args, z(materialize(tr.value)), z(materialize(tr.score)))
    if isinstance(tr, StaticTrace):
        subs = OrderedDict((a, _zero_like_trace(st)) for a, st in tr.subtraces.items())
        return StaticTrace(tr.gen_fn, tr.args, z(_tree_materialize(tr.retval)), subs)
    raise NotImplementedError(f"get_zero_trace of a {type(tr).__name__}")


def _tree_map_leaves(v, fn):
    import dataclasses
    from .core.mask import Mask
    if isinstance(v, Mask):
        return Mask(_tree_map_leaves(v.value, fn), _tree_map_leaves(v.flag, fn))
    if isinstance(v, tuple):
        return tuple(_tree_map_leaves(x, fn) for x in v)
    if isinstance(v, list):
        return [_tree_map_leaves(x, fn) for x in v]
    if isinstance(v, dict):
        return {k: _tree_map_leaves(x, fn) for k, x in v.items()}
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        return dataclasses.replace(v, **{f_.name: _tree_map_leaves(getattr(v, f_.name), fn) for f_ in dataclasses.fields(v)})
    return fn(v)


def _tree_materialize(v):
    import dataclasses
    from .core.mask import Mask
    if isinstance(v, Mask):
        return Mask(_tree_materialize(v.value), materialize(v.flag))
    if dataclasses.is_dataclass(v) and not isinstance(v, type) and not getattr(v, "__gmx_static__", False):
        return dataclasses.replace(v, **{f_.name: _tree_materialize(getattr(v, f_.name)) for f_ in dataclasses.fields(v)})
    if isinstance(v, tuple):
        return tuple(_tree_materialize(x) for x in v)
    if isinstance(v, list):
        return [_tree_materialize(x) for x in v]
    if isinstance(v, dict):
        return {k: _tree_materialize(x) for k, x in v.items()}
    return materialize(v)


def _plain_retval(v):
    """a return value as tuples / dicts of launch values: a Mask (a MaskCombinator's) becomes {"mvalue", "mflag"}"""
    from .core.mask import Mask
    if isinstance(v, Mask):
        return {"mvalue": _plain_retval(v.value), "mflag": v.flag}
    if isinstance(v, tuple):
        return tuple(_plain_retval(x) for x in v)
    return v


def _trace_tree(tr):
    """Trace -> plain pytree of launch values (for Flat)."""
    if isinstance(tr, DistributionTrace):
        es = getattr(tr, "_elem_scores", None)
        if es is not None and tuple(getattr(es, "shape", ())) == tuple(getattr(tr.value, "shape", (None,))):
            return {"value": tr.value, "score": tr.score, "escore": es}      # a bare distribution's plate: per-element scores
        return {"value": tr.value, "score": tr.score}
    if isinstance(tr, StaticTrace):
        return {"sub": {a: _trace_tree(s) for a, s in tr.subtraces.items()}, "retval": _plain_retval(tr.retval)}
    if isinstance(tr, VmapTrace):
        return {"vmap": _trace_tree(tr.inner), "score": tr.score, "retval": _plain_retval(tr.retval)}
    raise TypeError(f"cannot edit a trace of type {type(tr).__name__}")


# ---------------------------------------------------------------------------
# edit requests specific to the static language / inference
# ---------------------------------------------------------------------------
class StaticRequest(EditRequest):
    """static.py:126-128: address -> sub-request."""
    __match_args__ = ("addressed",)

    def __init__(self, addressed: dict):
        self.addressed = dict(addressed)


class Rejuvenate(EditRequest):
    """inference/requests/rejuvenate.py:44-94."""
    __match_args__ = ("proposal", "argument_mapping")

    def __init__(self, proposal, argument_mapping):
        self.proposal, self.argument_mapping = proposal, argument_mapping

    def edit(self, key, tr, argdiffs):
        return StaticRequest({(): self}).edit(key, tr, argdiffs) if isinstance(tr, DistributionTrace) \
            else super().edit(key, tr, argdiffs)


# ---------------------------------------------------------------------------
# trace-time handler
# ---------------------------------------------------------------------------
_HANDLERS: list = []


def trace(addr, gen_fn, args):
    """`gen_fn(*args) @ addr` (static.py:175-193)."""
    if not _HANDLERS:
        raise RuntimeError("`@ addr` used outside of a @gen function call")
    return _HANDLERS[-1].handle(addr, gen_fn, tuple(args))


def _selkey(sel):
    """Structural cache key of a Selection."""
    from .core import choice_map as cm
    if isinstance(sel, cm._All): return ("all",)
    if isinstance(sel, cm._None): return ("none",)
    if isinstance(sel, cm._Leaf): return ("leaf",)
    if isinstance(sel, cm._Static): return ("s", sel.comp, _selkey(sel.sub))
    if isinstance(sel, cm._Or): return ("or", _selkey(sel.a), _selkey(sel.b))
    if isinstance(sel, cm._And): return ("and", _selkey(sel.a), _selkey(sel.b))
    if isinstance(sel, cm._Complement): return ("not", _selkey(sel.s))
    if isinstance(sel, cm._Chm): return ("chm", tuple(sel.chm.addresses()))
    return ("id", id(sel))


def _fnkey(fn):
    code = getattr(fn, "__code__", None)
    if code is None:
        return ("id", id(fn))
    cells = ()
    if fn.__closure__:
        vals = []
        for c in fn.__closure__:
            try:
                v = c.cell_contents
                hash(v)
                vals.append(v)
            except Exception:
                vals.append(("id", id(c)))
        cells = tuple(vals)
    return ("fn", code, cells)


class _ReqSpec:
    """Static description of an edit request + where its values sit among the leaves."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)


def _flatten_request(req, flat: Flat):
    """-> (_ReqSpec, hashable key)."""
    if req is None or isinstance(req, EmptyRequest):
        return _ReqSpec("empty"), ("empty",)
    if isinstance(req, Update):
        tree = flat.add(req.constraint)
        return _ReqSpec("update", tree=tree), ("update", tree)
    if isinstance(req, Regenerate):
        return _ReqSpec("regen", selection=req.selection), ("regen", _selkey(req.selection))
    if isinstance(req, StaticRequest):
        subs, keys = {}, []
        for a, r in req.addressed.items():
            s, k = _flatten_request(r, flat)
            subs[_norm(a) if a != () else ()] = s
            keys.append((a, k))
        return _ReqSpec("static", subs=subs), ("static", tuple(keys))
    if isinstance(req, IndexRequest):
        sub, k = _flatten_request(req.request, flat)
        if isinstance(req.idx, int):
            return _ReqSpec("index", idx=req.idx, sub=sub), ("index", req.idx, k)
        tree = flat.add(req.idx)                      # one index per particle: a launch value
        return _ReqSpec("index", idx=None, idx_tree=tree, sub=sub), ("index", "traced", tree, k)
    if isinstance(req, Rejuvenate):
        return (_ReqSpec("rejuv", proposal=req.proposal, argmap=req.argument_mapping),
                ("rejuv", _gfkey(req.proposal), _fnkey(req.argument_mapping)))
    raise NotSupportedEditRequest(req)


_UID = itertools.count(1)


def _gfkey(gf):
    """Cache key of a generative function: a serial number stamped on first use (id() would be recycled
    after garbage collection) PLUS a fingerprint of every Python value its source captures — closure cells and
    the module globals its code names, through wrapped / called generative functions.  The reference re-traces on
    every GFI call (jit aside), so a model that reads `scale` from its enclosing scope sees a new value on the
    next call; a cache keyed on the function alone would keep replaying the program traced with the old one."""
    uid = getattr(gf, "_gmx_uid", None)
    if uid is None:
        uid = next(_UID)
        try:
            object.__setattr__(gf, "_gmx_uid", uid)
        except Exception:
            _KEEP.append(gf)
            return ("gfid", id(gf), _capture_fp(gf, 0))
    return ("gf", uid, _capture_fp_cached(gf))


def _capture_fp_cached(gf):
    """`_capture_fp(gf, 0)` without walking the capture graph on every GFI call (50 us per call on the
    functional SMC path).  The walk is done once and remembered together with the list of (getter, value seen) pairs it
    read; a later call only re-reads those slots and compares CHEAP tokens — identity for functions / generative
    functions / large arrays, (identity, version) for tensors, the value itself for scalars and small containers.  Any
    difference (a rebound cell, a global assigned, a tensor written in place, a small array edited) re-walks."""
    memo = gf.__dict__.get("_gmx_fp_memo") if hasattr(gf, "__dict__") else None
    if memo is not None:
        slots, seen, tokens, fp = memo
        try:
            for k, get in enumerate(slots):
                v = get()
                tok = tokens[k]
                if tok is None:                 # an immutable value or an object compared by identity
                    if v is not seen[k] and not (type(v) is type(seen[k]) and isinstance(v, (int, float, str, bytes)) and v == seen[k]):
                        break
                elif _cheap_token(v) != tok:
                    break
            else:
                return fp
        except Exception:
            pass
    slots = []
    _collect_slots(gf, 0, slots, set())
    fp = _capture_fp(gf, 0)
    try:
        seen = [get() for get in slots]
        tokens = [_cheap_token(v) if isinstance(v, (torch.Tensor, np.ndarray, np.generic, tuple, list, dict)) else None
                  for v in seen]
        object.__setattr__(gf, "_gmx_fp_memo", (slots, seen, tokens, fp))
    except Exception:
        pass
    return fp


def _cheap_token(v):
    if v is None or isinstance(v, (bool, int, float, str, bytes, complex)):
        return (type(v), v) if v == v else "nan"
    if isinstance(v, torch.Tensor):
        return ("t", id(v), v._version)
    if isinstance(v, (np.ndarray, np.generic, tuple, list, dict)):
        return _value_fp(v, _CAPTURE_DEPTH + 1)        # contents (small) or the sampled hash (large): never recurses
    return ("o", id(v))


def _collect_slots(obj, depth, out, seen):
    """the getters of every cell / global the fingerprint of `obj` reads, through nested functions and wrapped
    generative functions (the same graph _capture_fp walks)"""
    if depth > _CAPTURE_DEPTH or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, GenerativeFunction):
        fn = getattr(obj, "_fn", None)
        if fn is not None:
            _collect_slots(fn, depth, out, seen)
            part = getattr(obj, "_partial", ())
            if part:
                out.append(lambda o=obj: tuple(getattr(o, "_partial", ())))
        for attr in ("gen_fn", "kernel_gen_fn", "inner"):
            if hasattr(obj, "__dict__") and isinstance(obj.__dict__.get(attr), GenerativeFunction):
                out.append(lambda o=obj, a=attr: o.__dict__.get(a))
                _collect_slots(obj.__dict__[attr], depth + 1, out, seen)
        return
    fn = getattr(obj, "__func__", obj)
    if not isinstance(fn, types.FunctionType):
        return
    cells, g, names = _capture_plan(fn)
    for c in cells:
        def get(c=c):
            try:
                return c.cell_contents
            except ValueError:
                return None
        out.append(get)
        v = get()
        if isinstance(v, (GenerativeFunction, types.FunctionType, types.MethodType)):
            _collect_slots(v, depth + 1, out, seen)
        elif isinstance(v, (tuple, list)) and len(v) <= 64:
            for x in v:
                if isinstance(x, (GenerativeFunction, types.FunctionType, types.MethodType)):
                    _collect_slots(x, depth + 1, out, seen)
    for n in names:
        v = g.get(n)
        if isinstance(v, (types.ModuleType, type)) or v is fn:
            continue
        out.append(lambda g=g, n=n: g.get(n))
        if isinstance(v, (GenerativeFunction, types.FunctionType, types.MethodType)):
            _collect_slots(v, depth + 1, out, seen)


_CAPTURE_DEPTH = 3        # model -> helper / sub-model -> helper


def _code_names(code, out):
    out.update(code.co_names)
    for c in code.co_consts:
        if hasattr(c, "co_names"):
            _code_names(c, out)


def _capture_plan(fn):
    """(closure cells, (globals dict, names)) of a Python function, computed once per function object."""
    plan = getattr(fn, "_gmx_capture_plan", None)
    if plan is None:
        code = getattr(fn, "__code__", None)
        cells = tuple(getattr(fn, "__closure__", None) or ())
        names = ()
        g = getattr(fn, "__globals__", None)
        if code is not None and g is not None:
            ns = set()
            _code_names(code, ns)
            names = tuple(sorted(n for n in ns if n in g))
        plan = (cells, g, names)
        try:
            fn._gmx_capture_plan = plan
        except Exception:
            pass
    return plan


def _value_fp(v, depth):
    if v is None or isinstance(v, (bool, int, float, str, bytes, complex)):
        return v if not isinstance(v, float) else ("f", v.hex() if v == v else "nan")
    if isinstance(v, np.generic):
        return ("np", v.dtype.str, v.tobytes())
    if isinstance(v, np.ndarray):
        if v.size <= 4096:
            return ("nd", v.dtype.str, v.shape, v.tobytes())
        # a large captured array: its identity plus a hash of 4096 evenly spaced elements and of both ends, so that an
        # in-place edit is noticed unless it misses every sampled element (documented limit: DESIGN.md §8; pass large
        # data as an ARGUMENT — a launch leaf — and it is never part of a cache key)
        flat = v.reshape(-1)
        step_ = max(1, flat.size // 4096)
        return ("ndh", id(v), v.dtype.str, v.shape, zlib.crc32(np.ascontiguousarray(flat[::step_]).tobytes()),
                zlib.crc32(np.ascontiguousarray(flat[-64:]).tobytes()))
    if isinstance(v, torch.Tensor):
        return ("t", id(v), v._version)
    if isinstance(v, (tuple, list)) and len(v) <= 64:
        return (type(v).__name__,) + tuple(_value_fp(x, depth) for x in v)
    if isinstance(v, dict) and len(v) <= 64:
        try:
            return ("dict",) + tuple((k, _value_fp(x, depth)) for k, x in sorted(v.items(), key=lambda kv: repr(kv[0])))
        except Exception:
            return ("o", id(v))
    if isinstance(v, GenerativeFunction):
        uid = getattr(v, "_gmx_uid", None)
        return ("gf", uid if uid is not None else id(v), _capture_fp(v, depth + 1))
    if isinstance(v, (types.FunctionType, types.MethodType)):
        return ("fn", id(v), _function_fp(v, depth + 1))
    return ("o", id(v))            # modules, classes, other objects: identity


def _function_fp(fn, depth):
    if depth > _CAPTURE_DEPTH:
        return ()
    fn = getattr(fn, "__func__", fn)
    cells, g, names = _capture_plan(fn)
    out = []
    for c in cells:
        try:
            out.append(_value_fp(c.cell_contents, depth))
        except ValueError:          # empty cell
            out.append(None)
    for n in names:
        v = g.get(n)
        if isinstance(v, (types.ModuleType, type)) or v is fn:
            continue
        out.append(_value_fp(v, depth))
    return tuple(out)


def _capture_fp(gf, depth):
    """Fingerprint of the values a generative function's SOURCE reads from outside its arguments."""
    if depth > _CAPTURE_DEPTH:
        return ()
    out = []
    fn = getattr(gf, "_fn", None)
    if fn is not None:
        out.append(_function_fp(fn, depth))
        part = getattr(gf, "_partial", ())
        if part:
            out.append(_value_fp(tuple(part), depth))
    for attr in ("gen_fn", "kernel_gen_fn", "inner"):       # combinators / wrappers around another function
        inner = gf.__dict__.get(attr) if hasattr(gf, "__dict__") else None
        if isinstance(inner, GenerativeFunction):
            out.append(_capture_fp(inner, depth + 1))
    return tuple(out)


_KEEP: list = []


def _depends(node, changed: set, memo: dict, table_changed=None, var_updates=None) -> bool:
    """Does `node` depend on any node in `changed` — or read a table flagged changed? (incremental.py change propagation)
    A loop-carried value depends on what its iterations assign to it (program.Graph.set_var)."""
    stack = [node]
    seen = []
    while stack:
        n = stack.pop()
        if n is None:
            continue
        r = memo.get(n.idx)
        if r is True:
            for s in seen:
                memo[s] = True
            return True
        if r is False:
            continue
        if n.idx in changed or (table_changed is not None and n.op in ("LDTAB", "LDIN", "LDINX") and table_changed(n)):
            memo[n.idx] = True
            for s in seen:
                memo[s] = True
            return True
        memo[n.idx] = False      # provisional; flipped to True above if a dependency is found
        seen.append(n.idx)
        stack.extend(a for a in n.args if a is not None)
        if n.op == "LOOPVAR" and var_updates:
            stack.extend(var_updates.get(n.idx, ()))
    return False


def _nodes_of(v):
    if isinstance(v, Expr):
        return [v.node]
    if isinstance(v, T.LazyVec):
        return v.dep_nodes()
    if hasattr(v, "token"):                            # engine.StepInput2: the leaf's stand-in node
        return [v.token()]
    if isinstance(v, np.ndarray) and v.dtype == object:
        return [x.node for x in v.reshape(-1) if isinstance(x, Expr)]
    if isinstance(v, (tuple, list)):
        out = []
        for x in v:
            out += _nodes_of(x)
        return out
    if isinstance(v, dict):
        out = []
        for x in v.values():
            out += _nodes_of(x)
        return out
    return []


class _Ctx:
    """State shared by all handlers of one program trace."""

    def __init__(self, tr: Tracing):
        self.tr = tr
        self.changed: set = set()        # node indices whose value differs from the previous trace
        self.memo: dict = {}
        self.changed_slots: set = set()  # table slots whose contents differ from the previous trace's (changed table arguments)
        self.changed_tables: list = []   # ... and host tables flagged changed before they were given a slot
        self.changed_in_slots: set = set()   # INPUT slots of per-particle vectors of > 16 elements (engine.StepInput) that
        #                                      changed: a read at a loop's iteration number is a node of its own
        self._slot_memo: dict = {}
        self.store_sites = True          # False: the caller stores only what it needs (MinimalGenerate)
        self.gate = None                 # an edit applies only where this boolean holds (IndexRequest around a loop)

    def mark_changed(self, v):
        for n in _nodes_of(v):
            self.changed.add(n.idx)
        self._mark_tables(v)
        self.memo.clear()

    def _mark_tables(self, v):
        """a changed argument that is a TABLE (a launch-uniform vector of more than 16 elements: read with OP_LDTAB at a
        run-time index — element j of a mapped argument in a plate's loop, `means[idx]`): every read of it is changed"""
        if isinstance(v, (tuple, list)):
            for x in v:
                self._mark_tables(x)
        elif isinstance(v, dict):
            for x in v.values():
                self._mark_tables(x)
        elif hasattr(v, "token") and hasattr(v, "_slot"):      # engine.StepInput2: a per-particle leaf with several axes
            slot = v._slot
            self.changed_in_slots.update(int(x) for x in (slot if isinstance(slot, (list, tuple)) else [slot]))
        elif isinstance(v, np.ndarray):
            from .engine import StepInput
            slot = getattr(v, "_slot", None)
            if isinstance(v, StepInput):              # a per-particle [T, n] leaf: every step read of its slot(s) is changed
                if slot is not None:
                    self.changed_in_slots.update(int(x) for x in (slot if isinstance(slot, list) else [slot]))
            elif slot is not None:                    # numpy.RuntimeTable / a TableArray row: a device or pooled table
                self.changed_slots.add(int(slot))
            if v.dtype != object and v.size > 0:      # a host table (numpy.TableArray, or the array it will be made from)
                self.changed_tables.append(np.asarray(v).reshape(-1))
                self._slot_memo.clear()

    def _table_changed(self, node) -> bool:
        slot = getattr(node, "slot", None)
        if slot is None:
            return False
        if node.op in ("LDIN", "LDINX"):
            return slot in self.changed_in_slots
        if slot in self.changed_slots:
            return True
        if not self.changed_tables:
            return False
        r = self._slot_memo.get(slot)
        if r is None:
            tabs = self.tr.graph.__dict__.get("tables", [])
            t = tabs[slot] if slot < len(tabs) else None
            r = False
            if t is not None:
                flat = np.asarray(t).reshape(-1)
                r = any(c.size == flat.size and np.array_equal(c, flat) for c in self.changed_tables)
            self._slot_memo[slot] = r
        return r

    def args_changed(self, args) -> bool:
        tab = self._table_changed if (self.changed_slots or self.changed_tables or self.changed_in_slots) else None
        vu = self.tr.graph.__dict__.get("_var_updates")
        return any(_depends(n, self.changed, self.memo, tab, vu) for n in _nodes_of(args))


class _SiteRec:
    __slots__ = ("gen_fn", "value", "score", "discard", "origins", "escore")

    def __init__(self, gen_fn, value, score, discard=None):
        self.gen_fn, self.value, self.score, self.discard = gen_fn, value, score, discard
        self.origins = None          # (value, score, discard) origins once stored
        self.escore = None           # a BARE distribution's plate (`normal.vmap()`): its per-element scores — the trace keeps
        #                              them beside the plate sum, as the reference's inner trace does (vmap.py:73-75), so that an
        #                              edit under CHANGED arguments has each element's old score (new_j - old_j, summed: :265-275)


def _origin_of(tr, v):
    if v is None:
        return None
    if isinstance(v, Sym) and v.origin is not None:
        return v.origin
    return tr.emit_output(v.value if isinstance(v, Sym) else v)


def _store_site(ctx, rec: "_SiteRec"):
    """Emit the site's stores NOW (program order), so its registers die here
    instead of staying live until the end of the program."""
    if ctx.store_sites and rec.origins is None:
        tr = ctx.tr
        rec.origins = (_origin_of(tr, rec.value), _origin_of(tr, rec.score), _origin_of(tr, rec.discard))
    return rec


class DeferralAbort(Exception):
    """raised while tracing when a deferred plate's values are computed with (or it is not the model's last site): the
    call is traced again with the plate as a loop"""


class DeferredOutput:
    """what a DEFERRED plate hands back to the model while it is traced (run_gfi `defer`): its values do not exist yet —
    the plate runs after the program, with its elements on the launch axis — so they can be returned, not computed with"""

    def __init__(self, index):
        self.index = index

    def _no(self, *a, **k):
        raise DeferralAbort()
    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = __getitem__ = _no
    __neg__ = __pow__ = __lt__ = __gt__ = __le__ = __ge__ = __iter__ = __len__ = __array__ = __float__ = _no


class _DeferredPlateRec:
    """a large plate of a particle-batched call, lifted out of the program (Vmap.trace_call / run_gfi `defer`)"""

    def __init__(self, vm, mode, key_path, arg_origins, axes, con_leaves, n, index):
        self.gen_fn, self.mode, self.key_path, self.arg_origins, self.axes = vm, mode, key_path, arg_origins, axes
        self.con_leaves, self.n, self.index = con_leaves, n, index
        self.retval = DeferredOutput(index)


class _CallRec:
    """Result of tracing one generative-function call (nested or top-level)."""

    def __init__(self, gen_fn):
        self.gen_fn = gen_fn
        self.sites: "OrderedDict" = OrderedDict()   # addr -> _SiteRec | _CallRec
        self.retval = None
        self.weight = None
        self.score = None


class Handler:
    """One handler per (nested) static generative function call.

    mode: simulate | generate | assess | update | regen | static_edit
    """

    def __init__(self, ctx: _Ctx, mode: str, key, constraint=None, prev=None, req=None, req_leaves=None):
        self.ctx, self.mode, self.key = ctx, mode, key
        self.constraint = constraint if constraint is not None else ChoiceMap.empty()
        self.prev = prev               # symbolic previous trace: {"sub": {...}} / {"value","score"}
        self.req = req                 # _ReqSpec
        self.req_leaves = req_leaves   # leaf index -> Sym (for Update constraints inside requests)
        self.counter = 1
        self.rec = None
        g = ctx.tr.graph
        self.weight = Expr(g.const_f32(0.0))
        self.score = Expr(g.const_f32(0.0))

    # -- keys ------------------------------------------------------------------
    def fresh_key(self):
        c = self.counter
        self.counter += 1
        if self.key is None:
            return None
        g = self.ctx.tr.graph
        return Expr(g.add("KDERIVE", (self.key.node,), imm=c, dtype="key"))

    # -- site dispatch -------------------------------------------------------------
    def handle(self, addr, gen_fn, args):
        for comp in (addr if isinstance(addr, tuple) else (addr,)):
            if not isinstance(comp, (str, int, np.integer)):
                # a traced value is not an address (test_static_gen_fn.py:790-799): addresses are static
                raise TypeError(f"static addresses are strings, integers or tuples of them (got {type(comp).__name__})")
        if addr in self.rec.sites:
            raise AddressReuse(addr)
        sub_key = self.fresh_key()
        sub_con = self.constraint.get_submap(addr)
        sub_prev = None
        if self.prev is not None:
            try:
                sub_prev = self.prev["sub"][addr]
            except KeyError:
                raise KeyError(f"address {addr!r} is not in the previous trace") from None
        sub_req = self._subrequest(addr)
        if sub_req is not None and sub_req.kind == "update" and self.mode == "static_edit":
            sub_con = sub_req.constraint          # Update nested inside a StaticRequest
        try:
            rec, retval, w, s = call_gen_fn(self.ctx, self.mode, gen_fn, sub_key, args, sub_con, sub_prev,
                                            sub_req, self.req_leaves, addr)
        except MissingAddress as e:
            # the exception names the address (test_static_gen_fn.py:309-315: `exc.value.args == ("y2",)`); below a
            # nested call, the caller's address comes first
            inner = tuple(a for a in e.args if a != ())
            raise MissingAddress(*((addr,) + inner)) from None
        self.rec.sites[addr] = rec
        if w is not None:
            self.weight = self.weight + w          # static.py:377 / 454 / 559 / 668
        if s is not None and self.mode == "assess":
            self.score = self.score + s            # static.py:319
        return T.sym_array(retval)                 # (a vector of traced values: `means[z]` with a traced z selects)

    def _subrequest(self, addr):
        r = self.req
        if r is None:
            return None
        if r.kind == "static":
            return r.subs.get(_norm(addr), _ReqSpec("empty"))
        if r.kind == "regen":
            return _ReqSpec("regen", selection=r.selection(addr))
        return r                   # update: the constraint lookup goes through self.constraint


def _sym_arrays(tree):
    """the arguments a model's source is called with: short vectors of traced values (registers — a plate's element row,
    a scan's carry) as tracer.SymArray, so that `trans[z]` with a traced z selects"""
    if isinstance(tree, tuple):
        return tuple(_sym_arrays(x) for x in tree)
    if isinstance(tree, list):
        return [_sym_arrays(x) for x in tree]
    if isinstance(tree, dict):
        return {k: _sym_arrays(v) for k, v in tree.items()}
    return T.sym_array(tree)


def _leaf_call(ctx: _Ctx, mode, dist, key, args, constraint: ChoiceMap, prev, req, req_leaves):
    """Leaf (Distribution) semantics; returns (_SiteRec, retval, weight, score)."""
    g = ctx.tr.graph
    if getattr(ctx, "sitewise", False) and (sitewise.symbolic_vector_site_size(args) is not None
                                            or sitewise.symbolic_wide_dirichlet(dist, args)) \
            and not getattr(g, "_in_loop", False) and not g.loop_counts:
        raise sitewise.NeedsSiteBySite()      # ONE trace and a site of thousands of elements: not unrolled (sitewise.vector_site)
    args = dist.canon(args)
    cval = constraint.get_value() if constraint is not None else None
    zero = None
    looped = _vector_site_loop(ctx, mode, dist, key, args, cval, prev, req)
    if looped is not None:
        return looped
    if mode == "simulate":
        v = dist.sym_sample(key, args)
        s = dist.sym_logpdf(v, args)
        return _SiteRec(dist, v, s), v, None, s
    if mode == "assess" and isinstance(cval, Mask):
        cval = cval.value                 # distribution.py:405-416: assess scores a masked value whatever its flag
    if isinstance(cval, Mask) and isinstance(cval.flag, bool):
        cval = cval.value if cval.flag else None             # decided on the host: plain constrained / unconstrained
    if isinstance(cval, Mask):
        # a runtime-conditional constraint: `lax.cond(flag, importance, simulate)` (generate, distribution.py:129-142)
        # / `FlagOp.cond(flag, new value, old value)` (update, :189-224) becomes a select between the two branches'
        # results — the value is chosen first, its log-density is evaluated once (the same function in both branches)
        flag = cval.flag.value if isinstance(cval.flag, Sym) else cval.flag
        mval = cval.value.value if isinstance(cval.value, Sym) else cval.value
        if mode == "generate":
            vs = dist.sym_sample(key, args)
            v = T.where(flag, mval, vs)
            s = dist.sym_logpdf(v, args)
            return _SiteRec(dist, v, s), v, T.where(flag, s, 0.0), s
        if mode == "update" or (req is not None and req.kind == "update"):
            pv_sym, ps_sym = prev["value"], prev["score"]
            v = T.where(flag, mval, pv_sym.value)
            s = dist.sym_logpdf(v, args)
            ctx.mark_changed(v)
            return _SiteRec(dist, v, s, discard=pv_sym), v, s - ps_sym.value, s
        raise NotSupportedEditRequest(f"a Mask constraint in mode {mode!r}")
    if mode == "generate":
        if cval is None:
            v = dist.sym_sample(key, args)
            s = dist.sym_logpdf(v, args)
            return _SiteRec(dist, v, s), v, zero, s          # w = 0 (distribution.py:124-127)
        v = cval.value if isinstance(cval, Sym) else cval
        s = dist.sym_logpdf(v, args)
        return _SiteRec(dist, cval, s), v, s, s              # w = score = logpdf (:144-147)
    if mode == "assess":
        if cval is None:
            raise MissingAddress(())
        v = cval.value if isinstance(cval, Sym) else cval
        s = dist.sym_logpdf(v, args)
        return _SiteRec(dist, cval, s), v, None, s
    # ---- edits: need the previous site ----
    pv_sym, ps_sym = prev["value"], prev["score"]
    pv, ps = pv_sym.value, ps_sym.value
    kind = req.kind if req is not None else "empty"
    if kind == "update":
        if cval is None:
            # re-score the old value under the new args (distribution.py:225-233); the
            # Update path visits every site, unchanged args contribute exactly 0
            if not ctx.args_changed(args):
                return _SiteRec(dist, pv_sym, ps_sym), pv, None, ps
            s = dist.sym_logpdf(pv, args)
            return _SiteRec(dist, pv_sym, s), pv, s - ps, s
        v = cval.value if isinstance(cval, Sym) else cval
        s = dist.sym_logpdf(v, args)
        ctx.mark_changed(v)
        return _SiteRec(dist, cval, s, discard=pv_sym), v, s - ps, s       # (:235-242)
    if kind == "regen":
        if req.selection.check():
            v = dist.sym_sample(key, args)
            s = dist.sym_logpdf(v, args)
            ctx.mark_changed(v)
            return _SiteRec(dist, v, s, discard=pv_sym), v, s - ps, s     # (:266-277)
        kind = "empty"
    if kind == "empty":
        if not ctx.args_changed(args):
            return _SiteRec(dist, pv_sym, ps_sym), pv, None, ps            # requests.py:56-57
        s = dist.sym_logpdf(pv, args)
        return _SiteRec(dist, pv_sym, s), pv, s - ps, s
    if kind == "rejuv":
        # Rejuvenate.edit (rejuvenate.py:70-94), literally
        k_new = Expr(g.add("KDERIVE", (key.node,), imm=0, dtype="key"))   # key, sub_key = split(key)
        sub = Expr(g.add("KDERIVE", (key.node,), imm=1, dtype="key"))
        fwd_args = req.argmap(ChoiceMap.choice(pv))
        if not isinstance(fwd_args, tuple):
            fwd_args = (fwd_args,)
        keep, was_deferred = ctx.store_sites, getattr(ctx, "sites_deferred", False)
        ctx.store_sites = ctx.sites_deferred = False          # the proposal's own trace is never materialised
        prec, pret, _, pscore = call_gen_fn(ctx, "simulate", req.proposal, sub, fwd_args, ChoiceMap.empty(),
                                            None, None, None, ())
        proposed = _rec_choices(prec)
        fwd_score = _rec_score(prec)
        nv = proposed.get_value()
        if nv is None:
            raise NotImplementedError("Rejuvenate at a distribution site needs a distribution-valued proposal")
        from .distributions import Distribution as _Dist
        if isinstance(req.proposal, _Dist) and T._long_vector(pret):
            nv = pret          # (a long vector drawn in a counted loop: the READABLE form of the stored values, engine.StepAlias)
        s = None
        if T._long_vector(nv) and ctx.gate is None:
            # a proposal over a LONG vector-valued site (`Rejuvenate(normal, lambda chm: (chm.get_value(), 0.5))` on an
            # 8-schools theta of 1 000 schools): the proposal drew in a counted loop (its simulate is a vector-valued site
            # itself); the new values are scored in one as well, instead of one unrolled density per element
            out_ = _vector_site_loop(ctx, "assess", dist, None, args, nv, None, None)
            s = out_[3] if out_ is not None else None
        if s is None:
            s = dist.sym_logpdf(nv, args)        # Update(proposed).edit(key, tr, argdiffs)
        w = s - ps
        bwd_args = req.argmap(ChoiceMap.choice(pv))
        if not isinstance(bwd_args, tuple):
            bwd_args = (bwd_args,)
        _, _, _, bwd_score = call_gen_fn(ctx, "assess", req.proposal, None, bwd_args, ChoiceMap.choice(pv),
                                         None, None, None, ())
        ctx.store_sites, ctx.sites_deferred = keep, was_deferred
        final = (w + bwd_score) - fwd_score
        ctx.mark_changed(nv)
        del k_new
        return _SiteRec(dist, nv, s, discard=pv_sym), nv, final, s
    raise NotSupportedEditRequest(kind)


VECTOR_SITE_LOOP_MIN = 17       # elements from which a vector-valued site under a particle batch runs as a counted loop


_VECTOR_SITE_LOOPS_OFF = [0]


def _unrolled_when_values_are_used(fn):
    """a tracing entry point: when the model turns out to compute with the values of a vector-valued site that was
    lowered to a counted loop (engine.VectorSiteValueUsed — they exist in memory only), the call is traced again with
    such sites unrolled, the form that carries a few hundred elements (chains of launches beyond one); the program cache
    then holds that form under the same key"""
    import functools
    from .engine import VectorSiteValueUsed

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except VectorSiteValueUsed:
            if _VECTOR_SITE_LOOPS_OFF[0]:
                raise
        _VECTOR_SITE_LOOPS_OFF[0] += 1
        try:
            return fn(*a, **k)
        finally:
            _VECTOR_SITE_LOOPS_OFF[0] -= 1
    return wrapped


def _vector_site_loop(ctx, mode, dist, key, args, cval, prev, req):
    """A vector-valued distribution site of MANY elements under a particle batch (TFP batch semantics,
    tensorflow_probability/__init__.py:52-62: `normal(a * xs + b, sigma) @ "y"` with 500 observations): instead of one
    unrolled copy per element — registers and program size grow with the length — ONE counted loop in the site program:
    iteration j evaluates the parameters at element j (tracer.LazyVec: one table / step read each), draws with counter
    j from the ONE site key (sampler immediate GMX_ELEM_LOOP) or reads element j of the constraint /
    the previous value, and adds the element's log-density to a loop-carried sum in element order, which is the order
    `ExactDensity.estimate_logpdf` is summed in (distribution.py:383-396 as the oracle states it).  Returns None when
    the site is not of this shape (the unrolled form then applies)."""
    from .engine import StepInput, StepOutput
    from .numpy import RuntimeTable, TableArray
    from .program import ELEM_LOOP
    g, tr = ctx.tr.graph, ctx.tr
    if _VECTOR_SITE_LOOPS_OFF[0]:
        return None
    if len(g.loop_counts) >= 3 or getattr(g, "elem_from_index", False):
        return None
    custom = None        # a distribution with a loop body of its own (categorical's many draws at one site)
    if dist.logpdf_op is None or dist.sample_op is None:
        if not hasattr(dist, "loop_site") or len(g.loop_counts) >= 2:
            return None
    kind = req.kind if req is not None else "empty"
    regen = False
    if mode not in ("simulate", "generate", "assess") and kind == "regen":
        # Regenerate ON this site (distribution.py:258-300): new values from the prior — drawn in the loop, element j on
        # counter j of the ONE site key, as simulate does — the weight the new score minus the old; not selected: an empty
        # request (requests.py:56-57)
        if ctx.gate is not None:
            return None
        regen = bool(req.selection.check())
        kind = "empty"
    if mode not in ("simulate", "generate", "assess") and kind not in ("update", "empty"):
        return None
    cv = cval.value if isinstance(cval, Sym) else cval
    if regen:
        cv = None
    if isinstance(cv, Mask):
        return None
    if ctx.gate is not None and cv is not None and mode not in ("simulate", "generate", "assess"):
        # an edit under a gate (`IndexRequest(idx, Update(C["y"].set(v)))` around a loop-form scan / plate) that REPLACES
        # this site's values: the new values hold at the gated iteration only, the old ones elsewhere — a select per
        # element, which the unrolled form's _gate_site makes; the loop form would record the constraint at EVERY
        # iteration (weights summed over all steps, silently)
        return None
    pv = ps = None
    if mode not in ("simulate", "generate", "assess"):
        pv, ps = prev["value"].value, prev["score"].value
    if dist.logpdf_op is None or dist.sample_op is None:
        with T.tracing(g):
            custom = dist.loop_site(args)
        if custom is None:
            return None
        n = int(custom[0])
        for a in ([cv] if cv is not None else []) + ([pv] if pv is not None else []):
            if T._long_vector(a) != n:
                return None
        operands = []
    else:
        operands = list(args) + ([cv] if cv is not None else []) + ([pv] if pv is not None else [])
        n = T.lazy_length(operands)
        if n < VECTOR_SITE_LOOP_MIN or not any(T._long_vector(a) for a in operands):
            return None
    for a in operands:           # every vector operand must be readable at a run-time index
        if isinstance(a, (np.ndarray, list, tuple)) and np.ndim(a) > 0 and not T._long_vector(a):
            return None
    if cv is not None and not T._long_vector(cv):
        return None              # (a scalar constraint broadcast over a vector site: the unrolled form says what it means)
    if pv is not None and not T._long_vector(pv):
        return None
    new_value = mode in ("simulate",) or (mode == "generate" and cv is None) or regen
    if new_value and key is None:
        return None
    if mode not in ("simulate", "generate", "assess") and not regen:
        if cv is None and not ctx.args_changed(args):
            gv = prev["value"].value if isinstance(prev["value"], Sym) else prev["value"]
            so = gv.passthrough() if hasattr(gv, "passthrough") else None
            return _SiteRec(dist, so if so is not None else prev["value"], prev["score"]), pv, None, ps   # requests.py:56-57 / distribution.py:225-233
    zero = g.const_f32(0.0)
    svar = g.loop_var(zero)
    # HMC differentiates the model's score with respect to selected SCALAR choices (requests/hmc.py:69-97: jax.grad of
    # assess): the adjoint of this loop is a sum over its iterations too — d score / d w = sum_j d s_j / d w — accumulated
    # in the SAME loop, one loop-carried register per selected scalar, and handed to autodiff.grad as this loop variable's
    # derivative (Graph._custom_grads)
    wrt = [w for w in (getattr(ctx, "grad_wrt", None) or ()) if isinstance(w, Expr)]
    gvars = [g.loop_var(zero) for _ in wrt]
    # ... and with respect to selected LONG VECTORS (tracer.GradVec: `HMC(S["theta"])`): where this loop reads element j of
    # such a vector at its own iteration j — as the site's value or through its parameters — d s_j / d v_j is stored as
    # element j of one more output, the site's contribution to the vector's gradient
    gvecs = list(getattr(ctx, "grad_vecs", None) or ())
    marks = [len(gv.reads) for gv in gvecs]
    gmarks = [len(gv.gathers) for gv in gvecs]
    vec_origins, gather_origins = [], []
    g.loop_begin(n)
    with T.tracing(g):
        t = Expr(g.add("LDT", dtype="i32"))
        a_t = tuple(T.as_float(T._elem(a, t)) for a in args) if custom is None else ()
        if new_value:
            if custom is not None:
                x_t = custom[1](key, t)
            else:
                x_t = Expr(g.add(dist.sample_op, (key.node,) + tuple(a.node for a in a_t), imm=ELEM_LOOP, dtype=dist.value_dtype))
            origin = tr.store_step(x_t, n)
        else:
            x_t = dist._conv_value(T._elem(cv if cv is not None else pv, t))
        if custom is not None:
            s_t = custom[2](x_t, t)
        else:
            s_t = Expr(g.add(dist.logpdf_op, (x_t.node,) + tuple(a.node for a in a_t), dtype="f32"))
        updates = [(svar, (Expr(svar) + s_t).node)]
        used = []
        if wrt:
            from .autodiff import grad as _grad
            parts = _grad(s_t, wrt)
            for w_, gv, pt in zip(wrt, gvars, parts):
                if not (pt.node.op == "CONST" and pt.node.imm == 0):       # (this element does not depend on w)
                    updates.append((gv, (Expr(gv) + pt).node))
                    used.append((w_.node, gv))
        for gvec, mark in zip(gvecs, marks):
            if gvec.n != n:
                continue
            mine, seen = [], set()
            for k_, (i_, v_) in enumerate(gvec.reads[mark:]):
                if isinstance(i_, Expr) and i_.node is t.node:
                    gvec.consumed += 1
                    if v_.node.idx not in seen:
                        seen.add(v_.node.idx)
                        mine.append(v_)
            if mine:
                from .autodiff import grad as _grad
                parts = _grad(s_t, mine)
                tot = parts[0]
                for pt in parts[1:]:
                    tot = tot + pt
                vec_origins.append((gvec, tr.store_step(tot, n)))
        for gvec, gmark in zip(gvecs, gmarks):
            # GATHERED reads (`theta[group]`: element t of this site reads theta at group[t]): d s_t / d (that value), stored
            # per ITERATION; the sums per group follow the loop (_scatter_add)
            by_idx = {}
            for i_, j_, v_, idxv in gvec.gathers[gmark:]:
                if isinstance(i_, Expr) and i_.node is t.node:
                    gvec.gathers_consumed += 1
                    by_idx.setdefault(id(idxv), (idxv, []))[1].append(v_)
            for idxv, vals in by_idx.values():
                from .autodiff import grad as _grad
                uniq = list({v_.node.idx: v_ for v_ in vals}.values())
                parts = _grad(s_t, uniq)
                tot = parts[0]
                for pt in parts[1:]:
                    tot = tot + pt
                gather_origins.append((gvec, tr.store_step(tot, n), idxv))
        g.set_vars(updates)
    g.loop_end()
    for gvec, o_ in vec_origins:
        gvec.contribs.append((svar, tr.alias_step_input(o_, "f32", n)))
    for gvec, o_, idxv in gather_origins:
        gvec.contribs.append((svar, _scatter_add(tr, tr.alias_step_input(o_, "f32", n), idxv, n, gvec.n)))
    if used:
        g.__dict__.setdefault("_custom_grads", {})[svar.idx] = used
    score = Expr(svar)

    def given(v):           # a long row of a per-particle leaf that IS the site's value: recorded as that leaf (no copy)
        raw = v.value if isinstance(v, Sym) else v
        so = raw.passthrough() if hasattr(raw, "passthrough") else None
        return so if so is not None else v
    if new_value:
        v = StepOutput(origin, n, len(tr.outputs[origin[1]][1]), vector_site=True)
        # the MODEL gets the stored values as reads of that very output (engine.StepAlias; inside an enclosing loop a
        # recipe the next vector site's loop reads, engine.StepOutputAlias): a later vector site whose parameters they are
        # loops over them, `theta[3]` is one load — an 8-schools model of 5 000 schools
        ret = tr.alias_step_input(origin, dist.value_dtype, n)
        if regen:
            ctx.mark_changed(ret)
            return _SiteRec(dist, v, score, discard=given(prev["value"])), ret, score - ps, score     # (distribution.py:266-277)
        return _SiteRec(dist, v, score), ret, None, score   # (generate, unconstrained: w = 0, distribution.py:124-127)
    if mode == "generate":
        return _SiteRec(dist, given(cval), score), cv, score, score          # w = score = logpdf (:144-147)
    if mode == "assess":
        return _SiteRec(dist, given(cval), score), cv, None, score
    if cv is None:                                     # the old value re-scored under changed arguments
        return _SiteRec(dist, given(prev["value"]), score), pv, score - ps, score
    ctx.mark_changed(cv)
    return _SiteRec(dist, given(cval), score, discard=given(prev["value"])), cv, score - ps, score      # (:235-242)


def _scatter_add(tr, D, idxv, N, J):
    """G[j] = sum over i of D[i] where idxv[i] == j, i in element order — the adjoint of the gather `v[idxv]` (HMC on a long
    vector read at a table of group indices): two counted loops, J x N compares per particle and pass; every value is
    read at a register index, the sums land in one more [J, n] output of the launch, read back like any other."""
    g = tr.graph
    zero = g.const_f32(0.0)
    g.loop_begin(J)
    with T.tracing(g):
        tj = Expr(g.add("LDT", dtype="i32")) + 0           # (a register of its own: the inner loop's LDT is another node)
        acc = g.loop_var(zero)
        g.loop_begin(N)
        ti = Expr(g.add("LDT", dtype="i32"))
        ji = T.as_int(T._elem(idxv, ti))
        di = D._read_at(ti)
        g.set_vars([(acc, (Expr(acc) + T.where(ji == tj, di, 0.0)).node)])
        g.loop_end()
        o = tr.store_step(Expr(acc) + 0.0, J)
    g.loop_end()
    return tr.alias_step_input(o, "f32", J)


def _rec_choices(rec) -> ChoiceMap:
    if isinstance(rec, _SiteRec):
        v = rec.value
        return ChoiceMap.choice(v.value if isinstance(v, Sym) else v)
    if MASK_FLAG in rec.sites:          # a MaskCombinator call: the inner choices under its flag (mask.py:70)
        f = rec.sites[MASK_FLAG].value
        return _rec_choices(rec.sites[()]).mask(f.value if isinstance(f, Sym) else f)
    cm = ChoiceMap.empty()
    for a, r in rec.sites.items():
        cm = cm.set(a, _rec_choices(r))
    return cm


def _rec_score(rec):
    if isinstance(rec, _DeferredPlateRec):
        return 0.0                     # (added after the launch, by the caller: run_gfi)
    if getattr(rec, "plate_score", None) is not None:
        return rec.plate_score
    if isinstance(rec, _SiteRec):
        return rec.score.value if isinstance(rec.score, Sym) else rec.score
    if MASK_FLAG in rec.sites:          # a MaskCombinator call: flag * inner score (mask.py:71)
        f = rec.sites[MASK_FLAG].value
        return _masked_score(f.value if isinstance(f, Sym) else f, _rec_score(rec.sites[()]))
    acc = None
    for r in rec.sites.values():
        s = _rec_score(r)
        acc = s if acc is None else acc + s
    return acc if acc is not None else 0.0          # a callee without sites scores 0 (static.py:102-105: an empty sum)


def _gate_site(ctx, gate, out, prev, dist, args):
    """An edited leaf under a gate (`IndexRequest(idx, request)` around an element that runs a counted loop: the element
    is traced ONCE, with the request; where the gate does not hold a site does what carrying it over does — an empty
    Update: it keeps its value and is re-scored if its arguments changed (the step after an edited scan step sees a new
    carry: scan.py:325-416; the other elements of a plate see nothing new: vmap.py:277-332)."""
    rec, ret, w, s = out
    pv, ps = prev["value"].value, prev["score"].value
    nv = rec.value.value if isinstance(rec.value, Sym) else rec.value
    ns = rec.score.value if isinstance(rec.score, Sym) else rec.score
    from .engine import StepOutput
    if isinstance(nv, StepOutput) and not nv.vector_site and hasattr(pv, "passthrough") and rec.discard is None:
        # a long vector-valued site that KEPT its value (recorded by its origin: _vector_site_loop re-scored the old
        # value under the new arguments in its own loop): that IS the carried-over form — nothing to select between
        return rec, ret, w, ns
    if ctx.args_changed(dist.canon(args)):
        s_co = dist.sym_logpdf(pv, dist.canon(args))
        w_co = s_co - ps
    else:
        s_co, w_co = ps, None
    v = nv if nv is pv else T.where(gate, nv, pv)
    if isinstance(v, T.LazyVec):
        # (the select of two LONG vectors stays a recipe, and the previous values of a long vector site inside a loop are
        #  readable at the loop's own iteration number only: there is no storing this select element by element)
        raise NotImplementedError("an edit at one index (IndexRequest) that replaces the values of a vector-valued site of "
                                  f"more than 16 elements ({v.n}) inside a scan / plate run as a loop: edit the whole "
                                  "site with Update(C[:, addr].set(values)), or keep the site to 16 elements")
    sc = ns if ns is s_co else T.where(gate, ns, s_co)
    if w is None and w_co is None:
        wg = None
    else:
        wg = T.where(gate, w if w is not None else 0.0, w_co if w_co is not None else 0.0)
    ret = v if ret is nv else ret
    if nv is not pv:
        ctx.mark_changed(v)
    return _SiteRec(rec.gen_fn, v, sc, discard=rec.discard), ret, wg, sc


def call_gen_fn(ctx, mode, gen_fn, key, args, constraint, prev, req, req_leaves, addr):
    """Trace one callee; returns (record, retval, weight, score)."""
    from .core.generative import GenerativeFunctionClosure
    from .distributions import Distribution
    if isinstance(gen_fn, GenerativeFunctionClosure):
        gf, a = GenerativeFunctionClosure(gen_fn.gen_fn, gen_fn.args + tuple(args), gen_fn.kwargs)._target()
        return call_gen_fn(ctx, mode, gf, key, a, constraint, prev, req, req_leaves, addr)
    if isinstance(gen_fn, Distribution):
        if req is not None and req.kind == "static":      # StaticRequest({(): r}) at a leaf
            req = req.subs.get((), _ReqSpec("empty"))
            if req.kind == "update":
                constraint = req.constraint
        out = _leaf_call(ctx, mode, gen_fn, key, args, constraint, prev, req, req_leaves)
        if ctx.gate is not None and prev is not None and mode not in ("simulate", "generate", "assess"):
            out = _gate_site(ctx, ctx.gate, out, prev, gen_fn, args)
        _store_site(ctx, out[0])
        return out
    if isinstance(gen_fn, StaticGenerativeFunction):
        h = Handler(ctx, mode, key, constraint, prev, req, req_leaves)
        h.rec = _CallRec(gen_fn)
        _HANDLERS.append(h)
        try:
            with T.tracing(ctx.tr.graph):
                retval = gen_fn.source(*_sym_arrays(args))
        finally:
            _HANDLERS.pop()
        h.rec.retval = retval
        if mode == "simulate":
            return h.rec, retval, None, _rec_score(h.rec)
        if mode == "assess":
            return h.rec, retval, None, h.score
        return h.rec, retval, h.weight, None
    if hasattr(gen_fn, "trace_call"):
        return gen_fn.trace_call(ctx, mode, key, args, constraint, prev, req, req_leaves, addr)
    raise TypeError(f"cannot trace a callee of type {type(gen_fn).__name__}")


# ---------------------------------------------------------------------------
# launch driver
# ---------------------------------------------------------------------------
_CACHE = _new_program_cache()
_SITE_BY_SITE = ("site by site",)      # cache entry of a call signature that runs through sitewise.py


def _lib_jit_min() -> int:
    from . import engine
    return engine.JIT_MIN_PARTICLES


def _infer_batch(values) -> tuple:
    """Common leading shape of the device leaves (vector-valued leaves carry
    event axes after the particle axes); pass batch_shape= when ambiguous."""
    shape = None
    for v in values:
        if isinstance(v, Broadcast):            # launch-uniform whatever its shape: says nothing about the batch
            continue
        if isinstance(v, (torch.Tensor, Gathered)) and len(v.shape):
            s = tuple(v.shape)
            if shape is None:
                shape = s
            else:
                k = 0
                while k < min(len(s), len(shape)) and s[k] == shape[k]:
                    k += 1
                shape = shape[:k]
    return shape or ()


def _sym_constraint(tree, syms) -> ChoiceMap:
    return unflatten(tree, lambda j: syms[j])


def _emit_rec(tr: Tracing, rec):
    """Origins of every site (stores were emitted as the sites were traced;
    anything not stored yet is stored here); mirrors rec."""
    if isinstance(rec, _SiteRec):
        if rec.origins is None:
            rec.origins = (_origin_of(tr, rec.value), _origin_of(tr, rec.score), _origin_of(tr, rec.discard))
        vo, so, do = rec.origins
        if rec.escore is not None:
            return ("site", rec.gen_fn, vo, so, do, _origin_of(tr, rec.escore))
        return ("site", rec.gen_fn, vo, so, do)
    if isinstance(rec, _DeferredPlateRec):
        return ("deferred", rec.index)
    subs = OrderedDict((a, _emit_rec(tr, r)) for a, r in rec.sites.items())
    ps = getattr(rec, "plate_score", None)
    if ps is not None:
        return ("vmap", rec.gen_fn, subs, tr.emit_output(rec.retval), tr.emit_output(ps))
    return ("call", rec.gen_fn, subs, _emit_retval(tr, rec.retval))


def _emit_retval(tr, v):
    """emit_output of a return value that may BE a deferred plate's (or hold it in a tuple / dict)"""
    if isinstance(v, DeferredOutput):
        return ("deferred_ret", v.index)
    if isinstance(v, tuple) and any(isinstance(x, DeferredOutput) for x in v):
        return ("tuple", [_emit_retval(tr, x) for x in v])
    return tr.emit_output(v)


def _resolve_retval(origin, outs, leaves, deferred):
    if origin[0] == "deferred_ret":
        return deferred[origin[1]].get_retval()
    if origin[0] == "tuple" and deferred:
        return tuple(_resolve_retval(o, outs, leaves, deferred) for o in origin[1])
    return resolve(origin, outs, leaves)


def _build_trace(otree, outs, leaves, args, deferred=None):
    if otree[0] == "deferred":
        return deferred[otree[1]]
    if otree[0] == "site":
        gf, vo, so = otree[1], otree[2], otree[3]
        out = DistributionTrace(gf, args, resolve(vo, outs, leaves), resolve(so, outs, leaves))
        if len(otree) > 5 and otree[5] is not None:
            out._elem_scores = resolve(otree[5], outs, leaves)
        return out
    if otree[0] == "vmap":
        _, gf, subs, ro, po = otree
        st = OrderedDict((a, _build_trace(o, outs, leaves, None, deferred)) for a, o in subs.items())
        inner = (MaskTrace if MASK_FLAG in st else StaticTrace)(gf.gen_fn, None, resolve(ro, outs, leaves), st)
        return VmapTrace(gf, inner, resolve(po, outs, leaves), resolve(ro, outs, leaves), args)
    _, gf, subs, ro = otree
    st = OrderedDict((a, _build_trace(o, outs, leaves, None, deferred)) for a, o in subs.items())
    cls = MaskTrace if MASK_FLAG in st else StaticTrace
    return cls(gf, args, _resolve_retval(ro, outs, leaves, deferred) if deferred else resolve(ro, outs, leaves), st)


def _build_discard(otree, outs, leaves) -> ChoiceMap:
    if otree[0] == "site":
        do = otree[4]
        return ChoiceMap.choice(materialize(resolve(do, outs, leaves))) if do is not None else ChoiceMap.empty()
    cm = ChoiceMap.empty()
    for a, o in otree[2].items():
        d = _build_discard(o, outs, leaves)
        if not d.static_is_empty():
            cm = cm.set(a, d)
    if MASK_FLAG in otree[2]:           # mask.py:253: the inner discard under the NEW flag
        cm = cm.mask(_host_flag(materialize(resolve(otree[2][MASK_FLAG][2], outs, leaves))))
    return cm


def _broadcast_score(x, batch, device):
    if isinstance(x, torch.Tensor):
        return x
    return torch.full(batch, float(x), dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------
# noise ahead for the functional API (smc.capture): while a NoiseAheadContext is active, generate / simulate / MH
# launches over at least `min_particles` particles take the draws of their launch-keyed `normal` / `uniform` sites
# from memory (engine.NoiseHoist) and a BACKGROUND program draws them on the context's stream.  In a captured loop the
# noise launches depend on nothing but each other, so the graph lets them run ahead of the dependent chain
# [resample -> rejuvenate -> extend ...] and fill the issue slots its launch boundaries leave idle.
# ---------------------------------------------------------------------------
_NOISE_CTX = None


class NoiseAheadContext:
    """RECORD, then REPLAY AHEAD.  An eager pass of the loop under the context launches every background program where
    its draws are needed (and remembers the sequence: program, keys, buffer size).  For the capture, the whole recorded
    sequence is issued FIRST on the background stream into one arena — the draws depend on keys only — with an event
    after each group of launches (groups grow 2, 4, 8, ... up to `group`); the loop's launches then find their draws in
    the arena and the chain waits for an event only where a new group begins: cross-stream dependencies are the
    expensive part of a two-stream graph (one per launch made the captured loop SLOWER, 60.6 vs 43.7 us/step on
    config 3), and a block from the graph's pool must not be used (a noise launch that runs ahead would overwrite a
    recycled temporary of an earlier step)."""

    def __init__(self, stream, min_particles: int = 1 << 18, lds_pad: int = 56000, group: int = 16, kinds=None):
        self.stream, self.min_particles, self.lds_pad, self.group = stream, int(min_particles), int(lds_pad), int(group)
        # which launches hand their draws over: "mh" (rejuvenate: proposal + accept draws), "generate" / "simulate"
        # (extend, ImportanceK).  The two streams should carry about the same vector work: with an MH move per step,
        # hoisting everything makes the background stream the bottleneck (config 3 under smc.capture, us/step: one
        # stream 43.4, all 41.9, generate 40.6, mh 37.7).  Default "auto": the MH draws when the
        # recorded loop has MH moves, the generate / simulate draws otherwise.
        kinds = kinds if kinds is not None else "auto"
        self.kinds = tuple(kinds.split(",")) if isinstance(kinds, str) else tuple(kinds)
        self.plan, self.mode, self.cursor = [], "record", 0
        self.arena, self.views, self.events = None, [], {}

    def __enter__(self):
        global _NOISE_CTX
        self._prev, _NOISE_CTX = _NOISE_CTX, self
        self.cursor = 0
        if self.mode == "record":
            self.plan = []
        return self

    def __exit__(self, *exc):
        global _NOISE_CTX
        _NOISE_CTX = self._prev
        return False

    def applies(self, key, batch, kind: str = "generate") -> bool:
        if kind not in self.kinds and "all" not in self.kinds and "auto" not in self.kinds:
            return False
        return (key is not None and len(batch) == 1 and int(batch[0]) >= self.min_particles
                and getattr(key, "_lazy", None) is not None and key._lazy[0] == "split")

    @staticmethod
    def _sig(nprog, batch, key, n_draws):
        b = key.binding()
        return (id(nprog), tuple(batch), int(b[0]), int(b[1]), int(b[2]), int(getattr(key, "_offset", 0)), int(n_draws))

    @property
    def demand(self) -> int:
        return sum(e[3] * int(e[1][0]) for e in self.plan)

    def _schedule(self, device):
        """the recorded launches as groups (2, 4, 8, ... up to self.group launches, an event after each); consecutive
        launches of one program within a group become ONE launch over rows of keys (a 2-D grid, GMX_KEY_ROWSPLIT;
        gmx_program_run): fewer nodes for the runtime to walk when the graph is replayed.  Built BEFORE the capture
        (the rows' keys are uploaded here)."""
        from .random import Key as _Key
        groups, size, i = [], 2, 0
        while i < len(self.plan):
            last = min(len(self.plan), i + min(size, self.group))
            runs = []
            while i < last:
                nprog, batch, key, n_draws = self.plan[i][:4]
                j = i + 1
                while (j < last and self.plan[j][0] is nprog and self.plan[j][1] == batch and self.plan[j][3] == n_draws
                       and int(getattr(self.plan[j][2], "_offset", 0)) == 0 and int(getattr(key, "_offset", 0)) == 0):
                    j += 1
                rows, n = j - i, int(batch[0])
                rk = None
                if rows > 1 and rows * n < 2 ** 31 - 4096:
                    kd = np.stack([self.plan[i + r][2]._lazy[1].host() for r in range(rows)]).astype(np.uint32)
                    rk = _Key(lazy=("rowsplit", _Key(dev=torch.from_numpy(kd.view(np.int32)).to(device)), n), split_last=True)
                runs.append((i, j, rk))
                i = j
            groups.append(runs)
            size *= 2
        return groups

    def issue_ahead(self, device):
        """inside the capture, once the background stream has joined it: every recorded launch, now"""
        off, self.views, self.events = 0, [None] * len(self.plan), {}
        with torch.cuda.stream(self.stream):
            for runs in self.schedule:
                for i, j, rk in runs:
                    nprog, batch, key, n_draws = self.plan[i][:4]
                    rows, n = j - i, int(batch[0])
                    z = self.arena[off:off + n_draws * rows * n].view(n_draws, rows, n)
                    off += n_draws * rows * n
                    if rk is None:
                        for r in range(rows):
                            e = self.plan[i + r]
                            e[0].run(e[1], e[2], [z[k, r:r + 1] for k in range(n_draws)])
                    else:
                        nprog.run((rows * n,), rk, [z[k].reshape(1, rows * n) for k in range(n_draws)])
                    for r in range(rows):
                        self.views[i + r] = [z[k, r] for k in range(n_draws)]
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.events[runs[0][0]] = ev
        self.mode, self.cursor = "replay", 0

    def settle(self) -> bool:
        """after a recorded pass under "auto": fix the kinds; True when another eager pass is needed (so that every
        program variant the capture will launch exists before it starts)"""
        if "auto" not in self.kinds:
            return False
        self.kinds = ("mh",) if any(e[5] == "mh" for e in self.plan) else ("generate", "simulate")
        return True

    def reserve(self, device):
        """before the capture: one arena for every draw of the recorded loop, and the launch schedule"""
        self.schedule = self._schedule(device)
        self.arena = torch.empty((max(self.demand, 1),), dtype=torch.float32, device=device)

    def draw(self, nprog, batch, key, n_draws, kind: str = "generate"):
        """the [n] views of this launch's draws, ready on the current stream"""
        n = int(batch[0])
        cur = torch.cuda.current_stream()
        if self.mode == "replay":
            if self.cursor >= len(self.plan) or self.plan[self.cursor][4] != self._sig(nprog, batch, key, n_draws):
                raise _lib.GenmiError("noise-ahead: the captured loop differs from the eager pass that was recorded")
            ev = self.events.get(self.cursor)
            if ev is not None:
                cur.wait_event(ev)
            self.cursor += 1
            return self.views[self.cursor - 1]
        z = torch.empty((n_draws, n), dtype=torch.float32, device=cur.device)
        self.stream.wait_stream(cur)          # eager: the block may be a recycled one the current stream still reads
        z.record_stream(self.stream)
        self.plan.append((nprog, tuple(batch), key, int(n_draws), self._sig(nprog, batch, key, n_draws), kind))
        with torch.cuda.stream(self.stream):
            nprog.run(batch, key, [z[k:k + 1] for k in range(n_draws)])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        cur.wait_event(ev)
        return [z[k] for k in range(n_draws)]


def _noise_split(tr: Tracing, batch, ctx):
    """after tracing with a NoiseHoist: (draws, background program or None)"""
    hoist = tr.graph.__dict__.pop("noise_hoist", None)
    if hoist is None or not hoist.draws:
        return (), None
    q = NoiseProgram(hoist.draws, batch)
    if _lib.get().uses_streams and not torch.cuda.is_current_stream_capturing():
        q.comp.set_background(ctx.lds_pad)
        q.comp.specialize()
    return tuple(hoist.draws), q


@_unrolled_when_values_are_used
def run_gfi(gen_fn, mode, key: Key | None, args, constraint: ChoiceMap | None = None, batch_shape=None,
            weight_stats: bool = False, elem_index: bool = False):
    """simulate / generate / assess for any generative function: one launch.
    weight_stats (generate, 1-D batches): the program also reduces its importance weight per workgroup (OP_REDMAX) and,
    when it runs as a specialised 4-particles-per-thread kernel, leaves the resampler's CDF tile statistics
    (gmx_run_args.tile_agg_d): the returned weight tensor then carries them (`_gmx_tile_stats`) and
    smc.resample needs no pass of its own over the log-weights."""
    be = _lib.get()
    args = tuple(args)
    constraint = constraint if constraint is not None else ChoiceMap.empty()
    flat = Flat()
    atree = flat.add(args)
    ctree = flat.add(constraint)
    if key is not None:
        batch = tuple(key.shape)
        if elem_index:          # the elements of ONE vector-valued site on the launch axis: one key, counter = the index
            batch = tuple(batch_shape)
    elif batch_shape is not None:
        batch = tuple(batch_shape)
    else:
        batch = _infer_batch(flat.leaves)
    specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
    weight_stats = bool(weight_stats and mode == "generate" and len(batch) == 1)
    na = _NOISE_CTX if (_NOISE_CTX is not None and mode in ("generate", "simulate")
                        and _NOISE_CTX.applies(key, batch, mode)) else None
    # a large plate as the LAST site of a model called over a SMALL batch of particles runs after the program, over
    # particles x elements (combinators.Vmap._defer): the decision depends on the batch size, which is part of the key then
    from .combinators import DEFER_MAX_BATCH
    defer_B = int(batch[0]) if (isinstance(gen_fn, StaticGenerativeFunction) and len(batch) == 1 and not weight_stats
                                and na is None and not elem_index and 0 < int(batch[0]) <= DEFER_MAX_BATCH
                                and (key is None or tuple(key.shape) == tuple(batch))) else None
    ck = (_gfkey(gen_fn), mode, atree, ctree, specs, len(batch), key is not None, weight_stats, na is not None, elem_index,
          defer_B)
    ent = _CACHE.get(ck)
    if ent is _SITE_BY_SITE:
        return sitewise.run_gfi(gen_fn, mode, key, args, constraint)
    if ent is None and defer_B is not None and ck not in _NO_DEFER:
        try:
            ent = _trace_gfi(gen_fn, mode, key, batch, specs, atree, ctree, weight_stats, None, elem_index, defer_B)
            _CACHE[ck] = ent
        except DeferralAbort:
            _NO_DEFER.add(ck)
            ent = None
    first_error, attempt = None, 0
    while ent is None:
        # (second pass, when the program did not fit the launch slots / registers: top-level plates and scans as loops;
        #  third and fourth pass: the program CUT into a chain of launches — program.split_graph — as traced at first,
        #  then in the loop form)
        force, chain = attempt in (1, 3), attempt >= 2
        tr = Tracing(len(batch))
        tr.step_leaf_min = 0 if force else getattr(gen_fn, "step_leaf_min", tr.step_leaf_min)
        if na is not None:
            from .engine import NoiseHoist
            tr.graph.noise_hoist = NoiseHoist(tr, len(specs))
        tr.graph.elem_from_index = bool(elem_index)
        ctx = _Ctx(tr)
        ctx.store_sites = mode != "assess"
        # ONE trace of a `@gen` function: a large plate inside it sends the call to the site-by-site form (sitewise.py)
        ctx.sitewise = (len(batch) == 0 and isinstance(gen_fn, StaticGenerativeFunction) and not weight_stats and na is None
                        and (key is None or tuple(key.shape) == ()))
        try:
            from .combinators import forced_loops
            try:
                with T.tracing(tr.graph), forced_loops(force):
                    syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
                    sargs = unflatten(atree, lambda j: syms[j].value)
                    scon = _sym_constraint(ctree, syms)
                    kexpr = Expr(tr.graph.add("LDKEY", dtype="key")) if key is not None else None
                    rec, retval, w, s = call_gen_fn(ctx, mode, gen_fn, kexpr, sargs, scon, None, None, None, ())
            except sitewise.NeedsSiteBySite:
                _CACHE[ck] = _SITE_BY_SITE
                return sitewise.run_gfi(gen_fn, mode, key, args, constraint)
            with T.tracing(tr.graph):
                otree = _emit_rec(tr, rec) if mode != "assess" else None
                wo = tr.emit_output(w) if (mode == "generate" and w is not None) else None
                so = tr.emit_output(s) if mode == "assess" else None
                ro = tr.emit_output(retval) if mode == "assess" else None
                if weight_stats and isinstance(w, Expr) and w.node.op != "CONST":
                    tr.graph.add("REDMAX", (w.node,), dtype="none")
            ent = (Compiled(tr, chain=chain), otree, wo, so, ro) + (_noise_split(tr, batch, na) if na is not None else ((), None)) + ((),)
        except Exception as e:      # noqa: BLE001
            if attempt == 0 and not _over_the_slots(e):
                raise
            if attempt == 3 or (chain and (weight_stats or na is not None)):
                raise first_error from None        # no form fits (or applies): the first report stands
            first_error = first_error or e
            attempt += 1
            if attempt == 2 and (weight_stats or na is not None):
                raise first_error from None        # (a sweep's programs leave tile statistics: one launch or none)
            continue
        _CACHE[ck] = ent
    comp, otree, wo, so, ro, draws, nprog, plates = ent
    leaves = flat.leaves + (na.draw(nprog, batch, key, len(draws), mode) if nprog is not None else [])
    stats = None
    if weight_stats and comp.uses_red:
        n = int(batch[0])
        if n >= _lib_jit_min() and be.uses_streams and not torch.cuda.is_current_stream_capturing():
            comp.specialize()
        if comp.writes_tile_stats():
            from .inference.smc import cdf_shift
            partials = torch.empty((2, (n + 255) // 256), dtype=torch.float32, device=be.device)
            agg = torch.empty(((n + 1023) // 1024,), dtype=torch.int64, device=be.device)
            stats = (partials, agg, cdf_shift(n), n)
    if stats is not None:
        outs = comp.run(leaves, batch, key, red_out=stats[0], tile_stats=(stats[1], stats[2]))
    else:
        outs = comp.run(leaves, batch, key)
    deferred, w_def = None, None
    if plates:                         # the deferred plate: its arguments came out of the program; now its own launch
        from .engine import elementwise
        from .random import fold_in
        deferred = {}
        for d in plates:
            k = key
            for c in (d.key_path if key is not None else ()):
                k = fold_in(k, c)
            d.B = int(batch[0])
            a_c = tuple(resolve(o, outs, flat.leaves) for o in d.arg_origins)
            c_c = d.con_leaves.map_values(lambda r: flat.leaves[r[1]]) if d.con_leaves is not None else None
            res = d.gen_fn.run_deferred(d, k, a_c, c_c)
            if mode == "assess":
                deferred[d.index] = _AssessedPlate(res[1])
                w_def = res[0]
            else:
                deferred[d.index], w_def = res
    if mode == "assess":
        score = _broadcast_score(resolve(so, outs, flat.leaves), batch, be.device)
        if plates:
            score = elementwise(lambda a_, b_: a_ + b_, score, w_def)
            return score, _tree_materialize(_resolve_retval(ro, outs, flat.leaves, deferred))
        return score, _tree_materialize(resolve(ro, outs, flat.leaves))
    trc = _build_trace(otree, outs, flat.leaves, args, deferred)
    if mode == "simulate":
        return trc
    w = resolve(wo, outs, flat.leaves) if wo is not None else 0.0
    w = _broadcast_score(w, batch, be.device)
    if plates and w_def is not None:
        w = elementwise(lambda a_, b_: a_ + b_, w, w_def)
    if stats is not None and isinstance(w, torch.Tensor):
        # (block maxima, tile sums, shift, n, version): valid for exactly this tensor's values — an in-place change
        # of the weights afterwards (tempering, masking) bumps `_version` and smc.resample_fused drops the statistics
        w._gmx_tile_stats = tuple(stats) + (w._version,)
    return trc, w


_NO_DEFER: set = set()        # call signatures whose deferral was tried and abandoned (DeferralAbort)


class _AssessedPlate:
    def __init__(self, retval):
        self._r = retval

    def get_retval(self):
        return self._r


def _trace_gfi(gen_fn, mode, key, batch, specs, atree, ctree, weight_stats, na, elem_index, defer_B):
    """run_gfi's tracing step with plate deferral switched on; raises DeferralAbort when it does not apply"""
    tr = Tracing(len(batch))
    tr.step_leaf_min = getattr(gen_fn, "step_leaf_min", tr.step_leaf_min)
    tr.graph.elem_from_index = bool(elem_index)
    ctx = _Ctx(tr)
    ctx.store_sites = mode != "assess"
    ctx.defer = {"B": defer_B, "plates": [], "depth": len(_HANDLERS) + 1}
    with T.tracing(tr.graph):
        syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
        sargs = unflatten(atree, lambda j: syms[j].value)
        scon = _sym_constraint(ctree, syms)
        kexpr = Expr(tr.graph.add("LDKEY", dtype="key")) if key is not None else None
        rec, retval, w, s = call_gen_fn(ctx, mode, gen_fn, kexpr, sargs, scon, None, None, None, ())
        plates = ctx.defer["plates"]           # (none: this IS the ordinary trace)
        if plates and (not rec.sites or list(rec.sites.values())[-1] is not plates[0]):
            raise DeferralAbort()              # sites after the plate: their weights would be added in another order
        otree = _emit_rec(tr, rec) if mode != "assess" else None
        wo = tr.emit_output(w) if (mode == "generate" and w is not None) else None
        so = tr.emit_output(s if s is not None else 0.0) if mode == "assess" else None
        ro = _emit_retval(tr, retval) if mode == "assess" else None
    try:
        comp = Compiled(tr)
    except Exception as e:      # noqa: BLE001
        if not _over_the_slots(e):
            raise
        if not plates:
            raise DeferralAbort() from None      # (nothing deferred: run_gfi's ordinary passes deal with the size)
        comp = Compiled(tr, chain=True)          # the part in front of the deferred plate as a chain of launches
    return (comp, otree, wo, so, ro, (), None, tuple(plates))


class MinimalGenerate:
    """`generate` compiled to store ONLY the return value and the importance
    weight (plus per-block max partials of the weight for the resampler): the
    program BootstrapSweep launches once per SMC step.  Per particle-step this
    moves 4 B (ancestor) + 4*D B (gathered state) in and 4*D + 4 B out."""

    def __init__(self, gen_fn, args, constraint: ChoiceMap, batch: tuple, hoist_noise: bool = False):
        """hoist_noise: the program takes the standard-normal draws of its `normal` sites from memory (one extra
        f32 leaf per draw, `self.noise`: engine.NoiseHoist.draws; pass them to leaves(..., noise=[...])) — drawn by
        NoiseProgram(self.noise, batch) from the same keys, so every value is the one the plain program computes."""
        flat = Flat()
        self.atree = flat.add(tuple(args))
        self.ctree = flat.add(constraint)
        self.specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
        tr = Tracing(len(batch))
        hoist = None
        if hoist_noise:
            from .engine import NoiseHoist
            hoist = tr.graph.noise_hoist = NoiseHoist(tr, len(self.specs), hoist_noise)
        ctx = _Ctx(tr)
        ctx.store_sites = False
        with T.tracing(tr.graph):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(self.specs)]
            sargs = unflatten(self.atree, lambda j: syms[j].value)
            scon = _sym_constraint(self.ctree, syms)
            kexpr = Expr(tr.graph.add("LDKEY", dtype="key"))
            rec, retval, w, _ = call_gen_fn(ctx, "generate", gen_fn, kexpr, sargs, scon, None, None, None, ())
            self.ro = tr.emit_output(retval)
            if w is None:
                w = Expr(tr.graph.const_f32(0.0)) + 0.0
            self.wo = tr.emit_output(w)
            tr.graph.add("REDMAX", (w.node,), dtype="none")
        self.noise = tuple(hoist.draws) if hoist is not None else ()
        tr.graph.__dict__.pop("noise_hoist", None)
        self.comp = Compiled(tr)
        self.gen_fn = gen_fn

    def leaves(self, args, constraint, noise=()):
        flat = Flat()
        a = flat.add(tuple(args))
        c = flat.add(constraint)
        if a != self.atree or c != self.ctree:
            raise ValueError("MinimalGenerate: call structure differs from the compiled one")
        if len(noise) != len(self.noise):
            raise ValueError(f"MinimalGenerate: the program reads {len(self.noise)} noise leaves, got {len(noise)}")
        return flat.leaves + list(noise)


class NoiseProgram:
    """The background half of a noise-ahead step (engine.NoiseHoist): output k is the draw
    `jax.random.normal(fold_in(...fold_in(key_i, c_0)..., c_m))` (or `.uniform`) element e for
    draws[k] = (root, (c_0..c_m), e, kind), key_i the particle's launch key — exactly what the site with that key
    draws before its `* scale + loc` (S_NORMAL with loc 0, scale 1: z * 1 + 0 is z, bit for bit; S_UNIFORM(0, 1):
    0 + (1 - 0) * u is u).  Keys only: no inputs, nothing a chain produces.  All draws must share one root key."""

    def __init__(self, draws, batch: tuple):
        if not draws:
            raise ValueError("NoiseProgram: no draws")
        if len({d[0] for d in draws}) != 1:
            raise ValueError("NoiseProgram: the draws of one program come from one launch key")
        self.draws = tuple(draws)
        tr = Tracing(len(batch))
        g = tr.graph
        with T.tracing(g):
            root = g.add("LDKEY", dtype="key")
            self.outs = []
            for _, chain, e, kind in self.draws:
                k = root
                for c in chain:
                    k = g.add("KDERIVE", (k,), imm=c, dtype="key")
                z = g.add("S_NORMAL" if kind == "normal" else "S_UNIFORM", (k, g.const_f32(0.0), g.const_f32(1.0)),
                          imm=e, dtype="f32")
                self.outs.append(tr.emit_output(Expr(z)))
        self.comp = Compiled(tr)

    def run(self, batch: tuple, key, out_tensors):
        """out_tensors[k]: a [1, n] float tensor for draw k"""
        bufs = [None] * len(self.comp.outputs)
        for o, t in zip(self.outs, out_tensors):
            bufs[o[1]] = t
        self.comp.run([], batch, key, out_buffers=bufs)


def _rec_to_prev(rec):
    """A traced record as the symbolic PREVIOUS trace of an edit (values and scores stay expressions)."""
    def sym(v):
        return v if isinstance(v, Sym) else Sym(v, None)
    if isinstance(rec, _SiteRec):
        if rec.escore is not None:
            return {"value": sym(rec.value), "score": sym(rec.score), "escore": sym(rec.escore)}
        return {"value": sym(rec.value), "score": sym(rec.score)}
    from .core.mask import Mask
    ret = rec.retval
    if isinstance(ret, Mask):
        ret = {"mvalue": sym(ret.value), "mflag": sym(ret.flag)}
    else:
        ret = tuple(sym(r) for r in ret) if isinstance(ret, tuple) else sym(ret)
    if getattr(rec, "plate_score", None) is not None:
        return {"vmap": {"sub": {a: _rec_to_prev(r) for a, r in rec.sites.items()}, "retval": ret},
                "score": sym(rec.plate_score), "retval": ret}
    return {"sub": {a: _rec_to_prev(r) for a, r in rec.sites.items()}, "retval": ret}


def _trace_mh_move(tr, ctx, gen_fn, sargs, scon, rspec, syms):
    """Trace one MH move on the trace of `gen_fn(*sargs)` with choices `scon` into tr.graph (MinimalMH's program body):
    returns (the selected return value, the accept flag) as expressions."""
    g = tr.graph
    rec0, ret0, _, _ = call_gen_fn(ctx, "generate", gen_fn, None, sargs, scon, None, None, None, ())
    sprev = _rec_to_prev(rec0)
    kexpr = Expr(g.add("LDKEY", dtype="key"))
    k_acc = Expr(g.add("KDERIVE", (kexpr.node,), imm=1, dtype="key"))      # (k_edit, k_acc) = split(key_i)
    k_edit = Expr(g.add("KDERIVE", (kexpr.node,), imm=0, dtype="key"))
    mode, constraint = "static_edit", ChoiceMap.empty()
    if rspec.kind == "update":
        mode, constraint = "update", _sym_constraint(rspec.tree, syms)
    elif rspec.kind == "regen":
        mode = "regen"
    _bind_request_leaves(rspec, syms)
    rec, retval, w, _ = call_gen_fn(ctx, mode, gen_fn, k_edit, sargs, constraint, sprev, rspec, syms, ())
    from .distributions import uniform as _uniform
    if w is None:
        w = Expr(g.const_f32(0.0)) + 0.0
    u = _uniform.sym_sample(k_acc, (0.0, 1.0))
    acc = Expr(g.add("LOG", (u.node,), dtype="f32")) < w

    def sel(new, old):
        if isinstance(new, tuple):
            return tuple(sel(a, b) for a, b in zip(new, old))
        return T.where(acc, new, old)
    return sel(retval, ret0), acc


class MinimalMH:
    """One Metropolis-Hastings move per particle (`run_mh`'s program) on a trace that is never
    materialised: the inputs are the particle's arguments and choice VALUES (typically gathered
    through the ancestors of a resampling step); the old scores are recomputed from them — the same
    op sequence that produced the stored ones, so the same bits — and the only outputs are the
    selected return value and the accept flag.  BootstrapSweep launches it between resampling and
    the next extension."""

    def __init__(self, gen_fn, args, choices: ChoiceMap, request, batch: tuple):
        flat = Flat()
        self.atree = flat.add(tuple(args))
        self.ctree = flat.add(choices)
        rspec, self.rkey = _flatten_request(request, flat)
        self.n_fixed = len(flat.leaves)
        self.specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
        tr = Tracing(len(batch))
        ctx = _Ctx(tr)
        ctx.store_sites = False
        g = tr.graph
        with T.tracing(g):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(self.specs)]
            sargs = unflatten(self.atree, lambda j: syms[j].value)
            scon = _sym_constraint(self.ctree, syms)
            selected, acc = _trace_mh_move(tr, ctx, gen_fn, sargs, scon, rspec, syms)
            self.ro = tr.emit_output(selected)
            self.ao = tr.emit_output(acc)
        self.comp = Compiled(tr)

    def leaves(self, args, choices, request):
        flat = Flat()
        a = flat.add(tuple(args))
        c = flat.add(choices)
        _, rkey = _flatten_request(request, flat)
        if a != self.atree or c != self.ctree or rkey != self.rkey:
            raise ValueError("MinimalMH: call structure differs from the compiled one")
        return flat.leaves


class MinimalMHGenerate:
    """MinimalMH on `mh_fn`'s trace, THEN MinimalGenerate of `gen_fn` from the moved state — one program, one launch:
    what BootstrapSweep(rejuvenate=...) issues per step (the MH move on the resampled particles of step t-1, then the
    extension to step t).  The two calls keep their own keys: the move draws from the launch key (OP_LDKEY), the
    extension from split((k0, k1), n)[i] with (k0, k1) two launch values (OP_KSPLITU) — so every draw, weight and
    accept bit equals those of the two separate launches.  `gen_args(moved)` builds the extension's arguments from
    the moved state (an expression) and the launch values in `gen_extra`."""

    def __init__(self, mh_fn, mh_args, choices: ChoiceMap, request, gen_fn, gen_extra: tuple, gen_constraint: ChoiceMap,
                 batch: tuple, hoist_noise: bool = False):
        """hoist_noise: as MinimalGenerate — `self.noise` lists the draws taken from memory: root "LDKEY" ones (the
        move: proposal, accept) come from the launch key, root "KSPLITU" ones (the extension) from `key_words`.
        A tuple of roots instead of True hoists the draws of those keys only."""
        flat = Flat()
        self.atree = flat.add(tuple(mh_args))
        self.ctree = flat.add(choices)
        rspec, self.rkey = _flatten_request(request, flat)
        self.etree = flat.add(tuple(gen_extra))
        self.gtree = flat.add(gen_constraint)
        self.ktree = flat.add((0, 0))                 # the extension's launch key, two 32-bit words
        self.specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
        tr = Tracing(len(batch))
        hoist = None
        if hoist_noise:
            from .engine import NoiseHoist
            hoist = tr.graph.noise_hoist = NoiseHoist(tr, len(self.specs), hoist_noise)
        ctx = _Ctx(tr)
        ctx.store_sites = False
        g = tr.graph
        with T.tracing(g):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(self.specs)]
            sargs = unflatten(self.atree, lambda j: syms[j].value)
            scon = _sym_constraint(self.ctree, syms)
            moved, acc = _trace_mh_move(tr, ctx, mh_fn, sargs, scon, rspec, syms)
            self.mo = tr.emit_output(moved)
            self.ao = tr.emit_output(acc)
            extra = unflatten(self.etree, lambda j: syms[j].value)
            gcon = _sym_constraint(self.gtree, syms)
            kw = unflatten(self.ktree, lambda j: syms[j].value)
            k2 = Expr(g.add("KSPLITU", (kw[0].node, kw[1].node), dtype="key"))
            ctx2 = _Ctx(tr)
            ctx2.store_sites = False
            rec, retval, w, _ = call_gen_fn(ctx2, "generate", gen_fn, k2, (moved,) + tuple(extra), gcon, None, None, None, ())
            self.ro = tr.emit_output(retval)
            if w is None:
                w = Expr(g.const_f32(0.0)) + 0.0
            self.wo = tr.emit_output(w)
            g.add("REDMAX", (w.node,), dtype="none")
        self.noise = tuple(hoist.draws) if hoist is not None else ()
        g.__dict__.pop("noise_hoist", None)
        self.comp = Compiled(tr)

    def leaves(self, mh_args, choices, request, gen_extra, gen_constraint, key_words, noise=()):
        flat = Flat()
        a = flat.add(tuple(mh_args))
        c = flat.add(choices)
        _, rkey = _flatten_request(request, flat)
        e = flat.add(tuple(gen_extra))
        gc = flat.add(gen_constraint)
        flat.add((_as_i32(key_words[0]), _as_i32(key_words[1])))
        if a != self.atree or c != self.ctree or rkey != self.rkey or e != self.etree or gc != self.gtree:
            raise ValueError("MinimalMHGenerate: call structure differs from the compiled one")
        if len(noise) != len(self.noise):
            raise ValueError(f"MinimalMHGenerate: the program reads {len(self.noise)} noise leaves, got {len(noise)}")
        return flat.leaves + list(noise)


def _as_i32(u):
    """a 32-bit key word as the signed launch value with the same bits"""
    u = int(u) & 0xFFFFFFFF
    return u - (1 << 32) if u >= (1 << 31) else u


def _mh_select(tr: Tracing, acc: Expr, rec, prev):
    """Origins of where(accept, new, old) for every site; unchanged sites pass through."""
    def pick(new, old_sym):
        nv = new.value if isinstance(new, Sym) else new
        if isinstance(new, Sym) and new.origin is not None and new.origin == old_sym.origin:
            return old_sym.origin                      # untouched by the move
        return tr.emit_output(T.where(acc, nv, old_sym.value))
    if isinstance(rec, _SiteRec):
        return ("site", rec.gen_fn, pick(rec.value, prev["value"]), pick(rec.score, prev["score"]), None)
    if getattr(rec, "plate_score", None) is not None:
        inner = prev["vmap"]
        subs = OrderedDict((a, _mh_select(tr, acc, r, inner["sub"][a])) for a, r in rec.sites.items())
        ro = _tree_select(tr, acc, rec.retval, prev["retval"])
        po = tr.emit_output(T.where(acc, rec.plate_score, prev["score"].value))
        return ("vmap", rec.gen_fn, subs, ro, po)
    subs = OrderedDict((a, _mh_select(tr, acc, r, prev["sub"][a])) for a, r in rec.sites.items())
    old_ret = prev["retval"]
    ro = _tree_select(tr, acc, rec.retval, old_ret)
    return ("call", rec.gen_fn, subs, ro)


def _tree_select(tr, acc, new, old):
    if isinstance(new, (tuple, list)):
        seq = [_tree_select(tr, acc, n, o) for n, o in zip(new, old)]
        return ("tuple" if isinstance(new, tuple) else "list", seq)
    if new is None:
        return ("const", None)
    if isinstance(old, Sym):
        if isinstance(new, Expr) and tr.node_origin.get(id(new.node)) == old.origin and old.origin is not None:
            return old.origin
        return tr.emit_output(T.where(acc, new, old.value))
    return tr.emit_output(new)


def run_mh(gen_fn, key: Key, trace: Trace, request: EditRequest, argdiffs):
    """One Metropolis-Hastings move per particle as ONE launch: propose with
    `request.edit`, accept with log U < weight, select per particle — the fused
    form of the reference idiom (tests/inference/test_requests.py:131-137):

        new_tr, w, _, _ = request.edit(k_edit, tr, argdiffs)
        check = jnp.log(genjax.uniform.sample(k_acc, 0.0, 1.0)) < w
        tr = jtu.tree_map(lambda v1, v2: jnp.where(check, v1, v2), new_tr, tr)

    with (k_edit, k_acc) = split(key_i) for particle key key_i (build-defined key
    schedule).  Returns (selected trace, accept mask, weight)."""
    return run_edit(gen_fn, key, trace, request, argdiffs, mh=True)


@_unrolled_when_values_are_used
def run_edit(gen_fn, key: Key, trace: Trace, request: EditRequest, argdiffs, mh: bool = False):
    """edit(key, trace, request, argdiffs) -> (new trace, weight, retdiff, backward request)."""
    if getattr(trace, "_site_by_site", False) and not mh:
        return sitewise.run_edit(gen_fn, key, trace, request, argdiffs)     # ONE trace holding large plates
    be = _lib.get()
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    tangents = Diff.tree_tangent(argdiffs) if argdiffs is not None else ()
    flat = Flat()
    atree = flat.add(args)
    ptree = flat.add(_trace_tree(trace))
    rspec, rkey = _flatten_request(request, flat)
    batch = tuple(trace.batch_shape)
    if key is not None and tuple(key.shape) != batch:
        if key.shape == ():
            pass          # one key for a batched trace: every particle uses the same key
        else:
            raise ValueError(f"key batch {key.shape} does not match the trace batch {batch}")
    specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
    tkey = _tangent_key(tangents)
    na = _NOISE_CTX if (_NOISE_CTX is not None and mh and _NOISE_CTX.applies(key, batch, "mh")) else None
    ck = (_gfkey(gen_fn), "mh" if mh else "edit", atree, ptree, rkey, specs, tkey, len(batch), key is not None,
          na is not None)
    ent = _CACHE.get(ck)
    if ent is _MH_UNFUSED:
        return _run_mh_unfused(gen_fn, key, trace, request, argdiffs)
    first_error, attempt = None, 0
    while ent is None:
        force, chain = attempt in (1, 3), attempt >= 2        # (the passes of run_gfi)
        try:
            ent = _trace_edit(gen_fn, key, trace, request, argdiffs, mh, na, specs, batch, atree, ptree, rspec, tangents, ck,
                              force, chain)
        except _Unfused:
            return _run_mh_unfused(gen_fn, key, trace, request, argdiffs)
        except Exception as e:      # noqa: BLE001
            if attempt == 0 and not _over_the_slots(e):
                raise
            if attempt == 3 or (attempt >= 1 and na is not None):
                raise first_error from None
            first_error = first_error or e
            attempt += 1
            continue
        _CACHE[ck] = ent
    comp, otree, wo, ao, draws, nprog, ret_changed = ent
    outs = comp.run(flat.leaves + (na.draw(nprog, batch, key, len(draws), "mh") if nprog is not None else []), batch, key)
    return _finish_edit(request, mh, otree, wo, ao, outs, flat, args, batch, be, ret_changed)


class _Unfused(Exception):
    pass


def _over_the_slots(e) -> bool:
    from .program import ProgramTooLarge
    return isinstance(e, ProgramTooLarge) or (isinstance(e, ValueError) and "exceeds the ABI slot limits" in str(e))


def _trace_edit(gen_fn, key, trace, request, argdiffs, mh, na, specs, batch, atree, ptree, rspec, tangents, ck, force,
                chain=False):
    """run_edit's tracing step -> the cache entry"""
    if True:
        from .combinators import forced_loops
        tr = Tracing(len(batch))
        if na is not None:
            from .engine import NoiseHoist
            tr.graph.noise_hoist = NoiseHoist(tr, len(specs))
        ctx = _Ctx(tr)
        ctx.store_sites = not mh
        tr.step_leaf_min = 0 if force else getattr(gen_fn, "step_leaf_min", tr.step_leaf_min)
        with T.tracing(tr.graph), forced_loops(force):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
            sargs = unflatten(atree, lambda j: syms[j].value)
            # arguments flagged as changed seed the change set
            _seed_changed(ctx, sargs, tangents)
            sprev = unflatten(ptree, lambda j: syms[j])
            kexpr = Expr(tr.graph.add("LDKEY", dtype="key")) if key is not None else None
            k_acc = None
            if mh:          # (k_edit, k_acc) = split(particle key)
                k_acc = Expr(tr.graph.add("KDERIVE", (kexpr.node,), imm=1, dtype="key"))
                kexpr = Expr(tr.graph.add("KDERIVE", (kexpr.node,), imm=0, dtype="key"))
            mode, constraint = "static_edit", ChoiceMap.empty()
            req = rspec
            if rspec.kind == "update":
                mode = "update"
                constraint = _sym_constraint(rspec.tree, syms)
            elif rspec.kind == "regen":
                mode = "regen"
            _bind_request_leaves(rspec, syms)
            rec, retval, w, _ = call_gen_fn(ctx, mode, gen_fn, kexpr, sargs, constraint, sprev, req, syms, ())
            ao = None
            if mh and any(nd.op == "LOOP" for nd in tr.graph.nodes):
                # sites written INSIDE a counted loop (a long scan, a large plate) are in memory before the move is
                # accepted or refused: no selecting in registers — the move runs unfused (edit, accept, select)
                _CACHE[ck] = _MH_UNFUSED
                raise _Unfused()
            if mh:
                from .distributions import uniform as _uniform
                if w is None:
                    w = Expr(tr.graph.const_f32(0.0)) + 0.0
                u = _uniform.sym_sample(k_acc, (0.0, 1.0))
                acc = Expr(tr.graph.add("LOG", (u.node,), dtype="f32")) < w
                otree = _mh_select(tr, acc, rec, sprev)
                ao = tr.emit_output(acc)
            else:
                otree = _emit_rec(tr, rec)
            wo = tr.emit_output(w) if w is not None else None
            # the retdiff (static.py:948-981 returns the incremental interpreter's): NoChange only when the return value is
            # made of program values none of which depends on anything this edit changed
            ret_changed = True
            if not mh:
                flat_ret = []

                def walk(v):
                    if isinstance(v, Sym):
                        v = v.value
                    if v is None or isinstance(v, (bool, int, float, np.number)):
                        return True
                    if isinstance(v, Expr) or (isinstance(v, np.ndarray) and v.dtype == object):
                        flat_ret.append(v)
                        return True
                    if isinstance(v, (tuple, list)):
                        return all(walk(x) for x in v)
                    return False              # (stacked loop outputs, masks, pytrees: conservatively "changed")
                if walk(retval):
                    ret_changed = ctx.args_changed(flat_ret)
        return (Compiled(tr, chain=chain), otree, wo, ao) + (_noise_split(tr, batch, na) if na is not None else ((), None)) \
            + (ret_changed,)


def _finish_edit(request, mh, otree, wo, ao, outs, flat, args, batch, be, ret_changed=True):
    new_tr = _build_trace(otree, outs, flat.leaves, args)
    w = resolve(wo, outs, flat.leaves) if wo is not None else 0.0
    w = _broadcast_score(w, batch, be.device)
    if mh:
        return new_tr, resolve(ao, outs, flat.leaves), w
    discard = _build_discard(otree, outs, flat.leaves)
    if isinstance(request, Update):
        bwd = Update(discard)
    elif isinstance(request, Regenerate):
        bwd = Update(discard)
    elif isinstance(request, IndexRequest):
        bwd = IndexRequest(request.idx, Update(discard))       # for a per-particle idx the discard holds the kept
                                                               # values at the other indices
    elif isinstance(request, StaticRequest):
        bwd = _static_bwd(request, otree, outs, flat.leaves)
    else:
        bwd = request if isinstance(request, Rejuvenate) else Update(discard)
    retdiff = Diff.unknown_change(new_tr.get_retval()) if ret_changed else Diff.no_change(new_tr.get_retval())
    return new_tr, w, retdiff, bwd


_MH_UNFUSED = ("mh, unfused",)      # cache entry of an MH call signature whose edit holds counted loops


def _mh_accept(u, w):
    from . import numpy as jnp
    return jnp.log(u) < w


def _run_mh_unfused(gen_fn, key, trace, request, argdiffs):
    """run_mh as three launches — the edit with k_edit, the accept test log U(k_acc) < w, the per-particle select of
    every leaf — with the same keys ((k_edit, k_acc) = split(particle key)) and the same arithmetic as the fused form"""
    from .combinators import _trace_leaf_zip
    from .distributions import uniform as _uniform
    from .engine import elementwise
    from .random import split
    ks = split(key)
    k_edit, k_acc = ks[0], ks[1]
    new_tr, w, _, _ = run_edit(gen_fn, k_edit, trace, request, argdiffs)
    batch = tuple(trace.batch_shape)
    u = run_gfi(_uniform, "simulate", k_acc, (0.0, 1.0)).get_retval()
    w = _broadcast_score(w, batch, _lib.get().device)
    acc = elementwise(_mh_accept, u, w)

    def pick(old, new):
        old, new = materialize(old), materialize(new)
        if not isinstance(new, torch.Tensor):
            return new
        if not isinstance(old, torch.Tensor):
            old = torch.as_tensor(old, device=new.device).to(new.dtype).expand(new.shape)
        a = acc.reshape(tuple(acc.shape) + (1,) * (new.dim() - acc.dim()))
        return torch.where(a, new, old.to(new.dtype).expand(new.shape))
    args = tuple(Diff.tree_primal(argdiffs)) if argdiffs is not None else tuple(trace.get_args() or ())
    return _trace_leaf_zip(trace, new_tr, pick, args=args), acc, w


def _static_bwd(request, otree, outs, leaves):
    """The backward request of a StaticRequest (static.py:867-904: a StaticRequest of the sub-requests' backward
    requests): Update -> Update(discarded values) (distribution.py:235-242), Regenerate -> Update(old values)
    (:266-277), Rejuvenate -> itself (rejuvenate.py:89-94), a nested StaticRequest -> recursively."""
    if otree[0] == "site":
        sub = request.addressed.get(())
        subs = {(): sub} if sub is not None else {}
        trees = {(): otree}
    else:
        subs = {(_norm(a) if a != () else ()): r for a, r in request.addressed.items()}
        trees = {(_norm(a) if a != () else ()): (a, o) for a, o in otree[2].items()}
    out = {}
    for na, r in subs.items():
        if na not in trees:
            continue
        a, o = trees[na] if otree[0] != "site" else ((), otree)
        if isinstance(r, StaticRequest):
            out[a] = _static_bwd(r, o, outs, leaves)
        elif isinstance(r, Rejuvenate):
            out[a] = r
        elif isinstance(r, (Update, Regenerate)):
            out[a] = Update(_build_discard(o, outs, leaves))
        else:
            out[a] = r
    return StaticRequest(out)


def _tangent_key(tangents):
    if isinstance(tangents, (tuple, list)):
        return tuple(_tangent_key(t) for t in tangents)
    return tangents is NoChange


def _seed_changed(ctx, sargs, tangents):
    if isinstance(sargs, (tuple, list)) and isinstance(tangents, (tuple, list)) and len(sargs) == len(tangents):
        for a, t in zip(sargs, tangents):
            _seed_changed(ctx, a, t)
        return
    if tangents is not NoChange:
        ctx.mark_changed(sargs)


def _bind_request_leaves(rspec, syms):
    """Resolve Update constraints nested inside StaticRequests to symbolic choice maps."""
    if rspec.kind == "static":
        for s in rspec.subs.values():
            _bind_request_leaves(s, syms)
    elif rspec.kind == "index":
        if rspec.idx is None or isinstance(rspec.idx, Expr):
            rspec.idx = T.as_int(unflatten(rspec.idx_tree, lambda j: syms[j].value))
        _bind_request_leaves(rspec.sub, syms)
    elif rspec.kind == "update":
        rspec.constraint = _sym_constraint(rspec.tree, syms)


def _has_scalar_tensor_choice(sample) -> bool:
    flat = Flat()
    flat.add(sample)
    return any(isinstance(materialize(v), torch.Tensor) and materialize(v).dim() == 0 for v in flat.leaves
               if isinstance(v, (torch.Tensor, Gathered)))


# ---------------------------------------------------------------------------
# the generative function
# ---------------------------------------------------------------------------
class StaticGenerativeFunction(GenerativeFunction):
    """static.py:725-1036"""

    def __init__(self, source, partial_args=()):
        self._fn = source
        self._partial = tuple(partial_args)
        functools.update_wrapper(self, source, updated=())

    def source(self, *args, **kwargs):
        return self._fn(*self._partial, *args, **kwargs)

    def __get__(self, instance, _klass):
        return self.partial_apply(instance) if instance is not None else self

    def handle_kwargs(self):
        """A model taking ((args...), {kwargs}) (generative_function.py `handle_kwargs`); the same object every time,
        so that programs traced through it are cached and `gf.handle_kwargs() == gf.handle_kwargs()`."""
        kw = self.__dict__.get("_kwarged")
        if kw is None:
            fn = self

            def kwarged(args, kwargs):
                return fn.source(*args, **kwargs)
            kw = self.__dict__["_kwarged"] = StaticGenerativeFunction(kwarged)
            kw.__dict__["_kwarged"] = kw          # idempotent
        return kw

    def get_zero_trace(self, *args, **_kwargs):
        """A trace of the structure `simulate(key, args)` returns, every array leaf zero (generative_function.py
        `get_zero_trace`: the reference evaluates `simulate` abstractly; here the model runs once with a dummy key and
        its leaves are zeroed)."""
        from .random import key as _key
        tr = self.simulate(_key(0), tuple(args))
        return _zero_like_trace(tr)

    # GFI ---------------------------------------------------------------------------
    def simulate(self, key, args):
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        if batch_shape is None and _has_scalar_tensor_choice(sample):
            # assess has no key to tell particles from plate elements; a 0-d TENSOR among the choices (what an un-batched
            # trace's choices are) says the sample is ONE trace, whatever the plates' lengths suggest
            batch_shape = ()
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)

    def edit(self, key, trace, edit_request, argdiffs):
        assert isinstance(trace, StaticTrace)
        if isinstance(edit_request, (Update, StaticRequest, Regenerate)):
            return run_edit(self, key, trace, edit_request, argdiffs)
        raise NotSupportedEditRequest(edit_request)

    def project(self, key, trace, selection: Selection):
        """static.py:812-825: sum of the selected sub-trace scores."""
        weight = None
        for addr, st in trace.subtraces.items():
            w = st.get_gen_fn().project(key, st, selection(addr))
            weight = w if weight is None else weight + w
        return weight if weight is not None else 0.0

    def inline(self, *args):
        return self.source(*args)

    @property
    def partial_args(self):
        return self._partial

    def partial_apply(self, *args):
        return StaticGenerativeFunction(self._fn, self._partial + tuple(args))


def gen(f) -> StaticGenerativeFunction:
    """`@gen` (static.py:1044-1049)."""
    return StaticGenerativeFunction(f)
