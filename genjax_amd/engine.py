"""Launch engine: turns (values, choice maps, traces) into bound site-program
launches and back.

  flatten   a pytree of launch values into leaves + a hashable signature;
  Tracing   (trace time, once per signature) leaves -> symbolic values, while
            recording how each program input / uniform is bound to a leaf;
  Compiled  (launch time) binds leaves, allocates outputs, calls
            gmx_program_run, hands tensors back by "origin".

Layout in HBM: struct-of-arrays.  A per-particle scalar leaf is one contiguous
[N] vector; a per-particle value with event shape E is E such vectors ([E, N]
row-major, exposed to the user as an [N, E] strided view), so every load and
store in the kernel is one coalesced 256-byte wave access.
"""
from __future__ import annotations

import dataclasses
import hashlib

import os
import struct
import weakref
from collections import OrderedDict
from ctypes import POINTER, c_uint32, c_void_p, cast

import numpy as np
import torch

from . import _lib
from .core.choice_map import ChoiceMap
from .core.mask import Indexed, Mask
from .program import F_BCAST, F_GATHER, Graph, ProgramTooLarge, compile_graph, split_graph
from .random import Key
from .tracer import Expr, sym_array


# ---------------------------------------------------------------------------
# launch values
# ---------------------------------------------------------------------------
class Gathered:
    """`source[ancestors]` kept lazy so the consuming kernel fuses the gather
    (row = ancestors[i]) instead of materialising resampled particles."""

    def __init__(self, source: torch.Tensor, ancestors: torch.Tensor):
        self.source, self.ancestors = source, ancestors

    @property
    def shape(self):
        return tuple(self.ancestors.shape) + tuple(self.source.shape[1:])

    @property
    def dtype(self):
        return self.source.dtype

    def materialize(self) -> torch.Tensor:
        m = self.__dict__.get("_mat")
        if m is None:
            m = self.__dict__["_mat"] = gather_leaves([self.source], self.ancestors)[0]
        return m


def materialize_together(values):
    """the lazy gathers among `values` that share their ancestors, materialised by ONE gmx_gather launch per group (the
    ancestors are read once: a resampled trace's choices — ten latents of the 8-schools model — were one launch per
    site); results are kept by each `Gathered`"""
    groups = {}
    for v in values:
        if isinstance(v, Gathered) and v.__dict__.get("_mat") is None:
            groups.setdefault(id(v.ancestors), []).append(v)
    for members in groups.values():
        if len(members) < 2:
            continue
        outs = gather_leaves([m_.source for m_ in members], members[0].ancestors)
        for m_, o in zip(members, outs):
            m_.__dict__["_mat"] = o


class Patched:
    """`base` ([B, n, *event]: a plate trace's leaf) with element idx of every particle replaced by `rows` ([B, *event]),
    kept LAZY: an IndexRequest on a long plate edits ONE element (vmap.py:277-332 slices it with dynamic_slice and
    writes it back with dynamic_update_slice), and the new trace shares every other element with the old one instead
    of copying n of them per leaf.  `idx`: a Python int or one index per particle ([B] integer tensor).  A consumer
    that needs the whole leaf materialises it (one clone + one row store, cached); the next IndexRequest reads its
    element through the patch."""

    def __init__(self, base, idx, rows):
        self.base, self.idx, self.rows = base, idx, rows
        self._full = None
        self.depth = base.depth + 1 if isinstance(base, Patched) else 1

    @property
    def shape(self):
        return tuple(self.base.shape)

    @property
    def dtype(self):
        return self.base.dtype

    @property
    def device(self):
        return self.rows.device

    @property
    def ndim(self):
        return len(self.base.shape)

    def take(self, j):
        """element j of every particle ([B, *event]); j: int or [B] integer tensor"""
        if self._full is not None:
            return _take_plate(self._full, j)
        old = self.base.take(j) if isinstance(self.base, Patched) else _take_plate(self.base, j)
        if isinstance(j, int) and isinstance(self.idx, int):
            return self.rows if j == self.idx else old
        ji = j if isinstance(j, torch.Tensor) else torch.full_like(self.idx, int(j))
        hit = (ji.to(torch.int64) == (self.idx.to(torch.int64) if isinstance(self.idx, torch.Tensor) else int(self.idx)))
        return torch.where(hit.reshape(hit.shape + (1,) * (self.rows.ndim - 1)), self.rows, old)

    def materialize(self) -> torch.Tensor:
        if self._full is None:          # ONE clone of the oldest whole leaf, then the chain's patches oldest first
            chain, node = [], self
            while isinstance(node, Patched) and node._full is None:
                chain.append(node)
                node = node.base
            out = (node._full if isinstance(node, Patched) else materialize(node)).clone()
            if out.device != self.rows.device:
                raise ValueError("Patched: the leaf and its new rows live on different devices")
            rows_all = torch.arange(out.shape[0], device=out.device)
            for p in reversed(chain):
                if isinstance(p.idx, int):
                    out[:, p.idx] = p.rows
                else:
                    out[rows_all, p.idx.to(torch.int64)] = p.rows
            self._full = out
        return self._full


class Deferred:
    """a device tensor computed when somebody needs it (a plate's per-element scores: only a score read needs them)"""

    def __init__(self, fn, shape, dtype=torch.float32):
        self._fn, self.shape, self.dtype, self._v = fn, tuple(shape), dtype, None

    def materialize(self) -> torch.Tensor:
        if self._v is None:
            self._v, self._fn = self._fn(), None
        return self._v


def _take_plate(t, j):
    t = materialize(t)
    if isinstance(j, int):
        return t[:, j]
    return t[torch.arange(t.shape[0], device=t.device), j.to(torch.int64)]


class Broadcast(torch.Tensor):
    """A tensor marked LAUNCH-UNIFORM whatever its shape: `vmap(f, in_axes=(0, None))` marks its un-mapped tensor
    arguments with this, so a vector whose length happens to equal the particle count is still one vector shared by
    all particles (without the marker, batching is inferred from the leading shape).  It IS a tensor (a subclass), so
    the mapped function may compute with it before handing it to a generative function (`w * 2.0`, `w + 1.0`): the
    result of an operation whose tensor operands are ALL launch-uniform stays marked, anything mixed with an
    ordinary (per-instance) tensor is an ordinary tensor."""

    @staticmethod
    def __new__(cls, t):
        return t if isinstance(t, Broadcast) else t.as_subclass(cls)

    def __init__(self, t):
        pass

    @property
    def plain(self) -> torch.Tensor:
        return self.as_subclass(torch.Tensor)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        shared = [True]

        def look(v):
            if isinstance(v, torch.Tensor):
                if not isinstance(v, Broadcast):
                    shared[0] = False
            elif isinstance(v, (tuple, list)):
                for x in v:
                    look(x)
        look(args)
        look(tuple(kwargs.values()))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if not shared[0] or func in torch.overrides.get_default_nowrap_functions():
            return out

        def mark(v):
            if isinstance(v, torch.Tensor) and not isinstance(v, Broadcast):
                return v.as_subclass(cls)
            if isinstance(v, tuple):
                return tuple(mark(x) for x in v)
            if isinstance(v, list):
                return [mark(x) for x in v]
            return v
        return mark(out)


class Mapped(torch.Tensor):
    """A tensor `genjax.vmap` maps over its leading axis (one entry per instance).  Only a tag: an integer tensor
    carrying it, used as an address component (`C["ys", idx, "y"].set(v)`), is a RUN-TIME per-instance index
    (core.choice_map.DynamicIndex) rather than a static array of indices."""
    _gmx_mapped = True

    @staticmethod
    def __new__(cls, t):
        return t if isinstance(t, Mapped) else t.as_subclass(cls)

    def __init__(self, t):
        pass

    @property
    def plain(self) -> torch.Tensor:
        return self.as_subclass(torch.Tensor)


class Expanded(torch.Tensor):
    """What a BATCHED trace shows for a choice whose value was launch-uniform (a constraint given once for all particles:
    `C["v"].set(1)` under `vmap(model.importance)`): the value broadcast over the batch — a stride-0 view — as the reference's
    vmapped traces carry it (every leaf has the batch axis: `tree_map(lambda v: v[idx], traces.get_choices())`,
    importance_sampling.ipynb c8), with the launch-uniform value it stands for in `.base`: handed back to a generative
    function (ChangeTarget, `update(tr.get_choices())`) it is classified as that value again — no per-particle copy.
    Any torch operation on it yields a plain tensor."""

    @staticmethod
    def __new__(cls, t, base):
        out = t.as_subclass(cls)
        out.base = base
        return out

    def __init__(self, t, base):
        pass

    @property
    def plain(self) -> torch.Tensor:
        return self.as_subclass(torch.Tensor)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


_UNIFORM_DEV = {}


def _uniform_on_device(a: np.ndarray) -> torch.Tensor:
    """a small launch-uniform host value on the device, copied ONCE per content: `get_choices()` of a batched trace shows
    its launch-uniform constraints (the observations) with the batch axis, and a host-to-device copy from pageable memory
    waits for the stream — four of them per run were 0.1 ms of BASELINE config 4's 0.87 (round 6)"""
    dev = _lib.get().device
    if a.size > 4096:
        return torch.from_numpy(a).to(dev)
    k = (str(dev), a.dtype.str, a.shape, a.tobytes())
    t = _UNIFORM_DEV.get(k)
    if t is None:
        if len(_UNIFORM_DEV) >= 512:
            _UNIFORM_DEV.clear()
        t = _UNIFORM_DEV[k] = torch.from_numpy(a.copy()).to(dev)
    return t


def expand_over_batch(v, batch: tuple):
    """a launch-uniform value as a batched trace's leaf (Expanded), or v itself when it carries the batch already"""
    if not batch or v is None or isinstance(v, (Expanded, Gathered, Patched, PlateScore, Deferred)):
        return v
    if isinstance(v, torch.Tensor):
        if tuple(v.shape[:len(batch)]) == tuple(batch):
            return v
        t = v.plain if isinstance(v, (Broadcast, Mapped)) else v
        if t.device != _lib.get().device:
            return v
    elif isinstance(v, (bool, int, float, np.bool_, np.integer, np.floating)):
        dt = torch.bool if isinstance(v, (bool, np.bool_)) else (torch.int32 if isinstance(v, (int, np.integer)) else torch.float32)
        t = _uniform_on_device(np.asarray(v, dtype={torch.bool: np.bool_, torch.int32: np.int32, torch.float32: np.float32}[dt]))
    elif isinstance(v, np.ndarray) and v.dtype != object:
        a = np.ascontiguousarray(v)
        a = a.astype(np.float32) if a.dtype.kind == "f" else (a.astype(np.int32) if a.dtype.kind in "iu" else a)
        t = _uniform_on_device(a)
    else:
        return v
    # (inside a plate / scan the batch of a site's trace holds the plate axes too, and a launch-uniform table given for the
    #  plate carries them already: only the LEADING axes that are missing are added)
    nb = len(batch)
    m = min(nb, t.ndim)
    while m > 0 and tuple(batch[nb - m:]) != tuple(t.shape[:m]):
        m -= 1
    lead = tuple(batch[:nb - m])
    return Expanded(t.expand(lead + tuple(t.shape)) if lead else t.clone(), v)      # (never the cached copy itself)


def materialize(v):
    if isinstance(v, Expanded):
        return v.plain
    if isinstance(v, Mapped):
        return v.plain
    if isinstance(v, Broadcast):
        return v.plain
    return v.materialize() if isinstance(v, (Gathered, Patched, PlateScore, Deferred)) else v


_TDT = {torch.float32: "f32", torch.float64: "f32", torch.float16: "f32", torch.bfloat16: "f32",
        torch.int32: "i32", torch.int64: "i32", torch.int16: "i32", torch.int8: "i32",
        torch.uint8: "i32", torch.bool: "bool"}
_STORE = {"f32": torch.float32, "i32": torch.int32, "bool": torch.bool}


def _np_dt(a: np.ndarray) -> str:
    return "f32" if a.dtype.kind == "f" else ("bool" if a.dtype.kind == "b" else "i32")


DVEC_MAX = 16       # longer launch-uniform device vectors are bound as tables, not as one input slot per element
GMX_HVEC_MAX = 16   # launch-uniform HOST vectors: one operand-pool entry per element up to this length, a table beyond
POOL_VEC_MAX = 48   # a launch-uniform ARRAY (2 or more axes) with more elements than this is a table whatever its shape
STEP_LEAF_MIN = 16  # per-particle vectors longer than this are ONE step-indexed leaf (read inside a counted loop)


def leaf_spec(v, batch: tuple):
    """Classify one launch value (see module docstring)."""
    nb = len(batch)
    if isinstance(v, Expanded):
        v = v.base           # (a launch-uniform value a batched trace showed broadcast: classified as what it stands for)
    if v is None:
        return ("none",)
    if isinstance(v, (bool, np.bool_)):
        return ("uni", "bool")
    if isinstance(v, (int, np.integer)):
        return ("uni", "i32")
    if isinstance(v, (float, np.floating)):
        return ("uni", "f32")
    if isinstance(v, Broadcast):
        t = v.plain
        dt = _TDT[t.dtype]
        if t.ndim == 0:
            return ("bcast", dt)
        if (t.ndim == 1 and t.shape[0] > DVEC_MAX) or (t.ndim >= 2 and t.shape[0] > DVEC_MAX and t.numel() // t.shape[0] <= DVEC_MAX):
            return ("dtab", dt, tuple(t.shape))          # a long vector, or a long table of short rows ([T, D])
        return ("dvec", dt, tuple(t.shape))
    if isinstance(v, Gathered):
        if tuple(v.ancestors.shape) != tuple(batch):
            raise ValueError(f"gathered value with batch {tuple(v.ancestors.shape)} in a launch over {batch}")
        return ("gather", _TDT[v.dtype], tuple(v.source.shape[1:]))
    if isinstance(v, torch.Tensor):
        dt = _TDT[v.dtype]
        shp = tuple(v.shape)
        # a CPU tensor handed to a GPU launch is DATA like any other tensor (it is copied to the device when the
        # launch is bound): the same classification as under the CPU mirror, where every tensor is a "device" tensor
        if shp[:nb] == tuple(batch) and nb > 0:
            return ("part", dt, shp[nb:])
        if v.ndim == 0:
            return ("bcast", dt)
        if nb == 0:
            return ("part", dt, shp)
        if (v.ndim == 1 and shp[0] > DVEC_MAX) or (v.ndim >= 2 and shp[0] > DVEC_MAX and v.numel() // shp[0] <= DVEC_MAX) \
                or (v.ndim >= 2 and v.numel() > POOL_VEC_MAX):
            return ("dtab", dt, shp)           # launch-uniform table read with OP_LDTAB ([T] or [T, *short row]; [few, many])
        return ("dvec", dt, shp)
    if isinstance(v, (list, tuple)):
        v = np.asarray(v)
    if isinstance(v, np.ndarray):
        if v.dtype == object:
            raise TypeError("object arrays cannot be launch values")
        if v.ndim == 0:
            return leaf_spec(v.item(), batch)
        if (v.ndim == 1 and v.shape[0] > GMX_HVEC_MAX) or (v.ndim >= 2 and v.shape[0] > GMX_HVEC_MAX and v.size // v.shape[0] <= DVEC_MAX) \
                or (v.ndim >= 2 and v.size > POOL_VEC_MAX):
            # too long for the operand pool: a table, like a long device vector (a [few, many] array too: one row per
            # element of a small plate, each read step by step by that element's long scan)
            return ("dtab", _np_dt(v), tuple(v.shape))
        return ("hvec", _np_dt(v), tuple(v.shape))
    if getattr(v, "__gmx_static__", False):
        _STATIC_KEEP[id(v)] = v            # host object the traced function only reads (e.g. a Target)
        return ("static", id(v))
    raise TypeError(f"unsupported launch value of type {type(v).__name__}")


class ProgramCache(OrderedDict):
    """Bounded LRU of compiled site programs (GENMI_PROGRAM_CACHE entries, default 1024): a long-running job whose
    shapes / structures keep changing would otherwise accumulate device code buffers and hiprtc modules without
    limit.  An evicted entry's program is destroyed (gmx_program_destroy) when its last reference goes — see
    Compiled.__init__'s finalizer — so bindings still in flight stay valid."""

    def __init__(self):
        super().__init__()
        self.limit = int(os.environ.get("GENMI_PROGRAM_CACHE", "1024"))

    def get(self, key, default=None):
        try:
            self.move_to_end(key)
            return self[key]
        except KeyError:
            return default

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        while len(self) > self.limit:
            self.popitem(last=False)


_ALL_CACHES: list = []


def new_cache() -> ProgramCache:
    c = ProgramCache()
    _ALL_CACHES.append(c)
    return c


def clear_caches():
    """Drop every cached program (and the host objects kept alive for traced functions)."""
    for c in _ALL_CACHES:
        c.clear()
    _STATIC_KEEP.clear()
    _UNIFORM_DEV.clear()


_STATIC_KEEP: dict = {}


class Flat:
    """Flatten nested tuples / lists / dicts / ChoiceMaps into leaves."""

    def __init__(self):
        self.leaves = []

    def add(self, v):
        if isinstance(v, (Patched, PlateScore)):
            v = v.materialize()
        if isinstance(v, (tuple, list)) and not _is_number_seq(v):
            return (type(v).__name__, tuple(self.add(x) for x in v))
        if isinstance(v, dict):
            return ("dict", tuple((k, self.add(x)) for k, x in v.items()))
        if isinstance(v, ChoiceMap):
            return ("chm", tuple((a, self.add(v[a] if a else v.get_value())) for a in v.addresses()))
        if dataclasses.is_dataclass(v) and not isinstance(v, type) and not getattr(v, "__gmx_static__", False):
            # a Pytree (`@Pytree.dataclass`, pytree.py:40-205): its fields are the leaves; fields declared with
            # Pytree.static() ride in the structure (and so in the program cache key)
            items = []
            for f_ in dataclasses.fields(v):
                x = getattr(v, f_.name)
                if f_.metadata.get("static"):
                    items.append((f_.name, ("static_field", _static_token(x))))
                else:
                    items.append((f_.name, self.add(x)))
            return ("dc", type(v), tuple(items))
        if isinstance(v, Indexed):            # a run-time plate index (core/mask.py): value + index, both launch values
            return ("indexed", self.add(v.value), self.add(v.idx))
        tk = _trace_kind(v)
        if tk is not None:
            # a TRACE as an argument (`prop(tr, *_): orig_a = tr.get_choices()["a"]`, docs/cookbook/inactive/inference/
            # mcmc.ipynb c8 / c10): traces are pytrees in the reference (generative_function.py:72-230, static.py:80-119)
            # — their arguments, return value, values and scores are the leaves, the generative function rides in the
            # structure (and so in the program cache key)
            gf = ("static_field", _static_token(v.gen_fn))
            if tk == "dist":
                return ("trace", tk, gf, self.add(v.args), self.add(v.value), self.add(v.score))
            if tk == "vmap":
                return ("trace", tk, gf, self.add(v.args), self.add(v.inner), self.add(v.score), self.add(v.retval))
            return ("trace", tk, gf, self.add(v.args), self.add(v.retval),
                    tuple((a, self.add(st)) for a, st in v.subtraces.items()))
        if isinstance(v, Mask):
            # a runtime-conditional constraint (distribution.py:129-142): a flag known on the host resolves now
            if isinstance(v.flag, (bool, np.bool_)):
                return ("mask_static", bool(v.flag), self.add(v.value))
            return ("mask", self.add(v.value), self.add(v.flag))
        if isinstance(v, Expanded):
            return self.add(v.base)        # (what a batched trace showed broadcast goes in as the launch-uniform value it is)
        self.leaves.append(v)
        return ("leaf", len(self.leaves) - 1)


def _trace_kind(v):
    """"dist" | "static" | "vmap" for the trace types of static.py (imported late: static imports this module), else None"""
    name = type(v).__name__
    if name not in ("DistributionTrace", "StaticTrace", "MaskTrace", "VmapTrace"):
        return None
    from . import static as S
    if isinstance(v, S.DistributionTrace):
        return "dist"
    if isinstance(v, S.VmapTrace):
        return "vmap"
    if isinstance(v, S.StaticTrace):
        return "static"
    return None


def _hashable(x):
    try:
        hash(x)
        return x
    except TypeError:
        return ("id", id(x))


def _static_token(x):
    """what stands for a Pytree.static() field in a launch structure (and so in a program cache key): the value itself
    when it is hashable; else a token of its contents (dict / list / set of hashables: the reference's static fields
    are compared by value) with the raw value kept in a side table — never an unhashable object inside a key"""
    try:
        hash(x)
        return ("v", x)
    except TypeError:
        pass

    def freeze(v):
        if isinstance(v, dict):
            return ("dict", tuple(sorted(((repr(k), freeze(w)) for k, w in v.items()))))
        if isinstance(v, (list, tuple)):
            return (type(v).__name__, tuple(freeze(w) for w in v))
        if isinstance(v, (set, frozenset)):
            return ("set", tuple(sorted(repr(w) for w in v)))
        try:
            hash(v)
            return v
        except TypeError:
            # never key a program on an object's address: after a collection the address is another object's
            raise TypeError(f"a Pytree.static() field holds a {type(v).__name__} that is neither hashable nor a dict / list / "
                            "set of hashable values: it cannot take part in a program cache key") from None
    tok = ("u", freeze(x))
    _STATIC_FIELDS[tok] = x
    _STATIC_FIELDS.move_to_end(tok)
    while len(_STATIC_FIELDS) > 1024:        # (a token is read back while its own call is being traced: recent ones suffice)
        _STATIC_FIELDS.popitem(last=False)
    return tok


def _static_value(tok):
    return tok[1] if tok[0] == "v" else _STATIC_FIELDS[tok]


_STATIC_FIELDS: OrderedDict = OrderedDict()


def _make_dataclass(cls, values: dict):
    obj = object.__new__(cls)
    for k, v in values.items():
        object.__setattr__(obj, k, v)
    return obj


def _is_number_seq(v):
    return len(v) > 0 and all(isinstance(x, (int, float, np.number)) and not isinstance(x, bool) for x in v) \
        and isinstance(v, list)


def unflatten(tree, fn):
    """Rebuild the structure, mapping leaf index -> fn(index)."""
    kind, payload = tree[0], tree[1]
    if kind == "leaf":
        return fn(payload)
    if kind in ("tuple", "list"):
        out = [unflatten(t, fn) for t in payload]
        return tuple(out) if kind == "tuple" else out
    if kind == "dict":
        return {k: unflatten(t, fn) for k, t in payload}
    if kind == "chm":
        cm = ChoiceMap.empty()
        for a, t in payload:
            cm = cm.set(a, unflatten(t, fn))
        return cm
    if kind == "mask":
        return Mask(unflatten(tree[1], fn), unflatten(tree[2], fn))
    if kind == "mask_static":
        return Mask(unflatten(tree[2], fn), tree[1])
    if kind == "indexed":
        return Indexed(unflatten(tree[1], fn), unflatten(tree[2], fn))
    if kind == "trace":
        from . import static as S
        tk, gf = tree[1], _static_value(tree[2][1])
        args = unflatten(tree[3], fn) if tree[3] is not None else None
        if tk == "dist":
            return S.DistributionTrace(gf, args, unflatten(tree[4], fn), unflatten(tree[5], fn))
        if tk == "vmap":
            return S.VmapTrace(gf, unflatten(tree[4], fn), unflatten(tree[5], fn), unflatten(tree[6], fn), args)
        subs = OrderedDict((a, unflatten(t, fn)) for a, t in tree[5])
        return S.StaticTrace(gf, args, unflatten(tree[4], fn), subs)
    if kind == "dc":
        return _make_dataclass(payload, {name: (_static_value(t[1]) if t[0] == "static_field" else unflatten(t, fn)) for name, t in tree[2]})
    raise ValueError(kind)


# ---------------------------------------------------------------------------
# trace time
# ---------------------------------------------------------------------------
class Sym:
    """A symbolic value plus where it came from (for pass-through outputs)."""
    __slots__ = ("value", "origin")

    def __init__(self, value, origin=None):
        self.value, self.origin = value, origin


class NoiseHoist:
    """Noise-ahead plan of one traced program: the standard-normal / unit-uniform draws whose KEY depends on nothing
    but a launch key and constants (`fold_in` chains from the particle's key: every `normal(...) @ addr` /
    `uniform(...) @ addr` site of a static model outside a counted loop, the accept draw of an MH move).  Such a draw —
    two Threefry blocks and an `erf_inv`, most of a bootstrap step's vector instructions — needs nothing the
    resampling chain produces, so a BACKGROUND program (static.NoiseProgram) can draw it steps ahead on a second
    stream; the traced program reads it as one more per-particle input leaf.
    `draws[k] = (root, chain of fold_in counters from the root key, element counter, kind)` is input leaf
    first_leaf + k; root "LDKEY" = the particle's launch key, "KSPLITU" = split((k0, k1), n)[i] of two launch values
    (the second key of a chained program, static.MinimalMHGenerate); kind "normal" | "uniform"."""

    def __init__(self, tr: "Tracing", first_leaf: int, roots=None):
        self.tr, self.first_leaf, self.draws = tr, int(first_leaf), []
        self._ksplitu = None
        self.roots = None if roots is None or roots is True else tuple(roots)     # None: draws of every root key

    def chain(self, node):
        c = []
        while node.op == "KDERIVE":
            c.append(int(node.imm))
            node = node.args[0]
        if node.op == "KSPLITU":
            if self._ksplitu is None:
                self._ksplitu = node
            if node is not self._ksplitu:
                return None
        elif node.op != "LDKEY":
            return None
        return node.op, tuple(reversed(c))

    def request(self, key_node, e: int, kind: str = "normal"):
        rc = self.chain(key_node)
        if rc is None or (self.roots is not None and rc[0] not in self.roots):
            return None
        j = self.first_leaf + len(self.draws)
        self.draws.append((rc[0], rc[1], int(e), kind))
        return self.tr.sym_leaf(("part", "f32", ()), j).value


class Tracing:
    def __init__(self, batch_ndim: int):
        self.graph = Graph()
        self.graph._tracing = self     # (distributions reach the launch plan through the graph: Tracing.spill_vector)
        self.in_plan = []      # (slot, leaf, elem, kind)
        self.uni_plan = []     # (uni index, leaf, elem, dtype)
        self.tab_plan = []     # (table slot, leaf): launch-uniform device tables
        self.outputs = []      # (dtype, n_elems, [slots])
        self.node_origin = {}  # id(node) -> origin
        self.prestored = {}    # id(node) -> slot of a store emitted early (plate elements)
        self.uses_key = False
        self.step_leaf_min = STEP_LEAF_MIN     # (a masked scan reads per-particle vectors step by step from 5 elements)
        self.alias_plan = []   # (input slot, output index): a [T, n] OUTPUT of this very launch read back as an input

    def alias_step_input(self, origin, dt, T):
        """The values a counted loop stored as element t of a [T, n] output, READ BACK inside the same program: one more
        input slot that `Compiled.bind` points at the output's own buffer.  A thread reads only what it stored itself
        (the rows of its particle), after the loop that stored it — program order through memory — so the model can
        compute with the values of a long vector-valued site: element j of the next vector site reads element j of this
        one in its own loop, a static index is one load."""
        g = self.graph
        slot = g.n_in
        g.n_in += 1
        self.alias_plan.append((slot, origin[1]))
        if not g.loop_counts:
            return StepAlias.make(g, slot, dt, T, origin)
        # inside enclosing loops (a plate of such models run as a loop): element (t_outer.., j) of the
        # [T_outer.., T, n] output can be read where the SAME outer iterations are open and the innermost loop runs over
        # the site's own length — the next vector site's loop; anything else (a static element, a sum) unrolls the site
        from .program import F_FLAT, F_STEP, F_U8
        flags = F_STEP | F_FLAT | (F_U8 if dt == "bool" else 0)
        outer = tuple(g.loop_ids)

        def read(i):
            if not (isinstance(i, Expr) and i.node.op == "LDT" and tuple(g.loop_ids[:-1]) == outer
                    and len(g.loop_counts) == len(outer) + 1 and g.loop_counts[-1] == int(T)):
                raise VectorSiteValueUsed("the values of a long vector-valued site inside a loop, read outside the "
                                          "loop of another site of the same length")
            return Expr(g.add("LDIN", dtype=dt, flags=flags, slot=slot))
        return StepOutputAlias(origin, int(T), len(self.outputs[origin[1]][1]), read)

    def spill_vector(self, exprs, dt="f32"):
        """K values held in registers, SPILLED: stored once as the K elements of one [K, n] scratch output of this launch
        (OP_STOUT flagged GMX_F_STEP with a static element) and read back where they are used — inside a counted loop as
        one load each (StepInput.read_in_loop) instead of K registers alive across the loop: the 64 log-probabilities of
        a mixture's `categorical(jnp.log(probs), sample_shape=n)` next to a loop over the n draws.  Top level only."""
        from .program import F_STEP, F_U8
        g = self.graph
        if g.loop_counts:
            raise NotImplementedError("spill_vector inside a counted loop")
        slot = g.n_out
        g.n_out += 1
        flags = F_STEP | (F_U8 if dt == "bool" else 0)
        for k, e in enumerate(exprs):
            g.add("STOUT", (e.node,), imm=k << 8, dtype="none", flags=flags, slot=slot)
        o = ("out", len(self.outputs))
        self.outputs.append((dt, (len(exprs),), ("step", slot, 1)))
        return self.alias_step_input(o, dt, len(exprs))

    # leaves -> symbols ------------------------------------------------------
    def sym_leaf(self, spec, j) -> Sym:
        g = self.graph
        kind = spec[0]
        if kind == "none":
            return Sym(None, ("leaf", j))
        if kind == "static":
            return Sym(_STATIC_KEEP[spec[1]], ("leaf", j))
        if kind == "uni":
            n = g.uniform(spec[1])
            self.uni_plan.append((n.imm, j, None, spec[1]))
            e = Expr(n)
            self.node_origin[id(n)] = ("leaf", j)
            return Sym(e, ("leaf", j))
        if kind == "hvec":
            arr = np.empty(spec[2], dtype=object)
            for e, idx in enumerate(np.ndindex(spec[2])):
                n = g.uniform(spec[1])
                self.uni_plan.append((n.imm, j, e, spec[1]))
                arr[idx] = Expr(n)
            return Sym(sym_array(arr), ("leaf", j))       # (a short vector in registers: a traced index selects, tracer.SymArray)
        if kind == "dtab":
            from .numpy import RuntimeTable, runtime_table_slot
            slot = runtime_table_slot(g)
            self.tab_plan.append((slot, j))
            tab = RuntimeTable.make(g, slot, spec[1], spec[2] if len(spec[2]) > 1 else spec[2][0])
            tab._leaf, tab._picks = j, ()
            return Sym(tab, ("leaf", j))
        flags = {"bcast": F_BCAST, "dvec": F_BCAST, "part": 0, "gather": F_GATHER}[kind]
        dt = spec[1]
        event = () if kind == "bcast" else spec[2]
        if kind == "part" and len(event) >= 2 and event[0] > self.step_leaf_min and int(np.prod(event[1:])) <= DVEC_MAX:
            # [n, T, *site event]: the values of a vector-valued site over the steps of a long scan — one [T, n] plane
            # per element of the site's event
            E = int(np.prod(event[1:]))
            slots = list(range(g.n_in, g.n_in + E))
            g.n_in += E
            for e_, slot in enumerate(slots):
                self.in_plan.append((slot, j, e_, "step2"))
            return Sym(StepInput.make2(g, slots, dt, int(event[0]), event[1:]), ("leaf", j))
        if kind == "part" and len(event) >= 3 and event[1] > self.step_leaf_min and int(np.prod(event[2:])) <= DVEC_MAX:
            # [n, A, T, *site event]: a vector-valued site of the long scans of a plate — one [A * T, n] slot per
            # element of the site's event
            E = int(np.prod(event[2:]))
            slots = list(range(g.n_in, g.n_in + E))
            g.n_in += E
            for e_, slot in enumerate(slots):
                self.in_plan.append((slot, j, e_, "stepflat2"))
            return Sym(StepInput2(g, slots, dt, (int(event[0]), int(event[1])), event=tuple(int(x) for x in event[2:]), leaf=j), ("leaf", j))
        if kind == "part" and len(event) == 3 and event[2] > self.step_leaf_min and (event[0] > DVEC_MAX or event[1] > DVEC_MAX
                                                                                      or int(np.prod(event)) > 4 * DVEC_MAX):
            # [n, A, B, T]: the choices of a plate of plates of plates (or of long scans two plates deep) — one slot
            # ([A * B * T, n]); element (a, b, t) through GMX_F_FLAT under the three loops
            slot = g.n_in
            g.n_in += 1
            self.in_plan.append((slot, j, 0, "stepflat"))
            return Sym(StepInput2(g, slot, dt, tuple(int(x) for x in event), leaf=j), ("leaf", j))
        if kind == "part" and len(event) == 2 and event[1] > self.step_leaf_min:
            # [n, A, T] with a long last axis: the choices of the long scans of a plate — one slot ([A * T, n]); row a is
            # picked statically (an unrolled plate) or by the outer loop's iteration number (a plate run as a loop
            # around the scans' loop: element (a, t) through GMX_F_FLAT)
            slot = g.n_in
            g.n_in += 1
            self.in_plan.append((slot, j, 0, "stepflat"))
            return Sym(StepInput2(g, slot, dt, (int(event[0]), int(event[1])), leaf=j), ("leaf", j))
        if kind == "part" and len(event) == 1 and event[0] > self.step_leaf_min:
            # a long per-particle vector (the [n, T] choices of a scan): one slot, element t read by iteration t
            slot = g.n_in
            g.n_in += 1
            self.in_plan.append((slot, j, 0, "step"))
            return Sym(StepInput.make(g, slot, dt, int(event[0])), ("leaf", j))
        if event == ():
            n = g.input(dt, flags)
            self.in_plan.append((n.slot, j, 0, kind))
            self.node_origin[id(n)] = ("leaf", j)
            return Sym(Expr(n), ("leaf", j))
        arr = np.empty(event, dtype=object)
        for e, idx in enumerate(np.ndindex(event)):
            n = g.input(dt, flags)
            self.in_plan.append((n.slot, j, e, kind))
            arr[idx] = Expr(n)
        return Sym(sym_array(arr), ("leaf", j))

    # outputs ------------------------------------------------------------------
    def prestore(self, value):
        """Store one element of a later vector-valued output NOW, so that its
        register dies here (plates are unrolled; see combinators.Vmap)."""
        from .tracer import Expr as _E
        if isinstance(value, Sym):
            value = value.value
        if isinstance(value, np.ndarray) and value.dtype == object:      # a nested plate's (or a vector site's) elements
            for v in value.reshape(-1):
                self.prestore(v)
            return
        if isinstance(value, _E) and value.node.op != "CONST" and id(value.node) not in self.prestored:
            self.prestored[id(value.node)] = self.graph.store(value.node)

    def store_step(self, value, T: int):
        """Inside a counted loop: store this iteration's value as element t of a [T, n] output (exposed to the
        user as [n, T], like a plate); inside the inner of two loops, as element (t_outer, t_inner) of a [T0, T1, n]
        output ([n, T0, T1]: a plate of scans).  Returns the output's origin."""
        from . import tracer as Tm
        dims = tuple(int(c) for c in self.graph.loop_counts) or (int(T),)
        # (a long ROW of a launch-uniform table never comes here: it is recorded by its origin — numpy.RuntimeTable.passthrough.
        #  A copy loop of its own inside the enclosing loop was tried and withdrawn: the two specialised kernels hiprtc got
        #  wrong on the GPU box — profiles/r05z_jit_miscompile.txt and a scan's lost score sum — both held that loop.)
        if isinstance(value, np.ndarray) and value.dtype == object:
            # a vector-valued site: one [T, n] plane per element, exposed as [n, T, *event]
            es = [Tm.lift(v) for v in value.reshape(-1)]
            if len({e.dtype for e in es}) != 1:
                raise TypeError("store_step: mixed element types")
            slots = [self.graph.store(e.node, step=True) for e in es]
            o = ("out", len(self.outputs))
            self.outputs.append((es[0].dtype, dims + tuple(value.shape), ("step", slots, len(dims))))
            return o
        e = Tm.lift(value)
        slot = self.graph.store(e.node, step=True)
        o = ("out", len(self.outputs))
        self.outputs.append((e.dtype, dims, ("step", slot, len(dims))))
        return o

    def emit_output(self, value):
        """Store a symbolic value (Expr or object array) unless it is a pure
        pass-through / constant; returns its origin."""
        if isinstance(value, StepOutput):
            return value.origin
        if isinstance(value, StepAlias) and value.origin is not None:
            return value.origin
        g = self.graph
        if value is None:
            return ("const", None)
        from .tracer import LazyVec
        if isinstance(value, LazyVec):
            # a long elementwise vector (`lw - lse` over a row of 1000 log-weights; a model returning `a * xs + b`): one
            # counted loop storing element t, instead of one unrolled copy per element
            from . import tracer as Tm
            if g.loop_counts:
                return self.emit_output(value.materialize())
            g.loop_begin(value.n)
            e = Tm.lift(value.at(Expr(g.add("LDT", dtype="i32"))))
            o = self.store_step(e, value.n)
            g.loop_end()
            return o
        if isinstance(value, Expr):
            o = self.node_origin.get(id(value.node))
            if o is not None:
                return o
            if value.node.op == "CONST":
                return ("const", _const_value(value.node))
            slot = g.store(value.node)
            o = ("out", len(self.outputs))
            self.outputs.append((value.dtype, (), [slot]))
            self.node_origin[id(value.node)] = o
            return o
        if isinstance(value, np.ndarray) and value.dtype == object:
            from . import tracer as T
            flat = [T.lift(v) for v in value.reshape(-1)]
            dts = {v.dtype for v in flat}
            dt = "f32" if "f32" in dts else ("i32" if "i32" in dts else "bool")
            conv = {"f32": T.as_float, "i32": T.as_int, "bool": T.as_bool}[dt]
            nodes = [conv(v).node for v in flat]
            ck = (tuple(id(n) for n in nodes), tuple(value.shape))
            if ck in self.node_origin:
                return self.node_origin[ck]       # e.g. a plate's values returned as its retval
            slots = []
            for n in nodes:                  # an early store feeds ONE output; a second user stores again
                slot = self.prestored.pop(id(n), None)
                slots.append(slot if slot is not None else g.store(n))
            o = ("out", len(self.outputs))
            self.outputs.append((dt, tuple(value.shape), slots))
            self.node_origin[ck] = o
            return o
        if isinstance(value, Mask):          # a MaskCombinator's return value (mask.py:71): value and flag
            return ("maskv", self.emit_output(value.value), self.emit_output(value.flag))
        if isinstance(value, (tuple, list)):
            return ("tuple" if isinstance(value, tuple) else "list", [self.emit_output(v) for v in value])
        if isinstance(value, dict):
            return ("dict", {k: self.emit_output(v) for k, v in value.items()})
        if dataclasses.is_dataclass(value) and not isinstance(value, type) and not getattr(value, "__gmx_static__", False):
            return ("dc", type(value), {f_.name: (("const", getattr(value, f_.name)) if f_.metadata.get("static")
                                                  else self.emit_output(getattr(value, f_.name)))
                                        for f_ in dataclasses.fields(value)})
        return ("const", value)


class StepInput(np.ndarray):
    """A per-particle vector leaf of T > 16 elements bound as ONE input slot ([T, n] struct-of-arrays): an object
    array of element reads (OP_LDIN flagged GMX_F_STEP, imm = element; unused ones are eliminated), so every vector
    operation works on it; indexing it with the iteration number of a counted loop is one such read with the
    element chosen at run time."""

    @classmethod
    def make(cls, g, slot, dt, T):
        from .program import F_STEP, F_U8
        flags = F_STEP | (F_U8 if dt == "bool" else 0)
        arr = np.empty((T,), dtype=object)
        for k in range(T):
            arr[k] = Expr(g.add("LDIN", dtype=dt, flags=flags, slot=slot, imm=k))
        out = arr.view(cls)
        out._g, out._slot, out._dt, out._flags = g, slot, dt, flags
        return out

    @classmethod
    def make2(cls, g, slots, dt, T, event):
        """a per-particle [T, *event] leaf (a vector-valued site's values over the steps): one slot ([T, n] plane)
        per element of the event; row t is the object array of its element reads"""
        from .program import F_STEP, F_U8
        flags = F_STEP | (F_U8 if dt == "bool" else 0)
        arr = np.empty((T,) + tuple(event), dtype=object)
        for k in range(T):
            for e, idx in enumerate(np.ndindex(tuple(event))):
                arr[(k,) + idx] = Expr(g.add("LDIN", dtype=dt, flags=flags, slot=slots[e], imm=k))
        out = arr.view(cls)
        out._g, out._slot, out._dt, out._flags = g, list(slots), dt, flags
        return out

    def __array_finalize__(self, obj):
        for a in ("_g", "_slot", "_dt", "_flags"):
            setattr(self, a, getattr(obj, a, None))
        self._base = getattr(obj, "_base", 0)        # element of the leaf this (contiguous) view starts at: `xs[1:]`

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        from .numpy import _lazy_ufunc
        return _lazy_ufunc(self, ufunc, method, inputs, kwargs)

    @property
    def _lazy_ok(self):
        return self._slot is not None and self.ndim == 1 and not isinstance(self._slot, list)

    def _read_at(self, idx):
        """element `idx` — a TRACED index that is not a loop's iteration number (`mus[z]` with z a categorical draw) — of a
        [T, n] leaf: ONE load at a register index (OP_LDIN flagged GMX_F_IDX), at any loop depth.  A negative index wraps
        once, then the index is clamped to the leaf (jax wraps and clamps a gather's indices the same way; no lane may
        read outside its own rows)."""
        from .program import F_U8
        from .tracer import as_int, where
        g = self._g
        n = int(self.shape[0])
        i = as_int(idx)
        i = where(i < 0, i + n, i)
        i = where(i > n - 1, n - 1, i)
        i = where(i < 0, 0, i)
        return Expr(g.add("LDINX", (i.node,), imm=int(self._base or 0), dtype=self._dt,
                          flags=(F_U8 if self._dt == "bool" else 0), slot=self._slot))

    def read_in_loop(self, k: int):
        """element k (a Python int) read INSIDE a counted loop: a GMX_F_STEP read there means element t + k, so the static
        element goes through the register-indexed form with a constant index (a pool entry: no register)"""
        from .program import F_U8
        g = self._g
        return Expr(g.add("LDINX", (g.const_i32(0),), imm=int(self._base or 0) + int(k), dtype=self._dt,
                          flags=(F_U8 if self._dt == "bool" else 0), slot=self._slot))

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and self.ndim == 1:          # `xs[..., 1:]` of a vector is `xs[1:]`
            rest = [i for i in idx if i is not Ellipsis]
            if len(rest) <= 1 and len(idx) - len(rest) <= 1:
                idx = rest[0] if rest else slice(None)
        if isinstance(idx, np.ndarray) and idx.dtype == object and self._slot is not None and self.ndim == 1 \
                and not isinstance(self._slot, list):
            out = np.empty(idx.shape, dtype=object)         # `mus[zs]`: one read per index
            for pos in np.ndindex(idx.shape):
                out[pos] = self[idx[pos]]
            return out
        if self._slot is not None and self.ndim == 1 and not isinstance(self._slot, list) \
                and not isinstance(idx, (Expr, slice, int, np.integer, tuple)) and idx is not Ellipsis:
            from . import tracer as Tm
            n_i = Tm._long_vector(idx)
            kind = getattr(getattr(idx, "dtype", None), "kind", "")
            if n_i and (kind in "iu" or getattr(idx, "_dt", None) == "i32") and not GATHER_LAZY_OFF[0]:
                # `theta[group]` with `group` a LONG vector of indices readable at a run-time position (a table of group
                # labels, a per-particle assignment vector): a recipe — element i is ONE load of `group` and ONE
                # register-indexed load of this leaf (GMX_F_IDX) inside the consuming site's loop — instead of one
                # unrolled read per index (a hierarchical model's `normal(theta[group], s) @ "y"` over thousands of rows)
                src = self

                def elem(i):
                    j = Tm._elem(idx, i)
                    if isinstance(j, (int, np.integer)):
                        return np.ndarray.__getitem__(src, int(j))
                    return src._read_at(j)
                out = Tm.LazyVec(n_i, elem, parts=(self, idx))
                out._dt = self._dt
                return out
        if isinstance(idx, Expr) and self._slot is not None:
            if self.ndim == 1 and not isinstance(self._slot, list):
                if idx.node.op != "LDT":
                    return self._read_at(idx)
                return Expr(self._g.add("LDIN", dtype=self._dt, flags=self._flags, slot=self._slot, imm=int(self._base or 0)))
            if isinstance(self._slot, list) and self.ndim >= 2 and int(np.prod(self.shape[1:])) == len(self._slot):
                if idx.node.op != "LDT":
                    raise NotImplementedError("a traced index that is not a loop's iteration number into the rows of a "
                                              "per-particle [T, *event] leaf")
                row = np.empty(self.shape[1:], dtype=object)          # this iteration's row, element by element
                for e, ix in enumerate(np.ndindex(self.shape[1:])):
                    row[ix] = Expr(self._g.add("LDIN", dtype=self._dt, flags=self._flags, slot=self._slot[e]))
                return row
        r = np.ndarray.__getitem__(self, idx)
        if isinstance(r, StepInput) and not (isinstance(idx, slice) or idx is Ellipsis):
            return np.asarray(r, dtype=object) if r.ndim else r      # a static row / element: plain expressions
        if isinstance(r, StepInput) and isinstance(idx, slice):
            # a view read at a loop's iteration number t is element base + t of the leaf (GMX_F_STEP adds its immediate):
            # `xs[1:]`, `xs[10:40]`; a reversed / strided view — or a slice of several axes — is its element reads,
            # plainly (what reads element t of the WHOLE leaf would get silently wrong)
            if self.ndim == 1 and not isinstance(self._slot, list) and idx.step in (None, 1):
                r._base = int(self._base or 0) + idx.indices(self.shape[0])[0]
            elif idx.indices(self.shape[0]) != (0, self.shape[0], 1):
                return np.asarray(r, dtype=object)
        return r


GATHER_LAZY_OFF = [0]       # (tests: the unrolled form of `leaf[index_vector]`, to hold the lazy one against)


class StepAlias(StepInput):
    """Tracing.alias_step_input: the [T, n] output `origin` of this launch as an array of element reads.  Returned or
    recorded as it is, it IS that output (emit_output: no second copy); views and slices are plain arrays of reads."""
    origin = None

    @classmethod
    def make(cls, g, slot, dt, T, origin):     # noqa: D102
        out = StepInput.make(g, slot, dt, T).view(cls)
        out.origin = origin
        return out

    def __array_finalize__(self, obj):
        StepInput.__array_finalize__(self, obj)
        self.origin = None


class StepInput2:
    """A per-particle leaf with SEVERAL axes whose LAST axis is long ([n, A, T] or [n, A, B, T] at the boundary, one
    [A * T, n] / [A * B * T, n] input slot): what a plate of long scans — or a plate of plates of plates — reads back
    (assess; importance / update with per-particle values).  `leaf[a]` picks a row — a a Python int (an unrolled plate)
    or the enclosing loop's iteration number (the plate run as a loop) — and the last index, inside the innermost loop,
    is ONE load: element (a, ..., t) of the slot, row-major (OP_LDIN, GMX_F_STEP; with the leading rows picked
    statically imm = their offset; under nested loops GMX_F_FLAT: the flat index is the loops' own)."""

    def __init__(self, g, slot, dt, shape, row=None, event=(), rows=(), full=None, leaf=None):
        # slot: one input slot, or (a vector-valued site: event != ()) one per element of the site's event
        self._g, self._slot, self._dt, self.shape, self._event = g, slot, dt, tuple(shape), tuple(event)
        self._leaf = leaf          # the launch leaf this is (Tracing.sym_leaf): a row of it recorded as it is needs no copy
        self._rows = tuple(rows) if rows else (() if row is None else (row,))       # picks so far: int | "loop"
        self._full = tuple(full) if full is not None else tuple(self._rows_shape()) + self.shape
        self.ndim = len(self.shape)

    def _rows_shape(self):
        return ()

    @property
    def _row(self):
        return self._rows[-1] if self._rows else None

    @property
    def _lazy_row(self):
        """ONE long row ([T] per particle, every leading axis picked): readable at the iteration number of a counted
        loop, so elementwise arithmetic on it stays a recipe (tracer.LazyVec) and a vector-valued site over it — the
        values of a plate's long vector sites given per particle: assess, update, importance — runs as a loop"""
        return self.ndim == 1 and not self._event

    def token(self):
        """ONE node that stands for "the contents of this leaf" in the change propagation of an edit (static._nodes_of):
        the reads of a leaf with several axes are made where they are used (a new node each), so a constraint given as
        such a leaf is marked changed — and a later site's arguments are found to depend on it — through this node; it
        is never executed (nothing stored depends on it)"""
        toks = self._g.__dict__.setdefault("_leaf_tokens", {})
        key = self._slot[0] if isinstance(self._slot, list) else self._slot
        t = toks.get(key)
        if t is None:
            t = toks[key] = self._g.add("TOKEN", (), imm=int(key), dtype="none")
        return t

    def passthrough(self):
        """this long row as the stacked output it already is — the values of a site GIVEN per particle (a constraint, the
        previous trace's): recorded by their origin, never copied: the whole leaf where every leading row was picked by
        a loop, the statically picked rows of it (an unrolled plate) otherwise"""
        if self._leaf is None or not self._lazy_row:
            return None
        static = tuple(int(r) for r in self._rows if r != "loop")
        n_loop = len(self._rows) - len(static)
        origin = ("leafrow", self._leaf, static, n_loop) if static else ("leaf", self._leaf)
        return StepOutput(origin, int(self.shape[0]), 1 + n_loop)

    def _lazy(name):          # noqa: N805  (the LazyVec operators, for a long row)
        def op(self, *o):
            from . import tracer as Tm
            if not self._lazy_row:
                raise NotImplementedError("arithmetic on a per-particle leaf with several axes: pick a row first")
            return getattr(Tm.LazyVec(self.shape[0], lambda i: self[i], parts=(self,)), name)(*o)
        return op
    for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__pow__",
               "__rpow__", "__neg__", "__abs__", "__lt__", "__le__", "__gt__", "__ge__", "__and__", "__rand__", "__or__",
               "__ror__", "__invert__", "astype"):
        locals()[_n] = _lazy(_n)
    del _n, _lazy
    __array_ufunc__ = None

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        from .program import F_FLAT, F_STEP, F_U8
        if self.ndim >= 2:
            if isinstance(idx, (int, np.integer)):
                if not -self.shape[0] <= int(idx) < self.shape[0]:
                    raise IndexError(idx)          # (the sequence protocol ends here: `for row in leaf` is finite)
                pick = int(idx) % self.shape[0]
            elif isinstance(idx, Expr):
                if idx.node.op != "LDT":
                    raise NotImplementedError("a step leaf with several axes takes its rows from the plates' own iteration numbers")
                pick = "loop"
            else:
                raise NotImplementedError("a per-particle leaf with a long last axis is read row by row (a plate of long scans)")
            return StepInput2(self._g, self._slot, self._dt, self.shape[1:], event=self._event, rows=self._rows + (pick,),
                              full=self._full, leaf=self._leaf)
        if not (isinstance(idx, Expr) and idx.node.op == "LDT"):
            raise NotImplementedError("a long per-particle row is read by the scan's own iteration number")
        flags = F_STEP | (F_U8 if self._dt == "bool" else 0)
        rows = self._rows
        k = 0                                   # leading rows picked statically; the rest by the loops around this read
        while k < len(rows) and rows[k] != "loop":
            k += 1
        if any(r != "loop" for r in rows[k:]):
            raise NotImplementedError("a static row of a step leaf below a row picked by a loop")
        n_loops = len(rows) - k + 1
        if len(self._g.loop_counts) != n_loops:
            raise NotImplementedError(f"element of a step leaf with {n_loops} loop-indexed axes read under "
                                      f"{len(self._g.loop_counts)} counted loops")
        imm = 0
        for j in range(k):                      # row-major offset of the statically picked leading rows
            imm = imm * self._full[j] + int(rows[j])
        imm *= int(np.prod(self._full[k:], dtype=np.int64))
        if n_loops >= 2:
            flags |= F_FLAT
        if not self._event:
            return Expr(self._g.add("LDIN", dtype=self._dt, flags=flags, slot=self._slot, imm=imm))
        out = np.empty(self._event, dtype=object)          # a vector-valued site: this step's row, element by element
        for e, ix in enumerate(np.ndindex(self._event)):
            out[ix] = Expr(self._g.add("LDIN", dtype=self._dt, flags=flags, slot=self._slot[e], imm=imm))
        return out


class VectorSiteValueUsed(NotImplementedError):
    """the model computes with the values of a vector-valued site that was lowered to a counted loop (they live in
    memory only): the caller traces again with such sites unrolled (static.without_vector_site_loops)"""


class StepOutput:
    """The stacked per-iteration outputs of a counted loop ([n, T] after the launch): they exist only in memory, so
    inside the program they can be returned / recorded but not computed with."""

    def __init__(self, origin, T, trailing=1, vector_site=False):
        self.origin, self.T = origin, T
        self.trailing = trailing           # dims after the batch: (T, *event) — what an unrolled plate stacks in front of
        self.vector_site = vector_site     # the values of ONE long vector-valued site (static._vector_site_loop)

    @staticmethod
    def stack(parts):
        """the same loop output of the n elements of an UNROLLED plate (a plate of long scans): [n, T, *event] per
        particle, assembled after the launch from the elements' own [T, n] leaves"""
        return StepOutput(("stack", [p.origin for p in parts], parts[0].trailing), parts[0].T, parts[0].trailing + 1)

    def _no(self, *a, **k):
        if self.vector_site:
            raise VectorSiteValueUsed("the model computes with the values of a long vector-valued site")
        raise NotImplementedError("the stacked outputs of a long scan live in memory only: return them, or use a "
                                  "scan of at most 16 steps (unrolled) to compute with them inside the model")
    __add__ = __radd__ = __mul__ = __rmul__ = __sub__ = __rsub__ = __getitem__ = __truediv__ = _no


def _make_step_output_alias():
    from .tracer import LazyVec

    class StepOutputAlias(LazyVec, StepOutput):
        """Tracing.alias_step_input INSIDE enclosing loops: the stored values of a looped vector site — to the
        combinators the stacked output it is (returned / recorded as such), to the next vector site a recipe whose
        element j is one read of that output in the site's own loop.  Anything else (a static element, a sum) raises
        VectorSiteValueUsed: the call is traced again with the site unrolled."""

        def __init__(self, origin, T, trailing, read):
            LazyVec.__init__(self, T, read)
            StepOutput.__init__(self, origin, T, trailing, vector_site=True)

        def materialize(self):
            raise VectorSiteValueUsed("the elements of a long vector-valued site inside a loop, one by one")
    return StepOutputAlias


StepOutputAlias = _make_step_output_alias()


def _const_value(node):
    if node.dtype == "f32":
        return struct.unpack("<f", struct.pack("<I", node.imm))[0]
    v = node.imm if node.imm < 0x80000000 else node.imm - (1 << 32)
    return bool(v) if node.dtype == "bool" else v


# ---------------------------------------------------------------------------
# launch time
# ---------------------------------------------------------------------------
_F32 = struct.Struct("<f")
_U32 = struct.Struct("<I")


def _bits(v, dt) -> int:
    if dt == "f32":
        return _U32.unpack(_F32.pack(float(np.float32(v))))[0]
    if dt == "bool":
        return 1 if v else 0
    return int(v) & 0xFFFFFFFF


# launches at least this large are worth the ~2 s one-off hiprtc compile
JIT_MIN_WORK = 1 << 22          # cumulative particles after which a repeatedly launched program is specialised
JIT_MIN_PARTICLES = 1 << 18


# Programs launched while a stream capture is in progress belong to the graph being built: the graph's kernel nodes
# point into the program's device code buffer / hiprtc module, so whoever owns the graph must keep the `Compiled`
# objects alive (smc.CapturedLoop does, through `holding_captured_programs`) — the program caches are bounded LRUs and
# `clear_caches()` is public, so a cache entry is not a lifetime guarantee.
_CAPTURE_HOLDERS: list = []
_PENDING_DESTROY: list = []


class holding_captured_programs:
    """with holding_captured_programs() as held: ... every Compiled launched under an active stream capture inside the
    block is appended to `held` (once)."""

    def __enter__(self):
        self.held = []
        _CAPTURE_HOLDERS.append(self.held)
        return self.held

    def __exit__(self, *exc):
        _CAPTURE_HOLDERS.remove(self.held)
        return False


def _capturing(be) -> bool:
    return bool(be.uses_streams) and torch.cuda.is_current_stream_capturing()


def _destroy_program(be, handle):
    """gmx_program_destroy frees device memory and unloads a module: not while a stream capture is in progress (a
    hipFree inside a capture invalidates it) — such a release is deferred to the next launch outside a capture."""
    try:
        if _capturing(be):
            _PENDING_DESTROY.append((be, handle))
            return
    except Exception:
        pass
    be.c.gmx_program_destroy(handle)


def _drain_pending_destroys():
    while _PENDING_DESTROY:
        be, handle = _PENDING_DESTROY.pop()
        be.c.gmx_program_destroy(handle)


class _WideArgs:
    """gmx_run_args without the slot limits: what `Compiled.bind` fills for a CHAIN of programs before every segment
    takes its own slots from it (program.split_graph)"""

    def __init__(self, n_in, n_out, n_uni, n_tab):
        self.in_d, self.out_d, self.uni, self.tab_d = [0] * n_in, [0] * n_out, [0] * n_uni, [0] * n_tab
        self.ancestors_d = self.keys_d = self.red_out_d = None
        self.key_mode, self.key0, self.key1, self.key_inner, self.index_offset, self.step_stride = 0, 0, 0, 0, 0, 0


class _ChainLink:
    """one program of a chain: its handle and which slots of the whole binding its own slots are"""

    def __init__(self, handle, seg, finalizer):
        self.handle, self.in_src, self.out_dst, self.uni_src, self.tab_src = handle, seg.in_src, seg.out_dst, seg.uni_src, seg.tab_src
        self.has_red, self.n_regs, self.finalizer = seg.has_red, int(seg.blob[3]), finalizer
        self.seg = seg


_DIGEST_SCOPES: list = []


class program_digest:
    """`with engine.program_digest() as d: ...; d.hex()`: a digest of the SET of site programs created inside the block
    (sha256 over the sorted sha256s of their blobs) AND of the code objects hiprtc made of them inside the block
    (gmx_program_code_hash).  The second part is needed: the same source compiles to DIFFERENT code when hiprtc runs
    inside a process rocprofv3 has preloaded its tool library into (measured, round 5: BASELINE config 5's kernel with
    176 VGPRs and 0.60 ms when first compiled under the profiler, 60 VGPRs and 0.35 ms otherwise — and the JIT disk
    cache then hands the profiler's build to later plain runs).  With the sha of the library this names the code a
    workload ran: what bench.py holds a counter profile of one of the other configs to (profiles/counters.json
    `configs[*].programs`); tools/prof_config.sh therefore warms the JIT cache with a plain run first."""

    def __enter__(self):
        self._set = set()
        _DIGEST_SCOPES.append(self._set)
        return self

    def __exit__(self, *exc):
        _DIGEST_SCOPES.remove(self._set)
        return False

    def hex(self) -> str:
        return hashlib.sha256("".join(sorted(self._set)).encode()).hexdigest()[:16]


class Compiled:
    """A created program + its binding plan.  `chain=True`: a graph beyond the launch slots / the 64 live values is
    cut into several programs launched one after another (program.split_graph), values in flight between them in
    scratch leaves — the reference's handlers walk any number of sites (static.py:254-380)."""

    def __init__(self, tr: Tracing, chain: bool = False):
        be = _lib.get()
        self.in_plan, self.uni_plan, self.outputs = tr.in_plan, tr.uni_plan, tr.outputs
        self.tab_plan, self.alias_plan = tr.tab_plan, tuple(tr.alias_plan)
        self.n_in, self.n_out, self.n_uni = tr.graph.n_in, tr.graph.n_out, tr.graph.n_uni
        self.uses_red = any(n.op in ("REDMAX", "REDLSE") for n in tr.graph.nodes)
        tabs = tr.graph.__dict__.get("tables", [])
        self.links = None
        self._be = be
        self._jit_tried = False
        if chain:
            segs, self.n_spill = split_graph(tr.graph, _lib.GMX_MAX_IN, _lib.GMX_MAX_OUT, _lib.GMX_MAX_UNI, _lib.GMX_MAX_TAB)
        else:
            self.blob, self.const_pool = compile_graph(tr.graph)
            if max(self.n_in, self.n_out) > _lib.GMX_MAX_IN or self.n_uni > _lib.GMX_MAX_UNI:
                raise ValueError("site program exceeds the ABI slot limits "
                                 f"(in={self.n_in}, out={self.n_out}, uni={self.n_uni})")
            if len(tabs) > _lib.GMX_MAX_TAB:
                raise ValueError(f"site program exceeds the ABI slot limits (tables={len(tabs)} > {_lib.GMX_MAX_TAB}: "
                                 "launch-uniform vectors of more than 16 elements)")
        self.tables = []
        for t in tabs:
            if t is None:                      # runtime table: bound per launch (tab_plan)
                self.tables.append(None)
                continue
            a = np.ascontiguousarray(t)
            if a.dtype.kind == "f":
                a = a.astype(np.float32)
            elif a.dtype.kind == "b":
                a = a.astype(np.int32)
            else:
                a = a.astype(np.int32)
            self.tables.append(torch.from_numpy(a).to(be.device))
        if chain:
            self.links = []
            for seg in segs:
                h = self._create(seg.blob)
                fin = weakref.finalize(self, _destroy_program, be, h)
                fin.atexit = False
                self.links.append(_ChainLink(h, seg, fin))
            red = [l for l in self.links if l.has_red]
            self.handle = (red[0] if red else self.links[-1]).handle
            self.blob = segs[-1].blob
            self._finalizer = lambda: [l.finalizer() for l in self.links]
            return
        self.handle = self._create(self.blob)
        # the device code buffer and the specialised module go when the last reference to this object does
        self._finalizer = weakref.finalize(self, _destroy_program, be, self.handle)
        self._finalizer.atexit = False       # at interpreter exit the process (and the HIP runtime) goes anyway

    def _create(self, blob):
        be = self._be
        handle = c_void_p()
        words = np.ascontiguousarray(blob, dtype=np.uint32)
        for log in _DIGEST_SCOPES:
            log.add(hashlib.sha256(words.tobytes()).hexdigest())
        be.check(be.c.gmx_program_create(words.ctypes.data_as(POINTER(c_uint32)), words.size, handle),
                 "gmx_program_create")
        return handle

    @property
    def max_regs(self) -> int:
        return max(l.n_regs for l in self.links) if self.links else int(self.blob[3])

    def close(self):
        """Release the program now (idempotent); the object must not be launched afterwards."""
        self._finalizer()

    def specialize(self, only_needed: bool = False) -> bool:
        """Compile the kernel specialised to this program (gmx_program_specialize:
        the interpreter partially evaluated by hiprtc; bit-identical results).
        Returns False — and keeps the interpreter — if hiprtc is unavailable or
        GENMI_JIT=0.  only_needed: of a chain of launches, only the links the interpreter cannot hold."""
        be = self._be
        handles = [l.handle for l in self.links] if self.links else [self.handle]
        if all(be.c.gmx_program_is_specialized(h) for h in handles):
            return True
        if self._jit_tried:
            return False
        if only_needed:
            self._jit_needed = True
        else:
            self._jit_tried = True
        ok = True
        for h, regs in zip(handles, [l.n_regs for l in self.links] if self.links else [int(self.blob[3])]):
            if be.c.gmx_program_is_specialized(h) or (only_needed and regs <= 31):
                continue
            one = be.c.gmx_program_specialize(h) == 0
            if one:
                for log in _DIGEST_SCOPES:       # the code object itself: hiprtc's output is not a function of the source
                    log.add("code:%016x" % int(be.c.gmx_program_code_hash(h)))    # alone (see program_digest)
            if not one and regs > 31:
                msg = be.c.gmx_last_error()
                raise _lib.GenmiError("this program keeps more than 31 values live per particle and therefore needs the "
                                      f"hiprtc-specialised kernel, which could not be built: {msg.decode() if msg else ''}")
            ok = ok and one
        return ok

    def set_background(self, lds_pad: int):
        """Before specialize(): this program is background work on a second stream (include/genmi.h:
        gmx_program_set_background) — priority 0, `lds_pad` bytes of unused LDS per workgroup as a residency cap."""
        self._be.check(self._be.c.gmx_program_set_background(self.handle, int(lds_pad)), "gmx_program_set_background")

    def is_specialized(self) -> bool:
        if self.links:
            return all(bool(self._be.c.gmx_program_is_specialized(l.handle)) for l in self.links)
        return bool(self._be.c.gmx_program_is_specialized(self.handle))

    def set_fuse_resample(self, loop: bool = False):
        """Before specialize(): the kernel can resample the previous step first (include/genmi.h: gmx_run_args.rs).
        loop: the LOOPED form — fewer workgroups than tiles, for launches of more than 2^20 particles."""
        if loop:
            self._be.check(self._be.c.gmx_program_set_fuse_resample_loop(self.handle), "gmx_program_set_fuse_resample_loop")
        else:
            self._be.check(self._be.c.gmx_program_set_fuse_resample(self.handle), "gmx_program_set_fuse_resample")

    def fuses_resample(self) -> bool:
        """True when `run(..., resample_in=...)` is honoured (include/genmi.h: gmx_program_fuses_resample)."""
        return self.links is None and bool(self._be.c.gmx_program_fuses_resample(self.handle))

    def set_fuse_shard_step(self):
        """Before specialize(): the kernel can route the previous step of a sharded sweep first (include/genmi.h:
        gmx_run_args.sh)."""
        self._be.check(self._be.c.gmx_program_set_fuse_shard_step(self.handle), "gmx_program_set_fuse_shard_step")

    def fuses_shard_step(self) -> bool:
        return self.links is None and bool(self._be.c.gmx_program_fuses_shard_step(self.handle))

    def resident_particles(self) -> int:
        """particles one launch of the specialised kernel covers with every workgroup resident at once (include/genmi.h:
        gmx_program_resident_particles); 0 when not specialised"""
        return 0 if self.links else int(self._be.c.gmx_program_resident_particles(self.handle))

    def writes_tile_stats(self) -> bool:
        """True when a launch can also leave the CDF tile statistics (gmx_run_args.tile_agg_d): a specialised
        program running 4 particles per thread with exactly one block-max reduction."""
        return self.links is None and bool(self._be.c.gmx_program_writes_tile_stats(self.handle))

    def run(self, leaves, batch: tuple, key: Key | None, red_out=None, index_offset=0, out_buffers=None,
            tile_stats=None, peer=None, resample_in=None, shard_in=None):
        """Bind and launch.  Returns the list of output tensors (shape batch+event)."""
        bound = self.bind(leaves, batch, key, red_out, index_offset, out_buffers, tile_stats, peer, resample_in, shard_in)
        if not getattr(self, "_checked", False) and self._be.uses_streams and self.is_partly_specialized() \
                and tile_stats is None and peer is None and resample_in is None and shard_in is None:
            self._cross_check(leaves, batch, key, index_offset)
        self.launch(bound)
        return bound[3]

    def is_partly_specialized(self) -> bool:
        hs = [l.handle for l in self.links] if self.links else [self.handle]
        return any(bool(self._be.c.gmx_program_is_specialized(h)) for h in hs)

    CHECK_PARTICLES = 256

    def _cross_check(self, leaves, batch, key, index_offset):
        """A second opinion on a freshly SPECIALISED program, once, before its results are used (include/genmi.h:
        gmx_program_despecialize; hiprtc on a GPU box was caught twice compiling one wrongly — DESIGN.md section 5): the
        first CHECK_PARTICLES particles of this launch run through the specialised kernel and through the ahead-of-time
        interpreter (a twin handle made from the same words) into scratch outputs, and every output is compared bit for
        bit.  On a difference the specialised module is dropped — this and every later launch take the interpreter — and
        the event is reported (warning, gmx_last_error, gmx_jit_rejected_count).  Programs the interpreter cannot hold
        (more than 31 live values), launches that reduce over the block or hand statistics to a resampler, and batches of
        more than one axis are not checked: the on-device fuzz stands guard there.  Cost: two launches of <= 256
        particles per program, once."""
        import copy
        import warnings
        self._checked = True
        be = self._be
        if len(batch) != 1 or self.uses_red or torch.cuda.is_current_stream_capturing():
            return
        handles = [l.handle for l in self.links] if self.links else [self.handle]
        regs = [l.n_regs for l in self.links] if self.links else [int(self.blob[3])]
        if max(regs) > 31:
            return
        n = int(batch[0])
        m = min(n, self.CHECK_PARTICLES)
        per_particle = {j for _, j, _, kind in self.in_plan if kind not in ("bcast", "dvec")}

        anc_cut = {}           # (the gathered leaves of one launch share ONE ancestor vector: so must their cuts)

        def cut(j, v):
            if j not in per_particle or isinstance(v, Broadcast):
                return v
            if isinstance(v, Gathered):
                return Gathered(v.source, anc_cut.setdefault(id(v.ancestors), v.ancestors[:m]))
            if isinstance(v, torch.Tensor) and v.ndim >= 1 and int(v.shape[0]) == n:
                return v[:m]
            return v
        small = [cut(j, v) for j, v in enumerate(leaves)]
        ks = key
        if key is not None and tuple(key.shape) == (n,):
            if key._lazy is not None:
                kind, base, _ = key._lazy
                if kind != "split":
                    return
                ks = Key(lazy=("split", base, m), offset=key._offset)
            else:
                ks = key[:m]
        elif key is not None and tuple(key.shape) != ():
            return
        twin = copy.copy(self)
        twin._checked, twin._jit_tried, twin._jit_needed, twin._work = True, True, True, 0
        made = []
        try:
            if self.links:
                twin.links = []
                for l in self.links:
                    h = self._create(l.seg.blob)
                    made.append(h)
                    twin.links.append(_ChainLink(h, l.seg, lambda: None))
                twin.handle = twin.links[-1].handle
            else:
                twin.handle = self._create(self.blob)
                made.append(twin.handle)
            ref = twin.bind(small, (m,), ks, None, index_offset)
            twin.launch(ref)
            keep_work = getattr(self, "_work", 0)
            got = self.bind(small, (m,), ks, None, index_offset)
            self._work = keep_work
            self.launch(got)
            bad = None
            for k, (a, b) in enumerate(zip(ref[3], got[3])):
                if a is None or b is None:
                    continue
                ai = a.contiguous().view(torch.int32) if a.dtype == torch.float32 else a
                bi = b.contiguous().view(torch.int32) if b.dtype == torch.float32 else b
                if not torch.equal(ai, bi):
                    bad = k
                    break
        finally:
            torch.cuda.synchronize(be.device) if be.uses_streams else None
            for h in made:
                be.c.gmx_program_destroy(h)
        if bad is not None:
            why = (f"output {bad} of a {len(handles)}-launch program differs between the hiprtc-specialised kernel and the "
                   f"interpreter on the first {m} particles").encode()
            for h in handles:
                be.c.gmx_program_despecialize(h, why)
            self._jit_tried = True
            warnings.warn("genjax_amd: a specialised kernel disagreed with the interpreter on its first launch and was "
                          "dropped (the interpreter runs this program from now on): " + why.decode(), RuntimeWarning, stacklevel=3)

    def launch(self, bound):
        """Launch a binding made by `bind` (its buffers must still be alive: `bound` keeps them)."""
        be = self._be
        if int(bound[0]) == 0:
            return          # an empty batch (jax.vmap over zero keys): the outputs are empty tensors already
        if _CAPTURE_HOLDERS or _PENDING_DESTROY:
            if _capturing(be):
                for held in _CAPTURE_HOLDERS:
                    if not any(c is self for c in held):
                        held.append(self)
            elif _PENDING_DESTROY:
                _drain_pending_destroys()
        if self.links:                      # a chain: every program in turn on the one stream, same particles
            for link, A in zip(self.links, bound[1]):
                be.check(be.c.gmx_program_run(link.handle, bound[0], A, be.stream()), "gmx_program_run")
            return
        be.check(be.c.gmx_program_run(self.handle, bound[0], bound[1], be.stream()), "gmx_program_run")

    def bind(self, leaves, batch: tuple, key: Key | None, red_out=None, index_offset=0, out_buffers=None,
             tile_stats=None, peer=None, resample_in=None, shard_in=None):
        """Fill a gmx_run_args for these leaves: (n, args, keep-alive list, outputs).  Sweeps whose
        buffers are persistent bind every step once and re-launch the bindings.
        tile_stats = (int64 tensor [grid], shift): the launch also writes the per-workgroup fixed-point weight sums
        gmx_resample_tiles consumes (only if `writes_tile_stats()`).
        peer = a _lib.Peer (with tile_stats): the launch also puts its tile statistics into the other ranks' landing
        tables (include/genmi.h "Fused peer exchange").
        resample_in = dict(lw, tile_max, tile_agg, shift, key=(k0, k1), tag, max_out, total_out, status): the launch
        first RESAMPLES the previous step (gmx_run_args.rs, `fuses_resample()`): every workgroup writes its tile's
        offspring into the ancestors tensor the gathered leaves name (tagged words), waits for the words of its own
        particles and gathers through them."""
        be = self._be
        n = int(np.prod(batch, dtype=np.int64))
        self._work = getattr(self, "_work", 0) + n
        if be.uses_streams and not torch.cuda.is_current_stream_capturing():
            if (n >= JIT_MIN_PARTICLES or self._work >= JIT_MIN_WORK) and not self._jit_tried:
                # big ensembles and programs launched often enough to repay ~0.5 s of hiprtc
                self.specialize()
            elif self.max_regs > 31 and not self._jit_tried and not getattr(self, "_jit_needed", False):
                # ... and what the 31-register interpreter cannot hold — of a chain of launches, those links only: the
                # others stay on the interpreter until the work repays a compile
                self.specialize(only_needed=True)
        if self.links:
            if tile_stats is not None or peer is not None or resample_in is not None or shard_in is not None:
                raise NotImplementedError("a site program cut into a chain of launches takes no tile statistics / peers / "
                                          "fused resampling")
            A = _WideArgs(self.n_in, self.n_out, self.n_uni, len(self.tables))
        else:
            A = _lib.RunArgs()
        keep = []
        anc = None
        soa_cache = {}
        for slot, j, e, kind in self.in_plan:
            v = leaves[j]
            if isinstance(v, Broadcast):
                v = v.plain
            if kind == "gather":
                if anc is None:
                    anc = v.ancestors
                elif anc is not v.ancestors:
                    raise ValueError("all gathered values of one launch must share their ancestors")
                src = v.source
                key_ = (j, "g")
            else:
                src = v
                key_ = (j, kind)
            buf = soa_cache.get(key_)
            if buf is None:
                if kind == "step2":        # [n, T, *event] -> [E, T, n] planes (a copy: the step stride is n for every leaf)
                    t_ = _prepare_input(src, "bcast", None, be)
                    buf = t_.reshape(n, t_.shape[len(batch)], -1).permute(2, 1, 0).contiguous()
                elif kind == "stepflat":   # [n, T0, T1] -> [T0 * T1, n]
                    t_ = _prepare_input(src, "bcast", None, be)
                    buf = t_.reshape(n, -1).t().contiguous()
                elif kind == "stepflat2":  # [n, A, T, *event] -> [E, A * T, n] planes
                    t_ = _prepare_input(src, "bcast", None, be)
                    nb_ = len(batch)
                    buf = t_.reshape(n, t_.shape[nb_] * t_.shape[nb_ + 1], -1).permute(2, 1, 0).contiguous()
                else:
                    buf = _prepare_input(src, "part" if kind == "step" else kind, n if kind in ("part", "step") else None, be)
                soa_cache[key_] = buf
                keep.append(buf)
            item = buf.element_size()
            if kind in ("step2", "stepflat2"):
                A.in_d[slot] = buf.data_ptr() + e * buf.shape[1] * n * item
                A.step_stride = n
            elif kind == "stepflat":
                A.in_d[slot] = buf.data_ptr()
                A.step_stride = n
            elif kind == "step":
                if buf.dim() != 2 or buf.stride(1) != 1 or buf.stride(0) != n:
                    buf = buf.contiguous()
                    keep.append(buf)
                A.in_d[slot] = buf.data_ptr()
                A.step_stride = n
            elif kind in ("bcast",):
                A.in_d[slot] = buf.data_ptr()
            elif kind == "dvec":
                A.in_d[slot] = buf.data_ptr() + e * item
            else:   # part / gather: [E, rows] SoA, element rows `stride(0)` apart
                rows = buf.stride(0) if buf.shape[0] > 1 else buf.shape[-1]
                A.in_d[slot] = buf.data_ptr() + e * rows * item
        if anc is not None:
            a32 = anc.reshape(-1)
            if a32.dtype != torch.int32:
                a32 = a32.to(torch.int32)
            if a32.device != be.device:
                a32 = a32.to(be.device)
            keep.append(a32)
            A.ancestors_d = a32.data_ptr()
        for ui, j, e, dt in self.uni_plan:
            v = leaves[j]
            if e is not None:
                v = np.asarray(v).reshape(-1)[e]
            A.uni[ui] = _bits(v, dt)
        for s, t in enumerate(self.tables):
            if t is not None:
                A.tab_d[s] = t.data_ptr()
        for slot, j in self.tab_plan:
            tv = leaves[j]
            if isinstance(tv, Broadcast):
                tv = tv.plain
            if isinstance(tv, np.ndarray):                 # a long host vector (leaf_spec: "dtab")
                tv = torch.from_numpy(np.ascontiguousarray(tv.astype(np.float32) if tv.dtype.kind == "f" else tv.astype(np.int32)))
            t = _prepare_input(tv, "dvec", None, be)
            keep.append(t)
            A.tab_d[slot] = t.data_ptr()
        outs = []
        for k, (dt, event, slots) in enumerate(self.outputs):
            if isinstance(slots, tuple) and slots[0] == "step":
                # element t of a [T, n] leaf is written by iteration t of the program's loop (GMX_F_STEP)
                Tn = int(np.prod(event[:slots[2]]))       # one loop: T planes; two nested loops: T0 * T1, row-major
                if isinstance(slots[1], list):          # a vector-valued site: [E, T, n] planes -> [*batch, T, *site event]
                    E = len(slots[1])
                    buf = torch.empty((E, Tn, n), dtype=_STORE[dt], device=be.device)
                    for e_, slot in enumerate(slots[1]):
                        A.out_d[slot] = buf.data_ptr() + e_ * Tn * n * buf.element_size()
                    A.step_stride = n
                    outs.append(buf.permute(2, 1, 0).reshape(tuple(batch) + tuple(event)))
                    continue
                buf = torch.empty((Tn, n), dtype=_STORE[dt], device=be.device)
                A.out_d[slots[1]] = buf.data_ptr()
                A.step_stride = n
                outs.append(buf.reshape(event + tuple(batch)).permute(
                    *range(len(event), len(event) + len(batch)), *range(len(event))))
                continue
            E = len(slots)
            if out_buffers is not None and out_buffers[k] is not None:
                buf = out_buffers[k]
            else:
                buf = torch.empty((E, n), dtype=_STORE[dt], device=be.device)
            item = buf.element_size()
            # a caller's buffer may be a [E, n] window of wider rows (stride(0) > n): element rows stay
            # contiguous, which is all the kernel needs
            row = n
            if buf.dim() == 2 and buf.shape[0] == E and E > 1 and buf.stride(1) == 1:
                row = buf.stride(0)
            for e, slot in enumerate(slots):
                A.out_d[slot] = buf.data_ptr() + e * row * item
            if event == ():
                outs.append(buf.reshape(batch))
            elif row != n:
                if len(event) != 1 or len(batch) != 1:
                    raise ValueError("a strided output buffer is supported for one event axis and one batch axis")
                outs.append(buf.t())
            else:
                outs.append(buf.reshape(event + tuple(batch)).permute(
                    *range(len(event), len(event) + len(batch)), *range(len(event))))
        for slot, k in self.alias_plan:          # an output of this launch read back by it (Tracing.alias_step_input)
            A.in_d[slot] = A.out_d[self.outputs[k][2][1]]
            A.step_stride = n
        if key is not None:
            mode, k0, k1, kt, inner = key.binding()
            A.key_mode, A.key0, A.key1, A.key_inner = mode, k0, k1, inner
            if kt is not None:
                keep.append(kt)
                A.keys_d = kt.data_ptr()
        else:
            A.key_mode = _lib.KEY_NONE
        A.index_offset = index_offset or (getattr(key, "_offset", 0) if key is not None else 0)
        if self.uses_red:
            if red_out is None:
                grid = be.c.gmx_program_grid(self.handle, n)
                red_out = torch.empty((2, grid), dtype=torch.float32, device=be.device)
            A.red_out_d = red_out.data_ptr()
            self.last_red = red_out
            keep.append(red_out)
        if tile_stats is not None:
            agg, shift = tile_stats[0], tile_stats[1]
            A.tile_agg_d, A.tile_shift = agg.data_ptr(), int(shift)
            keep.append(agg)
            if peer is not None:
                A.peer = peer
        if resample_in is not None:
            r = resample_in
            if anc is None:
                raise ValueError("resample_in: the program gathers nothing")
            for name in ("lw", "tile_max", "tile_agg", "max_out", "total_out", "status"):
                keep.append(r[name])
            A.rs.lw_d, A.rs.tile_max_d, A.rs.tile_agg_d = r["lw"].data_ptr(), r["tile_max"].data_ptr(), r["tile_agg"].data_ptr()
            A.rs.max_out_d, A.rs.total_out_d, A.rs.status_d = r["max_out"].data_ptr(), r["total_out"].data_ptr(), r["status"].data_ptr()
            A.rs.shift, A.rs.tag = int(r["shift"]), int(r["tag"])
            A.rs.key0, A.rs.key1 = int(r["key"][0]), int(r["key"][1])
        if shard_in is not None:
            # the routing of the previous step of a SHARDED sweep first (gmx_run_args.sh, `fuses_shard_step()`): dict(lw,
            # stats_own, plan, total_out, max_out, status, shift, tag, key=(k0, k1), peer=_lib.Peer, state=[tensors], tail=[tensors])
            r = shard_in
            if anc is None:
                raise ValueError("shard_in: the program gathers nothing")
            for name in ("lw", "stats_own", "plan", "total_out", "max_out", "status"):
                keep.append(r[name])
            keep += list(r["state"]) + list(r["tail"])
            S = A.sh
            S.lw_d, S.stats_own_d, S.plan_d = r["lw"].data_ptr(), r["stats_own"].data_ptr(), r["plan"].data_ptr()
            S.total_out_d, S.max_out_d, S.status_d = r["total_out"].data_ptr(), r["max_out"].data_ptr(), r["status"].data_ptr()
            S.shift, S.tag = int(r["shift"]), int(r["tag"])
            S.key0, S.key1 = int(r["key"][0]), int(r["key"][1])
            S.peer = r["peer"]
            for l, (st_, tl_) in enumerate(zip(r["state"], r["tail"])):
                S.state_d[l], S.tail_d[l] = st_.data_ptr(), tl_.data_ptr()
        if self.links:
            return n, self._chain_args(A, n, keep), keep, outs
        return n, A, keep, outs

    def _chain_args(self, W: "_WideArgs", n: int, keep: list):
        """one gmx_run_args per program of the chain: its own slots out of the whole binding `W`, its spills and reloads
        in rows of one scratch buffer ([words, n] 32-bit cells, struct-of-arrays like every leaf)"""
        be = self._be
        scratch = torch.empty((max(self.n_spill, 1), max(n, 1)), dtype=torch.int32, device=be.device)
        keep.append(scratch)
        base, row = scratch.data_ptr(), max(n, 1) * 4
        out = []
        for link in self.links:
            S = _lib.RunArgs()
            for k, (kind, x) in enumerate(link.in_src):
                S.in_d[k] = W.in_d[x] if kind == "in" else base + x * row
            for k, (kind, x) in enumerate(link.out_dst):
                S.out_d[k] = W.out_d[x] if kind == "out" else base + x * row
            for k, x in enumerate(link.uni_src):
                S.uni[k] = W.uni[x]
            for k, x in enumerate(link.tab_src):
                S.tab_d[k] = W.tab_d[x]
            S.ancestors_d, S.key_mode, S.key0, S.key1, S.keys_d = W.ancestors_d, W.key_mode, W.key0, W.key1, W.keys_d
            S.key_inner, S.index_offset, S.step_stride = W.key_inner, W.index_offset, W.step_stride
            if link.has_red:
                S.red_out_d = W.red_out_d
            out.append(S)
        return out


def over_the_slots(e) -> bool:
    """does this exception say "the traced program does not fit ONE launch" (slots, tables, live values)?"""
    return isinstance(e, ProgramTooLarge) or (isinstance(e, ValueError) and "exceeds the ABI slot limits" in str(e))


def compile_fitting(tr: Tracing) -> Compiled:
    """`Compiled(tr)`, as a chain of launches when one does not hold it"""
    try:
        return Compiled(tr)
    except Exception as e:      # noqa: BLE001
        if not over_the_slots(e):
            raise
    return Compiled(tr, chain=True)


def _prepare_input(src, kind, n, be):
    """Make the tensor the kernel reads: 4-byte (or bool) elements, SoA rows."""
    t = src
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"expected a tensor, got {type(t).__name__}")
    if t.dtype in (torch.float64, torch.float16, torch.bfloat16):
        t = t.to(torch.float32)
    elif t.dtype in (torch.int64, torch.int16, torch.int8, torch.uint8):
        t = t.to(torch.int32)
    if t.device != be.device:
        t = t.to(be.device)
    if kind in ("bcast", "dvec"):
        return t.contiguous()
    rows = n if kind == "part" else t.shape[0]
    if rows == 0:       # an empty batch: reshape(0, -1) is ambiguous; the launch is skipped anyway
        return t.new_zeros((int(np.prod(t.shape[1:], dtype=np.int64)) if t.dim() > 1 else 1, 0))
    flat = t.reshape(rows, -1)
    soa = flat.t()                      # [E, rows]
    if soa.stride(1) == 1 or soa.shape[1] == 1:
        return soa                      # already struct-of-arrays — possibly a [E, rows] WINDOW of wider rows
                                        # (stride(0) > rows): bound in place, so a persistent buffer stays live
    return soa.contiguous()


def resolve(origin, outs, leaves):
    """origin -> concrete launch result."""
    kind = origin[0]
    if kind == "out":
        return outs[origin[1]]
    if kind == "leaf":
        return leaves[origin[1]]
    if kind == "leafrow":      # statically picked leading rows of a per-particle leaf ([*batch, A, .., T] -> [*batch, T])
        v = leaves[origin[1]]
        if isinstance(v, Broadcast):
            v = v.plain
        if not isinstance(v, torch.Tensor):
            v = np.asarray(v)
        # [*batch, static rows .., looped rows .., T]: the batch axes are what is left in front
        nb = (v.dim() if isinstance(v, torch.Tensor) else v.ndim) - (len(origin[2]) + int(origin[3]) + 1)
        return v[(slice(None),) * nb + tuple(origin[2])]
    if kind == "const":
        return origin[1]
    if kind in ("tuple", "list"):
        seq = [resolve(o, outs, leaves) for o in origin[1]]
        return tuple(seq) if kind == "tuple" else seq
    if kind == "dict":
        return {k: resolve(o, outs, leaves) for k, o in origin[1].items()}
    if kind == "dc":
        return _make_dataclass(origin[1], {k: resolve(o, outs, leaves) for k, o in origin[2].items()})
    if kind == "maskv":
        return Mask(resolve(origin[1], outs, leaves), resolve(origin[2], outs, leaves))
    if kind == "stack":        # an unrolled plate of loop outputs: elements [*batch, T, *event] -> [*batch, n, T, *event]
        parts = [resolve(o, outs, leaves) for o in origin[1]]
        if not isinstance(parts[0], torch.Tensor):           # rows of a launch-uniform HOST table, given as they were
            parts = [np.asarray(p_) for p_ in parts]
            return np.stack(parts, axis=max(0, parts[0].ndim - int(origin[2])))
        return torch.stack(parts, dim=parts[0].dim() - int(origin[2]))
    if kind == "prepend":      # iterate / accumulate over a counted loop: [init, ys[0], ..., ys[T-1]] along the step axis
        init, ys = resolve(origin[1], outs, leaves), resolve(origin[2], outs, leaves)
        ax = ys.dim() - int(origin[3])
        head = torch.as_tensor(init, device=ys.device).to(ys.dtype)
        head = head.expand(ys.shape[:ax] + ys.shape[ax + 1:]).unsqueeze(ax)
        return torch.cat([head, ys], dim=ax)
    raise ValueError(origin)


# ---------------------------------------------------------------------------
# thin wrappers over the remaining C-ABI entry points
# ---------------------------------------------------------------------------
def gather_leaves(leaves: list, ancestors: torch.Tensor) -> list:
    """dst[l][j] = src[l][ancestors[j]] for every leaf (gmx_gather)."""
    be = _lib.get()
    anc = ancestors.reshape(-1)
    if anc.dtype != torch.int32:
        anc = anc.to(torch.int32)
    n_out = anc.numel()
    srcs, dsts, sizes, outs = [], [], [], []
    for t in leaves:
        n_src = t.shape[0]
        event = tuple(t.shape[1:])
        soa = t.reshape(n_src, -1).t().contiguous()           # [E, n_src]
        E = soa.shape[0]
        dst = torch.empty((E, n_out), dtype=t.dtype, device=t.device)
        for e in range(E):
            srcs.append(soa.data_ptr() + e * n_src * soa.element_size())
            dsts.append(dst.data_ptr() + e * n_out * dst.element_size())
            sizes.append(soa.element_size())
        outs.append((dst, event, soa))
    if srcs:
        L = len(srcs)
        S = (c_void_p * L)(*srcs)
        D = (c_void_p * L)(*dsts)
        Z = (_lib.c_int32 * L)(*sizes)
        be.check(be.c.gmx_gather(cast(S, POINTER(c_void_p)), cast(D, POINTER(c_void_p)), Z, L,
                                 be.ptr(anc), n_out, be.stream()), "gmx_gather")
    res = []
    bshape = tuple(ancestors.shape)
    for dst, event, _ in outs:
        if event == ():
            res.append(dst.reshape(bshape))
        else:
            res.append(dst.reshape(event + bshape).permute(
                *range(len(event), len(event) + len(bshape)), *range(len(event))))
    return res


_EW_CACHE = new_cache()


def elementwise(fn, *xs, key=None):
    """`fn(*xs)` evaluated per particle in ONE launch of a traced program (with `key`: `fn(key, *xs)`, the launch's
    key as its first argument — ONE key for every element, e.g. a sampler whose element counter is the index): the float algebra the
    combinators do between GFI launches (`new_w - score + w`, `lw + w`, a trace's score = sum of its
    site scores, `logits - lse`) runs in the same kernels as everything else, not in torch ops.
    xs: tensors sharing a leading batch shape (the longest one's), 0-d tensors, Python numbers."""
    from . import tracer as T
    vals = [materialize(v) for v in xs]
    tens = [v for v in vals if isinstance(v, torch.Tensor) and v.ndim > 0]
    if not tens and not any(isinstance(v, torch.Tensor) for v in vals):
        return fn(*vals)                 # Python numbers only
    if not tens:                         # 0-d device values (an unbatched trace): a launch over one particle, f32 like
        tens = [v for v in vals if isinstance(v, torch.Tensor)]          # everything else (not float64 on the host)
    lead = max((tuple(v.shape) for v in tens), key=len)
    nb = len(lead)
    for v in tens:                       # the batch is the shape every tensor shares as a prefix
        k = 0
        s_ = tuple(v.shape)
        while k < min(len(s_), nb) and s_[k] == lead[k]:
            k += 1
        nb = min(nb, k)
    batch = tuple(lead[:nb])
    flat = Flat()
    tree = flat.add(tuple(vals))
    specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
    code = getattr(fn, "__code__", None)
    ck = (code, tree, specs, key is not None)
    ent = _EW_CACHE.get(ck) if code is not None else None
    if ent is None:
        tr = Tracing(len(batch))
        with T.tracing(tr.graph):
            syms = [tr.sym_leaf(sp, j) for j, sp in enumerate(specs)]
            ins = unflatten(tree, lambda j: syms[j].value)
            out = fn(*ins) if key is None else fn(T.Expr(tr.graph.add("LDKEY", dtype="key")), *ins)
            if isinstance(out, T.Expr) and tr.node_origin.get(id(out.node)) is not None:
                out = out + 0.0           # a pure pass-through still gets its own buffer
            oo = tr.emit_output(out)
        ent = (compile_fitting(tr), oo)
        if code is not None and not fn.__closure__:
            _EW_CACHE[ck] = ent
    comp, oo = ent
    outs = comp.run(flat.leaves, batch, key)
    return resolve(oo, outs, flat.leaves)


def logsumexp_rows(lw: torch.Tensor) -> torch.Tensor:
    """logsumexp over the last axis (gmx_logsumexp); lw: [..., cols]."""
    be = _lib.get()
    cols = lw.shape[-1]
    rows = int(np.prod(lw.shape[:-1], dtype=np.int64))
    x = lw.reshape(rows, cols)
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    ws_bytes = be.c.gmx_logsumexp_workspace(rows, cols)
    ws = torch.empty((max(ws_bytes, 16) + 3) // 4, dtype=torch.int32, device=x.device)
    be.check(be.c.gmx_logsumexp(be.ptr(x), rows, cols, be.ptr(out), None, be.ptr(ws), be.stream()),
             "gmx_logsumexp")
    return out.reshape(lw.shape[:-1])


def sum_rows_inorder(x) -> torch.Tensor:
    """[B, n] -> [B]: every row added in element order (gmx_sum_rows_inorder): the plate score of a batched plate trace"""
    be = _lib.get()
    x = materialize(x)
    if x.dtype != torch.float32:
        x = x.float()
    rows, cols = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    be.check(be.c.gmx_sum_rows_inorder(be.ptr(x), rows, cols, int(x.stride(0)), int(x.stride(1)), be.ptr(out), be.stream()),
             "gmx_sum_rows_inorder")
    return out


class PlateScore:
    """the score of a plate trace whose element `idx` was just replaced: the in-order sum over the elements' scores,
    computed when somebody asks (the reference's VmapTrace sums on access too)"""

    def __init__(self, elem_scores, batch=None):
        """elem_scores: [B, n] (a tensor or a lazy leaf), or a thunk giving it (then `batch` = (B,) is needed)"""
        self.elem_scores, self._v = elem_scores, None
        self._batch = tuple(batch) if batch is not None else tuple(elem_scores.shape[:1])

    @property
    def shape(self):
        return self._batch

    def materialize(self):
        if self._v is None:
            es = self.elem_scores() if callable(self.elem_scores) else self.elem_scores
            self._v = sum_rows_inorder(es)
        return self._v


def sum_rows(x) -> torch.Tensor:
    """sum over the last axis in the device's fixed tree (gmx_sum_rows): the plate score of a launch-axis plate."""
    be = _lib.get()
    x = materialize(x)
    cols = x.shape[-1]
    rows = int(np.prod(x.shape[:-1], dtype=np.int64))
    v = x.reshape(rows, cols)
    if v.dtype != torch.float32:
        v = v.float()
    v = v.contiguous()
    out = torch.empty((rows,), dtype=torch.float32, device=v.device)
    ws = torch.empty((max(int(be.c.gmx_sum_rows_workspace(rows, cols)), 16) + 3) // 4, dtype=torch.int32, device=v.device)
    be.check(be.c.gmx_sum_rows(be.ptr(v), rows, cols, be.ptr(out), be.ptr(ws), be.stream()), "gmx_sum_rows")
    return out.reshape(x.shape[:-1])


def categorical_rows(key: Key, logits: torch.Tensor) -> torch.Tensor:
    """One Gumbel-max index per row of logits[..., cols] (gmx_categorical_rows)."""
    be = _lib.get()
    cols = logits.shape[-1]
    rows = int(np.prod(logits.shape[:-1], dtype=np.int64))
    if key.size != rows:
        raise ValueError(f"need one key per row: key batch {key.shape}, logits {tuple(logits.shape)}")
    x = logits.reshape(rows, cols).float().contiguous()
    kd = key.data()
    out = torch.empty((rows,), dtype=torch.int32, device=x.device)
    be.check(be.c.gmx_categorical_rows(be.ptr(kd), be.ptr(x), rows, cols, be.ptr(out), be.stream()),
             "gmx_categorical_rows")
    return out.reshape(logits.shape[:-1])
