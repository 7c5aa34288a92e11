"""A small `jax.numpy`-shaped namespace for model code.

Inside a `@gen` function these build the site program (traced values are
`tracer.Expr` / object arrays of them); outside a trace they act on torch
tensors / numpy arrays so inference scripts can post-process results
(`jnp.mean(chm["p"])`).  Existing GenJAX models `import jax.numpy as jnp`; on a
machine without JAX they `from genjax_amd import numpy as jnp` instead — same
names, array namespace supplied by this build (SURVEY.md §7 "Hard parts").
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import tracer as T
from .tracer import Expr, is_symbolic

pi = math.pi
e = math.e
inf = float("inf")
nan = float("nan")
float32 = np.float32
int32 = np.int32
bool_ = np.bool_
newaxis = None


def _gather(table, idx):
    """`means[zs]` with `means` a 1-D table in memory and `zs` a VECTOR of traced indices (categorical draws, an integer
    table, a per-particle integer vector): element j is one table read at zs[j] — lazily for a long index vector (a
    vector site over the result then loops), else one read per element.  None when `idx` is no such vector."""
    if table.ndim != 1 or not T.is_tracing():
        return None
    if isinstance(idx, T.LazyVec) or (T._long_vector(idx) and getattr(idx, "dtype", None) is not None
                                      and (idx.dtype == object or idx.dtype.kind in "iu")):
        n = T._long_vector(idx)
        return T.LazyVec(n, lambda i: table[T.lift(T._elem(idx, i))], parts=(table, idx))
    if isinstance(idx, np.ndarray) and idx.dtype == object and idx.ndim >= 1:
        out = np.empty(idx.shape, dtype=object)
        for pos in np.ndindex(idx.shape):
            out[pos] = table[idx[pos]]
        return T.sym_array(out)
    return None


class TableArray(np.ndarray):
    """A constant array closed over by a model; indexing it with a traced
    integer becomes an OP_LDTAB lookup (e.g. `means[idx]`)."""

    @property
    def _lazy_ok(self):
        return self.ndim == 1 and self.dtype != object

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def __getitem__(self, idx):
        if not isinstance(idx, (Expr, int, slice, np.integer)) and self.dtype != object:
            got = _gather(self, idx)
            if got is not None:
                return got
        if isinstance(idx, Expr):
            g = T.current_graph()
            dt = "f32" if self.dtype.kind == "f" else ("bool" if self.dtype.kind == "b" else "i32")
            if self.ndim == 1:
                slot = table_slot(g, np.asarray(self))
                return Expr(g.add("LDTAB", (T.as_int(idx).node,), dtype=dt, slot=slot))
            # row `idx` of a [T, *event] table: element e of the row is entry idx * E + e of the flattened table
            E = int(np.prod(self.shape[1:]))
            slot = table_slot(g, np.ascontiguousarray(np.asarray(self)).reshape(-1))
            base = (T.as_int(idx) * E).node
            row = np.empty(self.shape[1:], dtype=object)
            for e, ix in enumerate(np.ndindex(self.shape[1:])):
                row[ix] = Expr(g.add("LDTAB", (base,), imm=e, dtype=dt, slot=slot))
            if E > 16 and self.ndim == 2:      # a long row stays indexable at run time (RuntimeTable: same LDTAB reads)
                out = row.view(RuntimeTable)
                out._slot, out._dt, out._base, out._dyn = slot, dt, 0, base
                return out
            return row
        r = np.ndarray.__getitem__(self, idx)
        if isinstance(r, np.ndarray) and r.ndim == 0:
            return r.item()
        return r

    # arithmetic with a DEVICE tensor (a model running site by site mixes host tables with device values: `jnp.array(mu) + x`
    # where x came from a site): the table moves to the tensor's device, f32 / i32 as a launch would read it
    def _with_tensor(self, other, op, swap=False):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(self).astype(np.float32 if self.dtype.kind == "f" else
                                                                            (np.int32 if self.dtype.kind in "iu" else self.dtype))))
        t = t.to(other.device)
        return op(other, t) if swap else op(t, other)

    def __add__(self, o): return self._with_tensor(o, lambda a, b: a + b) if isinstance(o, torch.Tensor) else np.ndarray.__add__(self, o)
    def __radd__(self, o): return self._with_tensor(o, lambda a, b: a + b, True) if isinstance(o, torch.Tensor) else np.ndarray.__radd__(self, o)
    def __sub__(self, o): return self._with_tensor(o, lambda a, b: a - b) if isinstance(o, torch.Tensor) else np.ndarray.__sub__(self, o)
    def __rsub__(self, o): return self._with_tensor(o, lambda a, b: a - b, True) if isinstance(o, torch.Tensor) else np.ndarray.__rsub__(self, o)
    def __mul__(self, o): return self._with_tensor(o, lambda a, b: a * b) if isinstance(o, torch.Tensor) else np.ndarray.__mul__(self, o)
    def __rmul__(self, o): return self._with_tensor(o, lambda a, b: a * b, True) if isinstance(o, torch.Tensor) else np.ndarray.__rmul__(self, o)
    def __truediv__(self, o): return self._with_tensor(o, lambda a, b: a / b) if isinstance(o, torch.Tensor) else np.ndarray.__truediv__(self, o)
    def __rtruediv__(self, o): return self._with_tensor(o, lambda a, b: a / b, True) if isinstance(o, torch.Tensor) else np.ndarray.__rtruediv__(self, o)


def _lazy_ufunc(self, ufunc, method, inputs, kwargs):
    """`__array_ufunc__` of the table types: arithmetic on a LONG table stays lazy (tracer.LazyVec) instead of building
    one node per element; anything else is numpy's own loop over PLAIN arrays (a computed array is not a table any more)"""
    if method == "__call__" and not kwargs and T.is_tracing():
        f = _LAZY_UFUNCS.get(ufunc)
        if f is not None and T.lazy_length(inputs):
            return f(*inputs)
    plain = [x.view(np.ndarray) if isinstance(x, np.ndarray) else x for x in inputs]
    if "out" in kwargs:
        kwargs = dict(kwargs, out=tuple(o.view(np.ndarray) if isinstance(o, np.ndarray) else o for o in kwargs["out"]))
    return getattr(ufunc, method)(*plain, **kwargs)


_LAZY_UFUNCS = {
    np.add: lambda a, b: T._arith("ADD", "IADD", a, b), np.subtract: lambda a, b: T._arith("SUB", "ISUB", a, b),
    np.multiply: lambda a, b: T._arith("MUL", "IMUL", a, b), np.true_divide: lambda a, b: T._fbin("DIV", a, b),
    np.negative: lambda a: T.lazy_apply(lambda x: -T.lift(x), a), np.power: lambda a, b: T.power(a, b),
    np.less: lambda a, b: T._cmp("FLT", "ILT", a, b), np.less_equal: lambda a, b: T._cmp("FLE", "ILE", a, b),
    np.greater: lambda a, b: T._cmp("FGT", "IGT", a, b), np.greater_equal: lambda a, b: T._cmp("FGE", "IGE", a, b),
}


class RuntimeTable(np.ndarray):
    """A launch-uniform device vector (an argument such as cluster means): an object array of
    OP_LDTAB reads at constant indices (dead ones are eliminated), so every vector operation
    works on it; indexing with a traced integer is one OP_LDTAB at a register index."""

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return _lazy_ufunc(self, ufunc, method, inputs, kwargs)

    @property
    def _lazy_ok(self):
        return self._slot is not None and self.ndim == 1

    @classmethod
    def make(cls, g, slot, dt, n):
        """n: the length of a vector, or the shape (T, *event) of a table of rows (flattened row-major in memory)"""
        shape = (int(n),) if isinstance(n, (int, np.integer)) else tuple(int(x) for x in n)
        zero = g.const_i32(0)
        arr = np.empty(shape, dtype=object)
        for k, ix in enumerate(np.ndindex(shape)):
            arr[ix] = Expr(g.add("LDTAB", (zero,), imm=k, dtype=dt, slot=slot))
        out = arr.view(cls)
        out._slot, out._dt, out._base = slot, dt, 0
        return out

    def __array_finalize__(self, obj):
        self._slot = getattr(obj, "_slot", None)
        self._dt = getattr(obj, "_dt", None)
        self._base = getattr(obj, "_base", 0)        # entry of the flattened table this (row-major) view starts at
        self._dyn = getattr(obj, "_dyn", None)       # + a run-time offset (a row picked by a loop's iteration number)
        self._leaf = getattr(obj, "_leaf", None)     # the launch leaf this table is (Tracing.sym_leaf)
        self._picks = None                           # the rows picked so far (ints / "loop"); None: not a tracked row

    def passthrough(self):
        """a LONG row of a launch-uniform table that is a launch leaf — the values of a plate's / scan's vector site given
        as one [n, m] table — as the stacked output it already is: recorded by its origin, never copied per particle
        (the values stay launch-uniform in the trace, as a top-level site's do)"""
        from .engine import StepOutput
        if self._leaf is None or self._picks is None or self.ndim != 1:
            return None
        static = tuple(int(r) for r in self._picks if r != "loop")           # (this module's `any` / `sum` are jnp's)
        if "loop" in self._picks[:len(static)]:
            # a row picked statically BELOW a looped axis (an unrolled plate inside a loop): its elements over the outer
            # iterations are a strided slice of the leaf — the enclosing plate runs as a loop instead (Vmap.trace_call
            # tries the unrolled form first and takes the loop form on this)
            raise NotImplementedError("a row of a launch-uniform table picked statically below a looped axis")
        n_loop = len(self._picks) - len(static)
        origin = ("leafrow", self._leaf, static, n_loop) if static else ("leaf", self._leaf)
        return StepOutput(origin, int(self.shape[0]), 1 + n_loop)

    def __getitem__(self, idx):
        if not isinstance(idx, (Expr, int, slice, np.integer)) and self._slot is not None:
            got = _gather(self, idx)
            if got is not None:
                return got
        base = int(self._base or 0)
        dyn = getattr(self, "_dyn", None)
        if isinstance(idx, Expr) and self._slot is not None and self.ndim == 1:
            g = T.current_graph()
            if dyn is not None:                                  # element idx of a row that was itself picked at run time
                return Expr(g.add("LDTAB", ((Expr(dyn) + T.as_int(idx)).node,), imm=base, dtype=self._dt, slot=self._slot))
            if idx.node.op == "CONST":                           # static after all: the shared element read
                return Expr(g.add("LDTAB", (g.const_i32(0),), imm=idx.node.imm + base, dtype=self._dt, slot=self._slot))
            return Expr(g.add("LDTAB", (T.as_int(idx).node,), imm=base, dtype=self._dt, slot=self._slot))
        if isinstance(idx, Expr) and self._slot is not None and self.ndim >= 2:
            g = T.current_graph()                                # row `idx` of a table of rows
            E = int(np.prod(self.shape[1:]))
            rb = T.as_int(idx) * E
            if dyn is not None:
                rb = rb + Expr(dyn)
            rbase = rb.node
            row = np.empty(self.shape[1:], dtype=object)
            for e, ix in enumerate(np.ndindex(self.shape[1:])):
                row[ix] = Expr(g.add("LDTAB", (rbase,), imm=base + e, dtype=self._dt, slot=self._slot))
            if E > 16:             # a LONG row: it stays a table, so that a loop inside can pick its elements at run time
                out = row.view(RuntimeTable)
                out._slot, out._dt, out._base, out._dyn = self._slot, self._dt, base, rbase
                out._leaf = self._leaf
                out._picks = ((self._picks or ()) + ("loop",)) if (idx.node.op == "LDT" and self._leaf is not None) else None
                return out
            return row
        r = np.ndarray.__getitem__(self, idx)
        if isinstance(r, RuntimeTable) and self.ndim >= 2 and isinstance(idx, (int, np.integer)):
            r._base = base + int(idx) * int(np.prod(self.shape[1:]))   # a static row stays a table: its elements can
            r._leaf = self._leaf                                       # still be picked at run time (a plate of scans)
            r._picks = ((self._picks or ()) + (int(idx) % self.shape[0],)) if (self._leaf is not None and self._picks is not None) else None
            return r
        if isinstance(r, RuntimeTable) and self.ndim >= 2 and not isinstance(idx, slice):
            return np.asarray(r, dtype=object)                   # any other static pick: plain expressions
        if isinstance(r, RuntimeTable) and isinstance(idx, slice) and self.ndim == 1:
            start = idx.indices(self.shape[0])[0] if (idx.step in (None, 1)) else None
            if start is None:
                return np.asarray(r, dtype=object)
            r._base = base + start
        return r


def runtime_table_slot(g) -> int:
    tabs = g.__dict__.setdefault("tables", [])
    tabs.append(None)
    g.n_tab = len(tabs)
    return len(tabs) - 1


def table_slot(g, arr) -> int:
    tabs = g.__dict__.setdefault("tables", [])
    for s, t in enumerate(tabs):
        if t is None:
            continue
        if t is arr or (t.shape == arr.shape and t.dtype == arr.dtype and np.array_equal(t, arr)):
            return s
    tabs.append(arr)
    g.n_tab = len(tabs)
    return len(tabs) - 1


def _is_torch(x):
    return isinstance(x, torch.Tensor)


def array(x, dtype=None):
    from .random import Key
    if isinstance(x, Key):
        return x
    if isinstance(x, (list, tuple)) and len(x) and builtins_all(isinstance(k_, Key) for k_ in x):
        # `key, *sub_keys = jax.random.split(key, N + 1); sub_keys = jnp.array(sub_keys)` (importance_sampling.ipynb c4,
        # custom_proposal.ipynb c3): a list of keys is a batch of keys
        from .random import stack_keys
        return stack_keys(x)
    if is_symbolic(x):
        return T.sym_array(np.asarray(x, dtype=object))
    if _is_torch(x):
        return x if dtype is None else x.to(_torch_dtype(dtype))
    a = np.asarray(x)
    if dtype is not None:
        a = a.astype(dtype)
    elif a.dtype == np.float64:
        a = a.astype(np.float32)
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return TableArray(a) if T.is_tracing() or a.ndim > 0 else a


asarray = array


def _torch_dtype(dt):
    dt = np.dtype(dt)
    return {"f": torch.float32, "i": torch.int32, "u": torch.int32, "b": torch.bool}[dt.kind]


def zeros(shape, dtype=np.float32):
    return TableArray(np.zeros(shape, dtype=dtype))


def ones(shape, dtype=np.float32):
    return TableArray(np.ones(shape, dtype=dtype))


def full(shape, v, dtype=None):
    if is_symbolic(v):
        out = np.empty(shape, dtype=object)
        out[...] = v
        return out
    return TableArray(np.full(shape, v, dtype=dtype or (np.float32 if isinstance(v, float) else None)))


def arange(*a, dtype=np.int32):
    return TableArray(np.arange(*a, dtype=dtype))


def _backend_or_none():
    from . import _lib
    if _lib._backend is None and not torch.cuda.is_available():
        return None
    try:
        return _lib.get()
    except _lib.GenmiError:
        return None


def _device_elementwise(name, x):
    """An eager call on a float tensor goes through the same fixed-sequence device math a traced model uses
    (`engine.elementwise`: one launch of a one-instruction site program), so `jnp.exp(t)` has the same bits outside
    a model as inside one.  Needs a backend (the HIP library, or the tests' CPU mirror); without one — plain CPU
    torch, no GPU — the call falls back to torch's own function (different last bits; nothing on the product path
    runs there)."""
    from . import _lib, engine
    if _lib._backend is None and not torch.cuda.is_available():
        return None
    try:
        be = _lib.get()
    except _lib.GenmiError:
        return None
    if x.device.type != be.device.type:
        return None
    y = engine.elementwise(globals()["_ew_" + name], x.reshape(-1) if x.ndim != 1 else x)
    return y.reshape(x.shape)


def _dispatch(name, op, torch_fn, np_fn):
    sym = T.unary(op)

    def f(x):
        if is_symbolic(x):
            return sym(x)
        if T.is_tracing() and isinstance(x, (int, float, np.number)):
            return sym(x)
        if _is_torch(x):
            if x.dtype == torch.float32 and x.numel() > 0:
                y = _device_elementwise(name, x)
                if y is not None:
                    return y
            return torch_fn(x)
        return np_fn(np.asarray(x, dtype=np.float32) if not isinstance(x, np.ndarray) else x)
    f.__name__ = name
    # a closure-free twin per op: engine.elementwise caches the traced program on the function's code object
    exec(f"def _ew_{name}(x):\n    return {name}(x)\n", globals())
    return f


exp = _dispatch("exp", "EXP", torch.exp, np.exp)
log = _dispatch("log", "LOG", torch.log, np.log)
log1p = _dispatch("log1p", "LOG1P", torch.log1p, np.log1p)
sqrt = _dispatch("sqrt", "SQRT", torch.sqrt, np.sqrt)
sin = _dispatch("sin", "SIN", torch.sin, np.sin)
cos = _dispatch("cos", "COS", torch.cos, np.cos)
tanh = _dispatch("tanh", "TANH", torch.tanh, np.tanh)
floor = _dispatch("floor", "FLOOR", torch.floor, np.floor)
ceil = _dispatch("ceil", "CEIL", torch.ceil, np.ceil)
square = _dispatch("square", "SQUARE", torch.square, np.square)
abs = _dispatch("abs", "ABS", torch.abs, np.abs)        # noqa: A001
absolute = abs
sigmoid = _dispatch("sigmoid", "SIGMOID", torch.sigmoid, lambda x: 1.0 / (1.0 + np.exp(-x)))
softplus = _dispatch("softplus", "SOFTPLUS", torch.nn.functional.softplus, lambda x: np.logaddexp(0.0, x))
lgamma = _dispatch("lgamma", "LGAMMA", torch.lgamma, lambda x: np.vectorize(math.lgamma)(x))


def where(c, a, b):
    if is_symbolic(c) or is_symbolic(a) or is_symbolic(b):
        return T.where(c, a, b)
    if _is_torch(c) or _is_torch(a) or _is_torch(b):
        dev = next(t.device for t in (c, a, b) if _is_torch(t))
        tt = lambda v: v if _is_torch(v) else torch.as_tensor(v, device=dev)
        cond = tt(c)
        if cond.dtype != torch.bool:          # jnp.where takes any condition: non-zero holds (a bernoulli's 0 / 1)
            cond = cond != 0
        return torch.where(cond, tt(a), tt(b))
    return np.where(c, a, b)


def power(a, b):
    if is_symbolic(a) or is_symbolic(b):
        return T.power(a, b)
    if _is_torch(a) or _is_torch(b):
        return torch.pow(a, b)
    return np.power(a, b)


def minimum(a, b):
    if is_symbolic(a) or is_symbolic(b):
        return T.minimum(a, b)
    if _is_torch(a) or _is_torch(b):
        return torch.minimum(torch.as_tensor(a), torch.as_tensor(b))
    return np.minimum(a, b)


def maximum(a, b):
    if is_symbolic(a) or is_symbolic(b):
        return T.maximum(a, b)
    if _is_torch(a) or _is_torch(b):
        return torch.maximum(torch.as_tensor(a), torch.as_tensor(b))
    return np.maximum(a, b)


def clip(x, lo, hi):
    return minimum(maximum(x, lo), hi)


def logical_and(a, b):
    return a & b


def logical_or(a, b):
    return a | b


def logical_not(a):
    if isinstance(a, Expr):
        return ~a
    if isinstance(a, np.ndarray) and a.dtype == object:
        return np.vectorize(lambda v: ~v, otypes=[object])(a)
    return ~a if _is_torch(a) else np.logical_not(a)


def _sum_loop(x, n):
    """`jnp.sum` of a LONG vector that is readable at a run-time index (the values a large plate / a long scan / a long
    vector-valued site left in memory, a launch-uniform table, a lazy expression of them): ONE counted loop adding
    element t to a loop-carried sum — element order, the order the unrolled form adds in (oracle: sum_vector), so a plate
    of 1000 elements costs two instructions per element instead of 1000 registers (ref: vmap.py:180-191 hands back plain
    stacked arrays; `jnp.sum` over them is what 3_speed_gains.ipynb c4 does)."""
    g = T.current_graph()
    gvecs, seen_ = [], set()

    def find(p_):
        if isinstance(p_, T.GradVec):
            if id(p_) not in seen_:
                seen_.add(id(p_))
                gvecs.append(p_)
        elif isinstance(p_, T.LazyVec):
            for q_ in p_.parts:
                find(q_)
    find(x)
    probe = [len(gv.reads) for gv in gvecs]
    dt = getattr(x, "_dt", None)
    if dt is None:
        try:
            dt = T.lift(T._elem(x, 0)).dtype          # (dead code afterwards: one element's expression, never stored)
        except NotImplementedError:
            dt = "f32"
    for gv, k_ in zip(gvecs, probe):                    # (the probe is no read of the model's)
        del gv.reads[k_:]
    isf = dt == "f32"
    acc = g.loop_var(g.const_f32(0.0) if isf else g.const_i32(0))
    # a long vector an HMC move differentiates with respect to (tracer.GradVec), summed: d (element t) / d v_t is stored
    # beside the sum — this loop's contribution to the vector's gradient, scaled later by the adjoint the sum reaches the
    # model score with (`normal(jnp.sum(theta), 1) @ "y"`, `jnp.mean(theta ** 2)`; static._vector_site_loop does the same
    # for vector-valued sites)
    gvecs = [gv for gv in gvecs if gv.n == n and isf and getattr(g, "_tracing", None) is not None]
    marks = [len(gv.reads) for gv in gvecs]
    pend = []
    g.loop_begin(n)
    t = Expr(g.add("LDT", dtype="i32"))
    v = T.lift(T._elem(x, t))
    v = T.as_float(v) if isf else T.as_int(v)
    for gv, mark in zip(gvecs, marks):
        mine, seen = [], set()
        for i_, v_ in gv.reads[mark:]:
            if isinstance(i_, Expr) and i_.node is t.node:
                gv.consumed += 1
                if v_.node.idx not in seen:
                    seen.add(v_.node.idx)
                    mine.append(v_)
        if mine:
            from .autodiff import grad as _grad
            parts = _grad(v, mine)
            tot = parts[0]
            for pt in parts[1:]:
                tot = tot + pt
            pend.append((gv, g._tracing.store_step(tot, n)))
    g.set_vars([(acc, (Expr(acc) + v).node)])
    g.loop_end()
    for gv, o_ in pend:
        gv.contribs.append((acc, g._tracing.alias_step_input(o_, "f32", n)))
    return Expr(acc)


def sum(x, axis=None):        # noqa: A001
    if T.is_tracing() and axis in (None, 0, -1):
        n = T._long_vector(x)
        if n and len(T.current_graph().loop_counts) < 3 and not getattr(x, "vector_site", False):
            return _sum_loop(x, n)
    if is_symbolic(x):
        a = np.asarray(x, dtype=object)
        if axis is None:
            flat = a.reshape(-1)
            acc = flat[0]
            for v in flat[1:]:
                acc = acc + v
            return acc
        a = np.moveaxis(a, axis, -1)
        out = np.empty(a.shape[:-1], dtype=object)
        for idx in np.ndindex(out.shape):
            acc = a[idx][0]
            for v in a[idx][1:]:
                acc = acc + v
            out[idx] = acc
        return out if out.ndim else out.item()
    if _is_torch(x):
        if x.dtype == torch.float32 and x.dim() >= 1 and x.numel() > 0 and (axis is None and x.dim() == 1 or axis in (-1, x.dim() - 1)):
            # a concrete float vector (a plate's values in a model that runs site by site, sitewise.py): the sum has a
            # DEFINED order, the plate score's — element order below VMAP_LAUNCH_MIN items, gmx_sum_rows' fixed tree from
            # there on (oracle: sum_vector) — and runs in this build's kernels (jnp.sum fixes no order: vmap.py:214-216).
            # f32 only: a float64 tensor keeps its precision and dtype through torch.sum
            from . import engine
            from .combinators import VMAP_LAUNCH_MIN
            be = _backend_or_none()
            if be is not None and x.device == be.device:
                if x.shape[-1] >= VMAP_LAUNCH_MIN:
                    return engine.sum_rows(x)
                return engine.sum_rows_inorder(x.reshape(-1, x.shape[-1])).reshape(x.shape[:-1])
        return torch.sum(x) if axis is None else torch.sum(x, dim=axis)
    return np.sum(x, axis=axis)


def mean(x, axis=None):
    if T.is_tracing() and axis in (None, 0, -1) and T._long_vector(x):
        return sum(x, axis) / float(T._long_vector(x))
    if is_symbolic(x):
        a = np.asarray(x, dtype=object)
        n = a.size if axis is None else a.shape[axis]
        return sum(a, axis) / float(n)
    if _is_torch(x):
        x = x.float()
        return torch.mean(x) if axis is None else torch.mean(x, dim=axis)
    return np.mean(x, axis=axis)


def var(x, axis=None):
    """population variance (ddof = 0, jnp's default): mean((x - mean(x))^2)"""
    if is_symbolic(x) or (T.is_tracing() and T._long_vector(x)):
        m = mean(x, axis)
        d = x - m if axis is None else x - np.expand_dims(np.asarray(m, dtype=object), axis)
        return mean(d * d, axis)
    if _is_torch(x):
        return torch.var(x.float(), unbiased=False) if axis is None else torch.var(x.float(), dim=axis, unbiased=False)
    return np.var(x, axis=axis)


def std(x, axis=None):
    if is_symbolic(x) or (T.is_tracing() and T._long_vector(x)):
        return sqrt(var(x, axis))
    if _is_torch(x):
        return torch.std(x.float(), unbiased=False) if axis is None else torch.std(x.float(), dim=axis, unbiased=False)
    return np.std(x, axis=axis)


def stack(xs, axis=0):
    if is_symbolic(xs):
        return T.sym_array(np.stack([np.asarray(v, dtype=object) for v in xs], axis=axis))
    if builtins_any(_is_torch(v) for v in xs):
        return torch.stack(list(xs), dim=axis)
    return np.stack(xs, axis=axis)


def builtins_all(it):
    for v in it:
        if not v:
            return False
    return True


def builtins_any(it):
    for v in it:
        if v:
            return True
    return False


def logsumexp(x, axis=-1):
    """jax.scipy.special.logsumexp over a traced vector: max, then
    log(sum(exp(x - max))) + max, accumulated in index order."""
    if is_symbolic(x):
        a = np.asarray(x, dtype=object)
        if a.ndim != 1:
            raise NotImplementedError("traced logsumexp over ndim != 1")
        m = a[0]
        for v in a[1:]:
            m = T.maximum(m, v)
        acc = None
        for v in a:
            t = exp(v - m)
            acc = t if acc is None else acc + t
        return log(acc) + m
    if _is_torch(x):
        return torch.logsumexp(x, dim=axis)
    from scipy.special import logsumexp as _l
    return _l(x, axis=axis)


class _Lax:
    """`jax.lax` subset: deterministic control flow on traced values."""

    @staticmethod
    def cond(pred, true_fun, false_fun, *operands):
        a = true_fun(*operands)
        b = false_fun(*operands)
        if isinstance(pred, (bool, np.bool_)):
            return a if pred else b
        return _tree_where(pred, a, b)

    @staticmethod
    def select(pred, a, b):
        return where(pred, a, b)


def _tree_where(pred, a, b):
    from .core.choice_map import ChoiceMap
    from .core.generative import Trace
    if isinstance(a, Trace) and not T.is_tracing():
        # `jax.lax.cond(accept, lambda: new_trace, lambda: trace)` (mcmc.ipynb c8): traces are pytrees — a select per leaf
        from .combinators import _trace_leaf_zip
        from .engine import materialize

        def pick(old, new):
            old, new = materialize(old), materialize(new)
            if not _is_torch(new) and not _is_torch(old):
                return new if (isinstance(pred, (bool, np.bool_)) and pred) else (old if isinstance(pred, (bool, np.bool_)) else new)
            t = new if _is_torch(new) else old
            new = new if _is_torch(new) else torch.as_tensor(new, device=t.device).to(t.dtype).expand(t.shape)
            old = old if _is_torch(old) else torch.as_tensor(old, device=t.device).to(t.dtype).expand(t.shape)
            c = pred if _is_torch(pred) else torch.as_tensor(pred, device=t.device)
            c = c.reshape(tuple(c.shape) + (1,) * (new.dim() - c.dim()))
            return torch.where(c if c.dtype == torch.bool else c != 0, new, old.to(new.dtype))
        return _trace_leaf_zip(b, a, pick, args=a.get_args())
    if isinstance(a, ChoiceMap) and isinstance(b, ChoiceMap):
        # (3_speed_gains.ipynb c15 returns one of two choice maps with the same addresses)
        out = ChoiceMap.empty()
        for addr in a.addresses():
            va = a[addr] if addr else a.get_value()
            vb = b[addr] if addr else b.get_value()
            out = out.set(addr, where(pred, va, vb)) if addr else ChoiceMap.choice(where(pred, va, vb))
        return out
    if isinstance(a, (tuple, list)):
        return type(a)(_tree_where(pred, x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return {k: _tree_where(pred, a[k], b[k]) for k in a}
    if a is None:
        return None
    return where(pred, a, b)


lax = _Lax()


# ---------------------------------------------------------------------------
# the rest of the everyday jax.numpy surface, as compositions of the ops above (so they trace
# into site programs; on concrete numpy / torch values they defer to the library)
# ---------------------------------------------------------------------------
float32, int32, bool_ = np.float32, np.int32, np.bool_
pi, inf, nan, e = math.pi, math.inf, math.nan, math.e
_LN2, _LN10 = math.log(2.0), math.log(10.0)
round = _dispatch("round", "ROUND", torch.round, np.rint)          # noqa: A001  (round half to even, like jnp)
rint = round


def _lib_of(*xs):
    return torch if builtins_any(_is_torch(x) for x in xs) else np


def add(a, b): return a + b
def subtract(a, b): return a - b
def multiply(a, b): return a * b
def divide(a, b): return a / b
def negative(a): return -a


true_divide = divide


def expm1(x):
    """exp(x) - 1.  Traced: the plain composition (no fused expm1 op in the site-program ISA)."""
    return exp(x) - 1.0 if (is_symbolic(x) or T.is_tracing()) else _lib_of(x).expm1(x)


def log2(x): return log(x) * (1.0 / _LN2) if is_symbolic(x) else _lib_of(x).log2(x)
def log10(x): return log(x) * (1.0 / _LN10) if is_symbolic(x) else _lib_of(x).log10(x)
def exp2(x): return exp(x * _LN2) if is_symbolic(x) else _lib_of(x).exp2(x)
def tan(x): return sin(x) / cos(x) if is_symbolic(x) else _lib_of(x).tan(x)
def sinh(x): return (exp(x) - exp(-x)) * 0.5 if is_symbolic(x) else _lib_of(x).sinh(x)
def cosh(x): return (exp(x) + exp(-x)) * 0.5 if is_symbolic(x) else _lib_of(x).cosh(x)


def logaddexp(a, b):
    if is_symbolic(a) or is_symbolic(b):
        m = maximum(a, b)
        return m + log1p(exp(-abs(a - b)))
    return _lib_of(a, b).logaddexp(torch.as_tensor(a), torch.as_tensor(b)) if _lib_of(a, b) is torch else np.logaddexp(a, b)


def sign(x):
    if is_symbolic(x):
        return where(x > 0.0, 1.0, where(x < 0.0, -1.0, 0.0))
    return _lib_of(x).sign(x)


def isnan(x): return (x != x) if is_symbolic(x) else _lib_of(x).isnan(x)
def isfinite(x): return ((x == x) & (abs(x) < inf)) if is_symbolic(x) else _lib_of(x).isfinite(x)


def nan_to_num(x, nan=0.0):                                           # noqa: A002
    return where(isnan(x), nan, x) if is_symbolic(x) else _lib_of(x).nan_to_num(x, nan=nan)


def mod(a, b):
    """Python / jnp convention: the result takes the sign of the divisor."""
    if is_symbolic(a) or is_symbolic(b):
        return a - floor(a / b) * b
    return _lib_of(a, b).remainder(a, b)


remainder = mod


def _reduce(x, axis, step):
    a = np.asarray(x, dtype=object)
    if axis is None:
        flat = a.reshape(-1)
        acc = flat[0]
        for v in flat[1:]:
            acc = step(acc, v)
        return acc
    a = np.moveaxis(a, axis, -1)
    out = np.empty(a.shape[:-1], dtype=object)
    for idx in np.ndindex(out.shape):
        acc = a[idx][0]
        for v in a[idx][1:]:
            acc = step(acc, v)
        out[idx] = acc
    return out if out.ndim else out.item()


def max(x, axis=None):                                                # noqa: A001
    if is_symbolic(x):
        return _reduce(x, axis, maximum)
    return (torch.amax(x) if axis is None else torch.amax(x, dim=axis)) if _is_torch(x) else np.max(x, axis=axis)


def min(x, axis=None):                                                # noqa: A001
    if is_symbolic(x):
        return _reduce(x, axis, minimum)
    return (torch.amin(x) if axis is None else torch.amin(x, dim=axis)) if _is_torch(x) else np.min(x, axis=axis)


amax, amin = max, min


def prod(x, axis=None):
    if is_symbolic(x):
        return _reduce(x, axis, lambda p, q: p * q)
    return (torch.prod(x) if axis is None else torch.prod(x, dim=axis)) if _is_torch(x) else np.prod(x, axis=axis)


def cumsum(x, axis=0):
    if is_symbolic(x):
        a = np.moveaxis(np.asarray(x, dtype=object), axis, 0)
        out = np.empty(a.shape, dtype=object)
        acc = None
        for j in range(a.shape[0]):
            acc = a[j] if acc is None else acc + a[j]
            out[j] = acc
        return np.moveaxis(out, 0, axis)
    return torch.cumsum(x, dim=axis) if _is_torch(x) else np.cumsum(x, axis=axis)


def dot(a, b):
    if is_symbolic(a) or is_symbolic(b):
        if np.ndim(a) != 1 or np.ndim(b) != 1:
            if np.ndim(a) in (1, 2) and np.ndim(b) in (1, 2):
                return a @ b                      # (jnp.dot of a matrix and a vector / matrix is their matmul)
            raise NotImplementedError("traced dot: vectors and matrices only")
        a, b = np.asarray(a, dtype=object), np.asarray(b, dtype=object)
        return sum(a * b)
    return _lib_of(a, b).dot(a, b)


def var(x, axis=None):
    m = mean(x, axis)
    if axis is not None and is_symbolic(x):
        raise NotImplementedError("traced var along an axis")
    return mean(square(x - m), axis)


def std(x, axis=None):
    return sqrt(var(x, axis))


def concatenate(xs, axis=0):
    if builtins_any(is_symbolic(v) for v in xs):
        return T.sym_array(np.concatenate([np.atleast_1d(np.asarray(v, dtype=object)) for v in xs], axis=axis))
    if builtins_any(_is_torch(v) for v in xs):
        return torch.cat([torch.as_tensor(v) for v in xs], dim=axis)
    return np.concatenate(xs, axis=axis)


def reshape(x, shape):
    return np.asarray(x, dtype=object).reshape(shape) if is_symbolic(x) else x.reshape(shape)


def tile(x, reps):
    return np.tile(np.asarray(x, dtype=object), reps) if is_symbolic(x) else (x.repeat(reps) if _is_torch(x) else np.tile(x, reps))


def zeros_like(x):
    return full(np.shape(x), 0.0) if is_symbolic(x) else (torch.zeros_like(x) if _is_torch(x) else np.zeros_like(x))


def ones_like(x):
    return full(np.shape(x), 1.0) if is_symbolic(x) else (torch.ones_like(x) if _is_torch(x) else np.ones_like(x))


def eye(n, m=None, dtype=np.float32):
    return TableArray(np.eye(n, m, dtype=dtype))


def broadcast_to(x, shape):
    if is_symbolic(x):
        return np.broadcast_to(np.asarray(x, dtype=object), shape)
    if _is_torch(x):
        return torch.broadcast_to(x, tuple(shape))
    return TableArray(np.ascontiguousarray(np.broadcast_to(np.asarray(x), shape)))


def repeat(x, repeats, axis=None):
    if is_symbolic(x):
        return np.repeat(np.asarray(x, dtype=object), repeats, axis=axis)
    if _is_torch(x):
        return torch.repeat_interleave(x, repeats, dim=axis)
    a = np.repeat(np.asarray(x), repeats, axis=axis)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return TableArray(a)


def matmul(a, b):
    return a @ b


def transpose(x, axes=None):
    return np.transpose(np.asarray(x, dtype=object), axes) if is_symbolic(x) else (
        x.permute(*axes) if (_is_torch(x) and axes is not None) else (x.t() if _is_torch(x) and x.ndim == 2 else np.transpose(x, axes)))


def array_equal(a, b):
    return bool(np.array_equal(np.asarray(a.detach().cpu() if _is_torch(a) else a), np.asarray(b.detach().cpu() if _is_torch(b) else b)))


def allclose(a, b, rtol=1e-5, atol=1e-8):
    return bool(np.allclose(np.asarray(a.detach().cpu() if _is_torch(a) else a), np.asarray(b.detach().cpu() if _is_torch(b) else b),
                            rtol=rtol, atol=atol))


def all(x, axis=None):                                                # noqa: A001
    """concrete arrays only (a truth value of launch values belongs to the host, not to a site program)"""
    a = np.asarray(x.detach().cpu() if _is_torch(x) else x)
    return bool(a.all()) if axis is None else a.all(axis=axis)


def any(x, axis=None):                                                # noqa: A001
    a = np.asarray(x.detach().cpu() if _is_torch(x) else x)
    return bool(a.any()) if axis is None else a.any(axis=axis)


def bincount(x, weights=None, minlength=0, length=None):
    """jnp.bincount (7_application_dirichlet_mixture_model.ipynb c10: `length=` fixes the output size under jit)"""
    n = int(length if length is not None else minlength)
    if _is_torch(x):
        out = torch.bincount(x.to(torch.int64).reshape(-1), weights=weights, minlength=n)
        return out[:n] if length is not None else out
    out = np.bincount(np.asarray(x).reshape(-1), weights=weights, minlength=n)
    return out[:n] if length is not None else out


def __getattr__(name):
    """Names this module does not define fall through to numpy (`jnp.meshgrid`, `jnp.ogrid`, `jnp.mask_indices`, ...: host-side
    array construction in notebooks) — they act on concrete arrays only, never on traced values."""
    if name.startswith("_"):
        raise AttributeError(name)
    try:
        return getattr(np, name)
    except AttributeError:
        raise AttributeError(f"genjax_amd.numpy has no attribute {name!r} (and numpy has none either)") from None


class _At:
    """`x.at[idx].set(v)` / `.add(v)` / `.multiply(v)` / `.get()`: jax's functional array update (4_index_request.ipynb c5:
    `trace.get_choices()["a"].at[IDX].set(value)`) — a new array, the operand untouched"""

    def __init__(self, x, idx=None):
        self._x, self._idx = x, idx

    def __getitem__(self, idx):
        return _At(self._x, idx)

    def _apply(self, v, fn):
        x = self._x
        if _is_torch(x):
            out = x.clone()
            vv = v if _is_torch(v) else torch.as_tensor(v, dtype=x.dtype, device=x.device)
            out[self._idx] = fn(out[self._idx], vv.to(out.dtype))
            return out
        a = np.array(np.asarray(x), copy=True)
        a[self._idx] = fn(a[self._idx], v.detach().cpu().numpy() if _is_torch(v) else v)
        return TableArray(a) if isinstance(x, TableArray) else a

    def set(self, v): return self._apply(v, lambda old, new: new)          # noqa: A003
    def add(self, v): return self._apply(v, lambda old, new: old + new)
    def multiply(self, v): return self._apply(v, lambda old, new: old * new)
    def get(self): return self._x[self._idx]


TableArray.at = property(lambda self: _At(self))
if not hasattr(torch.Tensor, "at"):
    # values a trace hands back are torch tensors on this stack: they take jax's `.at[...]` spelling too (a property added
    # to torch.Tensor at import — the one place this package touches torch's namespace; `GENMI_NO_TENSOR_AT=1` leaves it out)
    import os as _os
    if _os.environ.get("GENMI_NO_TENSOR_AT", "0") != "1":
        torch.Tensor.at = property(lambda self: _At(self))
