"""Symbolic values for tracing a `@gen` function's Python source.

An `Expr` wraps a program `Node`; Python operators and the functions in
`genjax_amd.numpy` build the site program.  Tensors with an event shape are
numpy object arrays of `Expr` (unrolled: event shapes on this path are small —
the eight schools, a handful of mixture components), so numpy's own
broadcasting rules apply to them.

dtype rules follow JAX's weak typing for the cases models use: Python floats
are f32, Python ints are i32, bools are bool; int (op) float -> float.
"""
from __future__ import annotations

import numpy as np

from .program import Graph, Node

_GRAPHS: list[Graph] = []


def current_graph() -> Graph:
    if not _GRAPHS:
        raise RuntimeError("symbolic value used outside of a trace")
    return _GRAPHS[-1]


class tracing:
    def __init__(self, graph: Graph):
        self.graph = graph

    def __enter__(self):
        _GRAPHS.append(self.graph)
        return self.graph

    def __exit__(self, *exc):
        _GRAPHS.pop()


def is_tracing() -> bool:
    return bool(_GRAPHS)


class Expr:
    __slots__ = ("node",)
    __array_priority__ = 1000          # numpy defers to our reflected operators

    def __init__(self, node: Node):
        self.node = node

    @property
    def dtype(self):
        return self.node.dtype

    shape = ()
    ndim = 0

    def __repr__(self):
        return f"Expr<{self.node.op}:{self.node.dtype}#{self.node.idx}>"

    def __bool__(self):
        raise TypeError(
            "the truth value of a traced value is not available while tracing a @gen function; "
            "use genjax_amd.numpy.where / lax.cond instead of Python `if`")

    def __hash__(self):
        return id(self)

    # arithmetic -------------------------------------------------------------
    def __add__(self, o): return _arith("ADD", "IADD", self, o)
    def __radd__(self, o): return _arith("ADD", "IADD", o, self)
    def __sub__(self, o): return _arith("SUB", "ISUB", self, o)
    def __rsub__(self, o): return _arith("SUB", "ISUB", o, self)
    def __mul__(self, o): return _arith("MUL", "IMUL", self, o)
    def __rmul__(self, o): return _arith("MUL", "IMUL", o, self)
    def __truediv__(self, o): return _fbin("DIV", self, o)
    def __rtruediv__(self, o): return _fbin("DIV", o, self)
    def __pow__(self, o): return power(self, o)
    def __rpow__(self, o): return power(o, self)
    def __neg__(self):
        if self.dtype == "f32":
            return _un("NEG", self)
        return Expr(current_graph().add("INEG", (as_int(self).node,), dtype="i32"))
    def __pos__(self): return self
    def __abs__(self): return _un("ABS", as_float(self))
    # comparisons ------------------------------------------------------------
    def __lt__(self, o): return _cmp("FLT", "ILT", self, o)
    def __le__(self, o): return _cmp("FLE", "ILE", self, o)
    def __gt__(self, o): return _cmp("FGT", "IGT", self, o)
    def __ge__(self, o): return _cmp("FGE", "IGE", self, o)
    def __eq__(self, o): return _cmp("FEQ", "IEQ", self, o)      # noqa: D105
    def __ne__(self, o): return _cmp("FNE", "INE", self, o)
    # logic --------------------------------------------------------------------
    def __and__(self, o): return _logic("AND", self, o)
    def __rand__(self, o): return _logic("AND", o, self)
    def __or__(self, o): return _logic("OR", self, o)
    def __ror__(self, o): return _logic("OR", o, self)
    def __xor__(self, o): return _logic("XOR", self, o)
    def __invert__(self): return Expr(current_graph().add("NOT", (self.node,), dtype="bool"))

    def astype(self, dt):
        dt = np.dtype(dt) if not isinstance(dt, str) or dt not in ("f32", "i32", "bool") else dt
        if dt in ("f32",) or (not isinstance(dt, str) and dt.kind == "f"):
            return as_float(self)
        if dt in ("i32",) or (not isinstance(dt, str) and dt.kind in "iu"):
            return as_int(self)
        return as_bool(self)


LAZY_MIN = 17          # arithmetic on a launch-uniform / per-particle VECTOR of at least this many elements stays lazy


class LazyVec:
    """An elementwise expression over a LONG vector (`a * xs + b` with `xs` a launch-uniform table or a per-particle
    [n, T] leaf of more than 16 elements), kept as a recipe instead of T unrolled copies: `at(i)` builds element i —
    `i` a Python int (the static element) or the iteration number of a counted loop (ONE table / step read at a
    run-time index).  A vector-valued SITE whose parameters or value are such vectors runs as a counted loop in the
    site program (static._vector_site_loop; TFP batch semantics, tensorflow_probability/__init__.py:52-62): element j
    draws with counter j from the one site key, the score is summed in element order (distribution.py:383-396 as the
    oracle states it).  Anything that needs the elements themselves (indexing, `jnp.sum`, a consumer that knows only
    object arrays) gets them on demand: iterating a LazyVec materialises it, as the unrolled form always did."""
    __array_ufunc__ = None         # numpy's binary operators defer to ours (an ndarray on the left included)
    __array_priority__ = 2000
    ndim = 1
    dtype = np.dtype(object)

    def __init__(self, n: int, fn, parts=()):
        self.n, self._fn, self.parts, self._full = int(n), fn, tuple(parts), None

    @property
    def shape(self):
        return (self.n,)

    def __len__(self):
        return self.n

    def at(self, i):
        if isinstance(i, (int, np.integer)):
            if not -self.n <= i < self.n:
                raise IndexError(i)
            i = int(i) % self.n
            if self._full is not None:
                return self._full[i]
        return self._fn(i)

    def materialize(self) -> np.ndarray:
        if self._full is None:
            out = np.empty((self.n,), dtype=object)
            for i in range(self.n):
                out[i] = lift(self._fn(i))
            self._full = out
        return self._full

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)) or (isinstance(idx, Expr) and idx.dtype in ("i32", "bool")):
            return self.at(idx)
        return self.materialize()[idx]

    def __iter__(self):
        return iter(self.materialize())

    def __array__(self, dtype=None, copy=None):
        return self.materialize()

    def reshape(self, *shape):
        return self.materialize().reshape(*shape)

    def dep_nodes(self) -> list:
        """the program nodes this vector is computed from (change propagation: static._nodes_of)"""
        out = []
        for p in self.parts:
            if isinstance(p, LazyVec):
                out += p.dep_nodes()
            elif isinstance(p, Expr):
                out.append(p.node)
            elif hasattr(p, "token"):                  # a per-particle leaf read where it is used (engine.StepInput2)
                out.append(p.token())
            elif isinstance(p, np.ndarray) and p.dtype == object:
                out += [x.node for x in p.reshape(-1) if isinstance(x, Expr)]
        return out

    # arithmetic: through the same scalar builders, element by element on demand
    def __add__(self, o): return _arith("ADD", "IADD", self, o)
    def __radd__(self, o): return _arith("ADD", "IADD", o, self)
    def __sub__(self, o): return _arith("SUB", "ISUB", self, o)
    def __rsub__(self, o): return _arith("SUB", "ISUB", o, self)
    def __mul__(self, o): return _arith("MUL", "IMUL", self, o)
    def __rmul__(self, o): return _arith("MUL", "IMUL", o, self)
    def __truediv__(self, o): return _fbin("DIV", self, o)
    def __rtruediv__(self, o): return _fbin("DIV", o, self)
    def __pow__(self, o): return power(self, o)
    def __rpow__(self, o): return power(o, self)
    def __neg__(self): return lazy_apply(lambda x: -lift(x), self)
    def __pos__(self): return self
    def __abs__(self): return lazy_apply(lambda x: abs(lift(x)), self)
    def __lt__(self, o): return _cmp("FLT", "ILT", self, o)
    def __le__(self, o): return _cmp("FLE", "ILE", self, o)
    def __gt__(self, o): return _cmp("FGT", "IGT", self, o)
    def __ge__(self, o): return _cmp("FGE", "IGE", self, o)
    def __and__(self, o): return _logic("AND", self, o)
    def __rand__(self, o): return _logic("AND", o, self)
    def __or__(self, o): return _logic("OR", self, o)
    def __ror__(self, o): return _logic("OR", o, self)
    def __invert__(self): return lazy_apply(lambda x: ~lift(x), self)

    def astype(self, dt):
        return lazy_apply(lambda x: lift(x).astype(dt), self)


class GradVec(LazyVec):
    """A long vector an HMC move differentiates the model's score with respect to, ELEMENT BY ELEMENT (the positions of
    `HMC(S["theta"])` with theta a vector-valued site of more than 16 elements; hmc.py:69-97 takes jax.grad of assess
    with respect to the whole vector).  It reads like the vector it wraps and records every read: a vector-valued site
    whose loop reads element j at its own iteration j stores d (its j-th term) / d v_j beside the score
    (static._vector_site_loop) — the adjoint of an elementwise consumer is elementwise.  A read any other way (a static
    index, a traced one, `jnp.sum`) is left unconsumed and the move is refused, naming the site."""

    def __init__(self, src):
        n = _long_vector(src)
        LazyVec.__init__(self, n, None, parts=[src])
        self.src = src
        self.reads = []            # (index, the element's Expr)
        self.consumed = 0
        self.contribs = []         # (the consuming site's loop-carried score variable, its stored d term_j / d v_j)
        self.gathers = []          # (loop index, gathered index, the element's Expr, the index vector): `v[group]` reads
        self.gathers_consumed = 0

    def at(self, i):
        v = as_float(_elem(self.src, i))
        self.reads.append((i, v))
        return v

    def __getitem__(self, idx):
        # `theta[group]` with `group` a long vector of indices (a table of group labels): element i of the result is
        # theta at group[i] — a GATHER: the consuming site's loop stores d term_i / d (the gathered value) and a second
        # pair of loops adds them up per group, j by j (static._scatter_add): the adjoint of a gather is a scatter-add
        n_i = _long_vector(idx)
        kind = getattr(getattr(idx, "dtype", None), "kind", "")
        if n_i and (kind in "iu" or getattr(idx, "_dt", None) == "i32"):
            gv = self

            def elem(i):
                j = _elem(idx, i)
                v = as_float(_elem(gv.src, int(j) if isinstance(j, (int, np.integer)) else j))
                gv.gathers.append((i, j, v, idx))
                return v
            out = LazyVec(n_i, elem, parts=(self, idx))
            out._dt = "f32"
            return out
        return LazyVec.__getitem__(self, idx)

    def materialize(self):
        out = np.empty((self.n,), dtype=object)
        for i in range(self.n):
            out[i] = self.at(i)
        return out


def _long_vector(a) -> int:
    """length of a LONG one-axis vector that can be read at a run-time index (a LazyVec; a launch-uniform table; a
    per-particle step leaf), else 0"""
    if isinstance(a, LazyVec):
        return a.n
    if isinstance(a, np.ndarray) and a.ndim == 1 and a.shape[0] >= LAZY_MIN and getattr(a, "_lazy_ok", False):
        return int(a.shape[0])
    if getattr(a, "_lazy_row", False) and a.shape[0] >= LAZY_MIN:      # one row of an [n, A, T] leaf (engine.StepInput2)
        return int(a.shape[0])
    return 0


def lazy_length(args) -> int:
    """n when the operands of an elementwise operation hold a long vector (all long / array operands of one length n,
    the others scalars) — the operation then stays lazy; else 0"""
    n = 0
    for a in args:
        m = _long_vector(a)
        if m:
            if n and m != n:
                return 0
            n = m
    if not n:
        return 0
    for a in args:
        if _long_vector(a):
            continue
        if isinstance(a, (np.ndarray, list, tuple)) and np.ndim(a) > 0:
            if np.ndim(a) != 1 or len(a) != n:
                return 0             # broadcasting against another shape: the unrolled rules apply
    return n


def _elem(a, i):
    if isinstance(a, LazyVec):
        return a.at(i)
    if _long_vector(a):
        return a[i]
    if isinstance(a, (np.ndarray, list, tuple)) and np.ndim(a) > 0:
        if isinstance(i, Expr):
            raise NotImplementedError("a long vector expression mixes a table with a short-lived array of computed "
                                      "values: index the table inside a plate (`vmap`) instead")
        v = a[i]
        return v.item() if isinstance(v, np.ndarray) and v.ndim == 0 else v
    return a


def lazy_apply(fn, *args):
    """fn over the elements of long-vector operands, lazily (see LazyVec)"""
    n = lazy_length(args)
    return LazyVec(n, lambda i: fn(*[_elem(a, i) for a in args]), parts=[a for a in args if not isinstance(a, (int, float, bool))])


def is_symbolic(x) -> bool:
    if isinstance(x, (Expr, LazyVec)):
        return True
    if isinstance(x, np.ndarray) and x.dtype == object:
        return True
    if isinstance(x, (list, tuple)):
        return any(is_symbolic(v) for v in x)
    return False


def lift(x) -> Expr:
    """Python / numpy scalar -> constant Expr."""
    if isinstance(x, Expr):
        return x
    g = current_graph()
    if isinstance(x, (bool, np.bool_)):
        return Expr(g.const_i32(1 if x else 0, "bool"))
    if isinstance(x, (int, np.integer)):
        return Expr(g.const_i32(int(x), "i32"))
    if isinstance(x, (float, np.floating)):
        return Expr(g.const_f32(float(x)))
    if hasattr(x, "shape") and tuple(x.shape) == () and hasattr(x, "item"):
        return lift(x.item())
    if getattr(x, "vector_site", False):          # engine.StepOutput of a looped vector-valued site: trace it unrolled
        from .engine import VectorSiteValueUsed
        raise VectorSiteValueUsed("the model computes with the values of a long vector-valued site")
    raise TypeError(f"cannot use a value of type {type(x).__name__} inside a traced expression")


def as_float(x) -> Expr:
    x = lift(x)
    if x.dtype == "f32":
        return x
    return Expr(current_graph().add("I2F", (x.node,), dtype="f32"))


def as_int(x) -> Expr:
    x = lift(x)
    if x.dtype == "i32":
        return x
    if x.dtype == "bool":
        return Expr(current_graph().add("MOV", (x.node,), dtype="i32"))
    return Expr(current_graph().add("F2I", (x.node,), dtype="i32"))


def as_bool(x) -> Expr:
    x = lift(x)
    if x.dtype == "bool":
        return x
    if x.dtype == "i32":
        z = current_graph().const_i32(0)
        return Expr(current_graph().add("INE", (x.node, z), dtype="bool"))
    z = current_graph().const_f32(0.0)
    return Expr(current_graph().add("FNE", (x.node, z), dtype="bool"))


def _vec(fn):
    """Apply a scalar Expr function elementwise over object arrays."""
    def wrapped(*args):
        if lazy_length(args):
            return lazy_apply(wrapped, *args)
        if any(isinstance(a, (np.ndarray, list, tuple)) and np.ndim(a) > 0 for a in args):
            arrs = np.broadcast_arrays(*[np.asarray(a, dtype=object) if not isinstance(a, np.ndarray) or a.dtype != object
                                         else a for a in args])
            out = np.empty(arrs[0].shape, dtype=object)
            for idx in np.ndindex(out.shape):
                out[idx] = fn(*[a[idx] for a in arrs])
            return out
        return fn(*[a.item() if isinstance(a, np.ndarray) and a.ndim == 0 and a.dtype == object else a
                    for a in args])
    return wrapped


_DEVICE_FOLD = {"LOG", "EXP", "LOG1P", "SQRT"}      # unary ops of a CONSTANT evaluated once, on the device, at trace time
_DEVICE_CONSTS: dict = {}


def _device_const(op: str, bits: int):
    """`op(constant)` in the device's own arithmetic (the fixed-sequence gmx_logf / gmx_expf ..., csrc/gmx_math.h):
    one one-element launch at trace time, cached; None when no launch can be made now (a stream capture in progress)"""
    k = (op, bits)
    r = _DEVICE_CONSTS.get(k)
    if r is not None:
        return r
    try:
        import torch
        from . import _lib, engine, numpy as jnp
        be = _lib.get()
        if be.uses_streams and torch.cuda.is_current_stream_capturing():
            return None
        x = torch.tensor([np.array([bits], np.uint32).view(np.float32)[0]], dtype=torch.float32, device=be.device)
        y = engine.elementwise(getattr(jnp, "_ew_" + op.lower()), x)
        r = int(y.detach().cpu().numpy().view(np.uint32)[0])
    except Exception:      # noqa: BLE001  (no backend at all: keep the operation in the program)
        return None
    _DEVICE_CONSTS[k] = r
    return r


def _un(op, x) -> Expr:
    x = as_float(x)
    if op in _DEVICE_FOLD and x.node.op == "CONST":
        r = _device_const(op, x.node.imm)
        if r is not None:
            return Expr(current_graph().const_bits(r, "f32"))
    return Expr(current_graph().add(op, (x.node,), dtype="f32"))


def _fbin(op, a, b) -> Expr:
    if isinstance(a, LazyVec) or isinstance(b, LazyVec) or lazy_length((a, b)):
        return lazy_apply(lambda p, q: _fbin(op, p, q), a, b)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return _vec(lambda p, q: _fbin(op, p, q))(a, b)
    a, b = as_float(a), as_float(b)
    return Expr(current_graph().add(op, (a.node, b.node), dtype="f32"))


def _arith(fop, iop, a, b) -> Expr:
    if isinstance(a, LazyVec) or isinstance(b, LazyVec) or lazy_length((a, b)):
        return lazy_apply(lambda p, q: _arith(fop, iop, p, q), a, b)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return _vec(lambda p, q: _arith(fop, iop, p, q))(a, b)
    a, b = lift(a), lift(b)
    if a.dtype == "f32" or b.dtype == "f32":
        return _fbin(fop, a, b)
    a, b = as_int(a), as_int(b)
    return Expr(current_graph().add(iop, (a.node, b.node), dtype="i32"))


def _cmp(fop, iop, a, b) -> Expr:
    if isinstance(a, LazyVec) or isinstance(b, LazyVec) or lazy_length((a, b)):
        return lazy_apply(lambda p, q: _cmp(fop, iop, p, q), a, b)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return _vec(lambda p, q: _cmp(fop, iop, p, q))(a, b)
    a, b = lift(a), lift(b)
    if a.dtype == "f32" or b.dtype == "f32":
        a, b = as_float(a), as_float(b)
        return Expr(current_graph().add(fop, (a.node, b.node), dtype="bool"))
    a, b = as_int(a), as_int(b)
    return Expr(current_graph().add(iop, (a.node, b.node), dtype="bool"))


def _logic(op, a, b) -> Expr:
    if isinstance(a, LazyVec) or isinstance(b, LazyVec) or lazy_length((a, b)):
        return lazy_apply(lambda p, q: _logic(op, p, q), a, b)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return _vec(lambda p, q: _logic(op, p, q))(a, b)
    a, b = as_bool(a), as_bool(b)
    return Expr(current_graph().add(op, (a.node, b.node), dtype="bool"))


@_vec
def power(a, b):
    if isinstance(b, (int, float, np.integer, np.floating)) and not isinstance(b, bool):
        if float(b) == 2.0:
            return _un("SQUARE", a)
        if float(b) == 1.0:
            return as_float(a)
        if float(b) == 0.5:
            return _un("SQRT", a)
    return _fbin("POW", a, b)


@_vec
def where(c, a, b):
    a, b = lift(a), lift(b)
    if a.dtype != b.dtype:
        if "f32" in (a.dtype, b.dtype):
            a, b = as_float(a), as_float(b)
        else:
            a, b = as_int(a), as_int(b)
    if a.node is b.node:
        return a                     # both branches are the same value (an untouched plate element under MH)
    c = as_bool(c)
    return Expr(current_graph().add("SEL", (c.node, a.node, b.node), dtype=a.dtype))


SYM_TAKE_MAX = 1024        # rows a traced index selects from (a chain of selects: the values sit in registers)


class SymArray(np.ndarray):
    """An object array of traced values that a TRACED integer can index — `means[z]` with `means` a latent vector (a
    vector-valued site's values, a plate's return values, `jnp.stack([...])`) and `z` a categorical draw, or an array of
    them: the values sit in registers, so the read is a chain of selects over the leading axis (<= 1024 rows; a table in
    memory — an argument, `jnp.array(...)` of numbers — is read at a run-time index instead: numpy.RuntimeTable).  An
    index past the end reads the last row, as jax clamps."""

    # the reductions models call as METHODS (`jnp.array(obs).mean()`, `.std()`: custom_proposal.ipynb c4) go through
    # genjax_amd.numpy's traced forms — numpy's own reductions do not know traced elements
    def _red(name):            # noqa: N805
        def method(self, axis=None, **_kw):
            from . import numpy as jnp
            return getattr(jnp, name)(np.asarray(self, dtype=object).view(np.ndarray), axis)
        return method
    mean, sum, std, var = _red("mean"), _red("sum"), _red("std"), _red("var")
    del _red

    def __getitem__(self, idx):
        if isinstance(idx, Expr) or (isinstance(idx, np.ndarray) and idx.dtype == object):
            return sym_take(self, idx)
        if isinstance(idx, tuple) and idx and (isinstance(idx[0], Expr) or (isinstance(idx[0], np.ndarray) and idx[0].dtype == object)):
            rows = sym_take(self, idx[0])
            lead = np.ndim(idx[0])
            return rows[(slice(None),) * lead + tuple(idx[1:])] if isinstance(rows, np.ndarray) else rows
        return super().__getitem__(idx)


def sym_array(x):
    """`x` as a SymArray when it is an object array of traced values (anything else is returned as it is)"""
    if type(x) is np.ndarray and x.dtype == object and x.ndim >= 1:
        return x.view(SymArray)
    return x


def sym_take(arr, idx):
    base = np.asarray(arr)
    n = base.shape[0]
    if n > SYM_TAKE_MAX:
        raise NotImplementedError(f"a traced index into {n} values computed in the model: a chain of selects (<= {SYM_TAKE_MAX}); "
                                  "index a table (an argument / `jnp.array` of numbers) instead, or write the reads as a plate")
    if isinstance(idx, np.ndarray):
        out = np.empty(idx.shape + base.shape[1:], dtype=object)
        for pos in np.ndindex(idx.shape):
            out[pos] = sym_take(arr, idx[pos])
        return out.view(SymArray)
    i = as_int(idx)
    i = where(i < 0, i + n, i)          # jax wraps a negative index once (`x[-2]` is row n - 2), then clamps past the end
    out = base[n - 1]
    for j in range(n - 2, -1, -1):
        out = where(i == j, base[j], out)
    return sym_array(out)


def unary(op):
    @_vec
    def f(x):
        return _un(op, x)
    return f


@_vec
def minimum(a, b):
    return _fbin("MIN", a, b)


@_vec
def maximum(a, b):
    return _fbin("MAX", a, b)
