"""ChoiceMap / Selection: the address -> value trees of the generative function
interface.  Host-side only (no device work), re-designed as a plain immutable
trie; API names follow the reference so models and inference scripts read the
same (src/genjax/_src/core/generative/choice_map.py: Selection :124-663,
ChoiceMap :752-1393, Static :1534, Choice :1396, Or.build :1699-1733).

Supported address components: strings, ints and tuples of them (static addresses), the full slice `:` (plate
values carry the plate axis themselves) and an ARRAY OF INDICES (`Indexed`, choice_map.py:1453-1531;
address rules `_validate_addr` :699-749: scalar components first, at most one array component, then full slices):
`C["ys", np.array([2, 5, 6]), "y"].set(v)` constrains plate elements 2, 5, 6 to v[0], v[1], v[2] — stored as the
integer sub-addresses 2, 5, 6, which is what `Indexed.get_inner_map(j)` resolves to for a static index array
(`Mask(v[k], True)` where j == idx[k], nothing elsewhere).  A runtime-conditional constraint is a `Mask(value, flag)`
leaf (core/mask.py): the flag may be a launch value (one bool per particle).
"""
from __future__ import annotations

from typing import Any, Callable


class _IndexArray:
    """An array-of-indices address component (hashable wrapper: address tuples are dict keys)."""
    __slots__ = ("idx",)

    def __init__(self, idx):
        self.idx = tuple(int(i) for i in idx)

    def __repr__(self):
        return f"idx{list(self.idx)}"


def _index_array(a):
    """a 1-D integer array / list used as an address component, else None"""
    if isinstance(a, _IndexArray):
        return a
    if isinstance(a, (str, int, bytes)) or a is Ellipsis or isinstance(a, slice) or isinstance(a, tuple):
        return None
    if isinstance(a, list) and a and all(isinstance(i, (int,)) and not isinstance(i, bool) for i in a):
        return _IndexArray(a)
    shape, dt = getattr(a, "shape", None), getattr(a, "dtype", None)
    if shape is not None and dt is not None and len(shape) == 1 and ("int" in str(dt)):
        return _IndexArray([int(i) for i in (a.tolist() if hasattr(a, "tolist") else a)])
    return None


def _norm(addr) -> tuple:
    if isinstance(addr, tuple):
        out = ()
        for a in addr:
            out += _norm(a)
        if sum(isinstance(a, _IndexArray) for a in out) > 1:
            raise ValueError("an address may hold at most one array of indices (choice_map.py:699-749)")
        return out
    ia = _index_array(addr)
    if ia is not None:
        return (ia,)
    shape, dt = getattr(addr, "shape", None), getattr(addr, "dtype", None)
    if shape is not None and tuple(shape) == () and dt is not None and "int" in str(dt):
        return (int(addr),)    # `C[jnp.array(0), ...]`: a concrete scalar index is the integer address
    if addr is Ellipsis:
        return ()              # `Selection.at[..., "y"]`: any plate index — plate values carry the axis themselves
    if isinstance(addr, slice):
        if addr == slice(None):
            return ()          # chm["plate", :, "x"]: plate values carry the plate axis themselves
        raise NotImplementedError("partial slices in addresses (Indexed choice maps): SURVEY.md §8(f) item 2")
    return (addr,)


class ChoiceMapNoValueAtAddress(Exception):
    """Raised by chm[addr] when there is no value at addr (choice_map.py:665)."""


# ---------------------------------------------------------------------------
# Selections
# ---------------------------------------------------------------------------
class Selection:
    """A set of addresses.  `sel(addr)` = sub-selection below addr,
    `sel[addr]` / `addr in sel` = membership, `sel.check()` = membership of ()."""

    # -- constructors -----------------------------------------------------
    @staticmethod
    def all() -> "Selection":
        return _All()

    @staticmethod
    def none() -> "Selection":
        return _None()

    @staticmethod
    def leaf() -> "Selection":
        return _Leaf()

    # -- algebra ------------------------------------------------------------
    def __or__(self, other): return _Or(self, other)
    def __and__(self, other): return _And(self, other)
    def __invert__(self): return _Complement(self)
    def complement(self): return ~self

    def extend(self, *addr) -> "Selection":
        out = self
        for a in reversed(_norm(addr)):
            out = _Static(a, out)
        return out

    # -- queries --------------------------------------------------------------
    def check(self) -> bool:
        raise NotImplementedError

    def get_subselection(self, comp) -> "Selection":
        raise NotImplementedError

    def __call__(self, addr) -> "Selection":
        s = self
        for a in _norm(addr):
            s = s.get_subselection(a)
        return s

    def __getitem__(self, addr) -> bool:
        return self(addr).check()

    def __contains__(self, addr) -> bool:
        return self[addr]


class _All(Selection):
    def check(self): return True
    def get_subselection(self, comp): return self
    def __repr__(self): return "Selection.all()"


class _None(Selection):
    def check(self): return False
    def get_subselection(self, comp): return self
    def __repr__(self): return "Selection.none()"


class _Leaf(Selection):
    """Selects exactly () (LeafSel, choice_map.py:405)."""
    def check(self): return True
    def get_subselection(self, comp): return _None()


class _Static(Selection):
    def __init__(self, comp, sub):
        self.comp, self.sub = comp, sub

    def check(self): return False

    def get_subselection(self, comp):
        return self.sub if comp == self.comp else _None()

    def __repr__(self): return f"S[{self.comp!r}]({self.sub!r})"


class _Or(Selection):
    def __init__(self, a, b): self.a, self.b = a, b
    def check(self): return self.a.check() or self.b.check()
    def get_subselection(self, comp): return _Or(self.a.get_subselection(comp), self.b.get_subselection(comp))


class _And(Selection):
    def __init__(self, a, b): self.a, self.b = a, b
    def check(self): return self.a.check() and self.b.check()
    def get_subselection(self, comp): return _And(self.a.get_subselection(comp), self.b.get_subselection(comp))


class _Complement(Selection):
    def __init__(self, s): self.s = s
    def check(self): return not self.s.check()
    def get_subselection(self, comp): return _Complement(self.s.get_subselection(comp))
    def __invert__(self): return self.s


class _Chm(Selection):
    """Addresses that hold a value in a choice map (ChmSel, choice_map.py:627-663)."""
    def __init__(self, chm): self.chm = chm
    def check(self): return self.chm.has_value()
    def get_subselection(self, comp): return _Chm(self.chm.get_submap(comp))


class _SelectionBuilder:
    """`S["x"]`, `S["x", "y"]`; a selected address selects everything below it."""

    def __getitem__(self, addr) -> Selection:
        return Selection.all().extend(*_norm(addr))

    # `from genjax import Selection as S` is as common in the reference's tests as SelectionBuilder: S.all() / S.none()
    @staticmethod
    def all() -> Selection:
        return Selection.all()

    @staticmethod
    def none() -> Selection:
        return Selection.none()


SelectionBuilder = _SelectionBuilder()
Selection.at = SelectionBuilder          # Selection.at["x"]


# ---------------------------------------------------------------------------
# Choice maps
# ---------------------------------------------------------------------------
_NOVALUE = object()


class ChoiceMap:
    """Immutable trie: an optional value at this node plus named children."""

    __slots__ = ("_value", "_children", "_plate")

    def __init__(self, value=_NOVALUE, children=None, plate=None):
        self._value = value
        self._children = children or {}
        # the choices of a plate / scan trace: every value below carries the plate axis at position `plate` (after the
        # particle axes), and an INTEGER address component here reads element j of all of them — the reference's
        # `chm[j, "x"]` on a vmapped trace (choice_map.py:1453-1531 `Indexed.get_inner_map`)
        self._plate = plate

    def with_plate(self, axis: int) -> "ChoiceMap":
        return ChoiceMap(self._value, self._children, int(axis))

    # -- builders ---------------------------------------------------------------
    @staticmethod
    def empty() -> "ChoiceMap":
        return _EMPTY

    @staticmethod
    def choice(v) -> "ChoiceMap":
        return ChoiceMap(value=v)

    value = choice
    v = choice

    @staticmethod
    def n() -> "ChoiceMap":
        return _EMPTY

    @staticmethod
    def d(mapping: dict) -> "ChoiceMap":
        out = _EMPTY
        for a, val in mapping.items():
            out = out.set(a, val)
        return out

    from_mapping = d

    @staticmethod
    def kw(**kwargs) -> "ChoiceMap":
        return ChoiceMap.d(kwargs)

    @staticmethod
    def entry(v, *addr) -> "ChoiceMap":
        return _EMPTY.set(addr, v)

    def set(self, addr, v) -> "ChoiceMap":
        """New map with `v` (a value or a ChoiceMap) at addr; existing entries
        elsewhere are kept, the new entry wins at addr."""
        addr = _norm(addr) if not (isinstance(addr, tuple) and not addr) else ()
        if not addr:
            return v if isinstance(v, ChoiceMap) else ChoiceMap(value=v)
        head, rest = addr[0], addr[1:]
        kids = dict(self._children)
        if isinstance(head, _IndexArray):
            # Indexed (choice_map.py:1453-1531) with a static index array: element idx[k] takes v[k]
            if len(set(head.idx)) != len(head.idx):
                raise ValueError("an array of indices in an address must not repeat an index")
            for k, j in enumerate(head.idx):
                kids[j] = kids.get(j, _EMPTY).set(rest, _take_indexed(v, k, len(head.idx)))
            return ChoiceMap(self._value, kids)
        kids[head] = kids.get(head, _EMPTY).set(rest, v)
        return ChoiceMap(self._value, kids, self._plate)

    def extend(self, *addr) -> "ChoiceMap":
        out = self
        for a in reversed(_norm(addr)):
            out = ChoiceMap(children={a: out})
        return out

    @property
    def at(self):
        return _AddressIndex(self)

    def __class_getitem__(cls, addr):            # ChoiceMap["x"].set(v)
        return _AddressIndex(_EMPTY)[addr]

    # -- queries ------------------------------------------------------------------
    def has_value(self) -> bool:
        return self._value is not _NOVALUE

    def get_value(self):
        return None if self._value is _NOVALUE else self._value

    def static_is_empty(self) -> bool:
        return self._value is _NOVALUE and all(c.static_is_empty() for c in self._children.values())

    def get_submap(self, *addr) -> "ChoiceMap":
        cm = self
        for a in _norm(addr):
            if isinstance(a, int) and not isinstance(a, bool) and a not in cm._children and cm._plate is not None:
                cm = cm._take_plate(a)
                continue
            cm = cm._children.get(a, _EMPTY)
        return cm

    def _take_plate(self, j: int) -> "ChoiceMap":
        """element j along this node's plate axis, for every value below; deeper plates move up one axis"""
        ax = self._plate

        def take(v):
            shape = getattr(v, "shape", None)
            if shape is None or len(shape) <= ax:
                return v
            if not -shape[ax] <= j < shape[ax]:
                raise IndexError(f"plate index {j} out of range for an axis of length {shape[ax]}")
            return v.select(ax, j) if hasattr(v, "select") else v.take(j, axis=ax)

        def go(cm, top):
            plate = None if top else (cm._plate - 1 if (cm._plate is not None and cm._plate > ax) else cm._plate)
            return ChoiceMap(cm._value if cm._value is _NOVALUE else take(cm._value),
                             {a: go(c, False) for a, c in cm._children.items()}, plate)
        return go(self, True)

    def __call__(self, *addr) -> "ChoiceMap":
        return self.get_submap(*addr)

    def __getitem__(self, addr):
        sub = self.get_submap(addr)
        if not sub.has_value():
            raise ChoiceMapNoValueAtAddress(addr)
        return sub._value

    def __contains__(self, addr) -> bool:
        return self.get_submap(addr).has_value()

    def get_selection(self) -> Selection:
        return _Chm(self)

    def addresses(self, prefix=()) -> list:
        out = [prefix] if self.has_value() else []
        for a, c in self._children.items():
            out += c.addresses(prefix + (a,))
        return out

    def items(self):
        return [(a, self[a] if a else self._value) for a in self.addresses()]

    def to_dict(self) -> dict:
        return {(a if len(a) != 1 else a[0]): v for a, v in self.items()}

    # -- algebra ------------------------------------------------------------------
    def merge(self, other: "ChoiceMap") -> "ChoiceMap":
        """`self | other`; on overlap the FIRST operand wins (Or.build,
        choice_map.py:1699-1733) — so a Target's own constraints take precedence."""
        if self.static_is_empty():
            return other
        if other.static_is_empty():
            return self
        value = self._value if self._value is not _NOVALUE else other._value
        kids = dict(self._children)
        for a, c in other._children.items():
            kids[a] = kids[a].merge(c) if a in kids else c
        return ChoiceMap(value, kids)

    def __or__(self, other): return self.merge(other)
    def __xor__(self, other): return self.merge(other)

    def filter(self, selection: Selection) -> "ChoiceMap":
        value = self._value if (self._value is not _NOVALUE and selection.check()) else _NOVALUE
        kids = {}
        for a, c in self._children.items():
            f = c.filter(selection.get_subselection(a))
            if not f.static_is_empty():
                kids[a] = f
        return ChoiceMap(value, kids, self._plate)

    def map_values(self, fn: Callable[[Any], Any]) -> "ChoiceMap":
        return ChoiceMap(self._value if self._value is _NOVALUE else fn(self._value),
                         {a: c.map_values(fn) for a, c in self._children.items()}, self._plate)

    def mask(self, flag) -> "ChoiceMap":
        from .mask import Mask
        return self.map_values(lambda v: Mask.build(v, flag))

    # -- misc -------------------------------------------------------------------------
    def __eq__(self, other):
        return isinstance(other, ChoiceMap) and self.to_dict().keys() == other.to_dict().keys() and all(
            _same(self[a], other[a]) for a in self.addresses() if a) and _same(self.get_value(), other.get_value())

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return "ChoiceMap(" + ", ".join(f"{a}: {_short(v)}" for a, v in self.to_dict().items()) + ")"

    def structure(self):
        """Hashable description of which addresses hold values (program cache key)."""
        return tuple(self.addresses())


def _take_indexed(v, k, m):
    """entry k of a value (or of every leaf of a choice map) whose leading axis — or, for a per-particle [n, m]
    tensor, whose last axis — runs over the m indices of an array-of-indices address"""
    if isinstance(v, ChoiceMap):
        return v.map_values(lambda x: _take_indexed(x, k, m))
    shape = getattr(v, "shape", None)
    if shape is None:
        if isinstance(v, (list, tuple)) and len(v) == m:
            return v[k]
        raise ValueError("a value set at an array-of-indices address needs one entry per index")
    shape = tuple(shape)
    if len(shape) >= 1 and shape[0] == m:
        return v[k]
    if len(shape) == 2 and shape[1] == m:
        return v[:, k]
    raise ValueError(f"a value of shape {shape} does not have one entry for each of the {m} indices")


def _same(a, b):
    if a is b:
        return True
    try:
        import torch
        if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
            return bool(torch.equal(torch.as_tensor(a), torch.as_tensor(b)))
    except Exception:
        pass
    try:
        import numpy as np
        return bool(np.array_equal(a, b))
    except Exception:
        return a == b


def _short(v):
    s = getattr(v, "shape", None)
    return f"<{type(v).__name__}{tuple(s)}>" if s is not None and tuple(s) != () else repr(v)


_EMPTY = ChoiceMap()


class _AddressIndex:
    """`chm.at["a", "b"].set(v)` / `C["a"].set(v)` / `.get()`."""

    def __init__(self, chm, addr=()):
        self.chm, self.addr = chm, addr

    def __getitem__(self, addr):
        return _AddressIndex(self.chm, self.addr + _norm(addr))

    def set(self, v) -> ChoiceMap:
        return self.chm.set(self.addr, v) if self.addr else (v if isinstance(v, ChoiceMap) else ChoiceMap.choice(v))

    def get(self):
        return self.chm[self.addr]

    def n(self) -> ChoiceMap:
        return self.chm


class _ChoiceMapBuilder:
    """`from genjax import ChoiceMapBuilder as C`: C["x"].set(v), C.kw(...),
    C.d({...}), C.v(v), C.n() (choice_map.py:752-845)."""

    def __getitem__(self, addr):
        return _AddressIndex(_EMPTY)[addr]

    n = staticmethod(ChoiceMap.n)
    v = staticmethod(ChoiceMap.choice)
    d = staticmethod(ChoiceMap.d)
    kw = staticmethod(ChoiceMap.kw)
    choice = staticmethod(ChoiceMap.choice)
    empty = staticmethod(ChoiceMap.empty)


ChoiceMapBuilder = _ChoiceMapBuilder()
