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


class DynamicIndex:
    """`C["ys", dynamic_index(idx), "y"].set(v)`: a RUN-TIME index, one per particle (or one, known only on the
    device) — unlike a concrete array of indices, which names several static elements.  `transforms.vmap` tags the
    integer tensors it maps (engine.Mapped), so under `genjax.vmap` the reference's plain spelling
    `C["ys", idx, "y"].set(v)` means this."""
    __slots__ = ("idx",)

    def __init__(self, idx):
        self.idx = idx

    def __repr__(self):
        return "dynamic_index(...)"


def dynamic_index(idx) -> DynamicIndex:
    return DynamicIndex(idx)


class _IndexArray:
    """An array-of-indices address component (hashable wrapper: address tuples are dict keys)."""
    __slots__ = ("idx",)

    def __init__(self, idx):
        self.idx = tuple(int(i) for i in idx)

    def __repr__(self):
        return f"idx{list(self.idx)}"


def _index_array(a):
    """a 1-D integer array / list used as an address component, else None"""
    if isinstance(a, (_IndexArray, DynamicIndex)):
        return a
    if getattr(a, "_gmx_mapped", False) and "int" in str(getattr(a, "dtype", "")):
        return DynamicIndex(a)             # an integer tensor mapped by genjax.vmap: one index per instance
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
        if sum(isinstance(a, (_IndexArray, DynamicIndex)) for a in out) > 1:
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
        return (_Slice(addr),)  # a partial slice: allowed when READING a plate (chm[0:4, "x"]), refused by set()
    return (addr,)


class _Slice:
    """a partial slice as an address component (hashable)"""
    __slots__ = ("sl",)

    def __init__(self, sl):
        self.sl = sl

    def __repr__(self):
        return f"{self.sl.start}:{self.sl.stop}" + (f":{self.sl.step}" if self.sl.step is not None else "")


class ChoiceMapNoValueAtAddress(Exception):
    """Raised by chm[addr] when there is no value at addr (choice_map.py:665)."""


# ---------------------------------------------------------------------------
# Selections
# ---------------------------------------------------------------------------
_WILD = Ellipsis       # S[..., "y"]: any first component


def _sel_norm(addr) -> tuple:
    """address components of a SELECTION query / builder: like _norm, but a leading `...` is kept as the wildcard
    (choice_map.py:261-347); anywhere else it is refused"""
    if not isinstance(addr, tuple):
        addr = (addr,)
    out = ()
    for k, a in enumerate(addr):
        if a is Ellipsis:
            if k != 0:
                raise TypeError("`...` may only be the FIRST component of a selection address")
            out += (_WILD,)
        elif isinstance(a, tuple):
            out += _sel_norm(a)
        else:
            out += _norm(a)
    return out


class Selection:
    """A set of addresses.  `sel(addr)` = sub-selection below addr,
    `sel[addr]` / `addr in sel` = membership, `sel.check()` = membership of ().
    Structural equality; the constructors simplify (`~~s == s`, `all & s == s`, `none | s == s`, `s | s == s`, ...:
    choice_map.py:196-259)."""

    # -- constructors -----------------------------------------------------
    @staticmethod
    def all() -> "Selection":
        return _ALL

    @staticmethod
    def none() -> "Selection":
        return _NONE

    @staticmethod
    def leaf() -> "Selection":
        return _LEAF

    # -- algebra ------------------------------------------------------------
    def __or__(self, other):
        if isinstance(self, _All) or isinstance(other, _All):
            return _ALL
        if isinstance(self, _None):
            return other
        if isinstance(other, _None) or self == other:
            return self
        return _Or(self, other)

    def __and__(self, other):
        if isinstance(self, _None) or isinstance(other, _None):
            return _NONE
        if isinstance(self, _All):
            return other
        if isinstance(other, _All) or self == other:
            return self
        return _And(self, other)

    def __invert__(self):
        if isinstance(self, _All):
            return _NONE
        if isinstance(self, _None):
            return _ALL
        if isinstance(self, _Complement):
            return self.s
        return _Complement(self)

    def complement(self): return ~self

    def extend(self, *addr) -> "Selection":
        if isinstance(self, _None):
            return self                      # nothing to extend
        out = self
        for a in reversed(_sel_norm(addr)):
            out = _Static(a, out)
        return out

    def filter(self, chm: "ChoiceMap") -> "ChoiceMap":
        return chm.filter(self)

    # -- queries --------------------------------------------------------------
    def check(self) -> bool:
        raise NotImplementedError

    def get_subselection(self, comp) -> "Selection":
        raise NotImplementedError

    def __call__(self, *addr) -> "Selection":
        s = self
        for a in _sel_norm(addr):
            if a is _WILD:
                raise TypeError("`...` is for building selections (S[..., 'y']), not for querying them")
            s = s.get_subselection(a)
        return s

    def __getitem__(self, addr) -> bool:
        return self(addr).check()

    def __contains__(self, addr) -> bool:
        return self[addr]

    # -- structure ---------------------------------------------------------------
    def _key(self):
        return (type(self).__name__,)

    def __eq__(self, other):
        return isinstance(other, Selection) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())


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
    def get_subselection(self, comp): return _NONE
    def __repr__(self): return "Selection.leaf()"


_ALL, _NONE, _LEAF = _All(), _None(), _Leaf()


class _Static(Selection):
    def __init__(self, comp, sub):
        self.comp, self.sub = comp, sub

    def check(self): return False

    def get_subselection(self, comp):
        return self.sub if (self.comp is _WILD or comp == self.comp) else _NONE

    def _key(self): return ("static", "..." if self.comp is _WILD else self.comp, self.sub._key())
    def __repr__(self): return f"S[{self.comp!r}]({self.sub!r})"


class _Or(Selection):
    def __init__(self, a, b): self.a, self.b = a, b
    def check(self): return self.a.check() or self.b.check()
    def get_subselection(self, comp): return self.a.get_subselection(comp) | self.b.get_subselection(comp)
    def _key(self): return ("or", self.a._key(), self.b._key())


class _And(Selection):
    def __init__(self, a, b): self.a, self.b = a, b
    def check(self): return self.a.check() and self.b.check()
    def get_subselection(self, comp): return self.a.get_subselection(comp) & self.b.get_subselection(comp)
    def _key(self): return ("and", self.a._key(), self.b._key())


class _Complement(Selection):
    def __init__(self, s): self.s = s
    def check(self): return not self.s.check()
    def get_subselection(self, comp): return ~self.s.get_subselection(comp)
    def _key(self): return ("not", self.s._key())


class _Chm(Selection):
    """Addresses that hold a value in a choice map (ChmSel, choice_map.py:627-663).  An `Indexed` layer of the map — a
    constraint on a SUBSET of a plate's elements — selects NOTHING below it: `ChmSel.get_subselection(site)` asks
    `Indexed.get_inner_map(site)` with a static component and gets the empty map (choice_map.py:1494-1496).  Here a
    static array of indices is stored as integer sub-addresses (never met by a plate trace's own addresses, so nothing
    below is selected either way) and a run-time index as an `Indexed(value, idx)` value AT the site: that value does
    not make the site selected."""
    def __init__(self, chm): self.chm = chm

    def check(self):
        from .mask import Indexed
        return self.chm.has_value() and not isinstance(self.chm._value, Indexed)

    def get_subselection(self, comp):
        sub = self.chm.get_submap(comp)
        return _Chm(sub) if not sub.static_is_empty() else _NONE

    def _key(self): return ("chm", tuple(self.chm.addresses()))


class _SelProp:
    """`S.all` / `S.none` / `S.leaf` read as properties (`S.all["x"]`) AND as calls (`S.all()`): the reference's tests
    use both spellings (SelectionBuilder properties; `from genjax import Selection as S`)."""

    def __init__(self, sel):
        self._sel = sel

    def __get__(self, obj, owner=None):
        return _CallableSel(self._sel)


class _CallableSel:
    def __init__(self, sel): self._sel = sel
    def __call__(self, *addr): return self._sel if not addr else self._sel(*addr)
    def __getitem__(self, addr): return self._sel[addr]
    def __eq__(self, other): return self._sel == (other._sel if isinstance(other, _CallableSel) else other)
    def __hash__(self): return hash(self._sel)
    def __getattr__(self, name): return getattr(self._sel, name)
    def __or__(self, o): return self._sel | o
    def __and__(self, o): return self._sel & o
    def __invert__(self): return ~self._sel
    def __contains__(self, addr): return addr in self._sel
    def __repr__(self): return repr(self._sel)


class _SelectionBuilder:
    """`S["x"]`, `S["x", "y"]`, `S[..., "y"]`, `S[()]`; a selected address selects everything below it."""
    all = _SelProp(_ALL)
    none = _SelProp(_NONE)
    leaf = _SelProp(_LEAF)

    def __getitem__(self, addr) -> Selection:
        comps = _sel_norm(addr)
        if not comps:
            return _LEAF                      # S[()]
        return _ALL.extend(*comps)


SelectionBuilder = _SelectionBuilder()
Selection.at = SelectionBuilder          # Selection.at["x"]


# ---------------------------------------------------------------------------
# Choice maps
# ---------------------------------------------------------------------------
_NOVALUE = object()


def _is_index_array(a) -> bool:
    """a HOST 1-d integer array used as an address component: a static list of plate elements"""
    return (not isinstance(a, (str, bytes, int))) and getattr(getattr(a, "dtype", None), "kind", "") in "iu" \
        and type(a).__module__.split(".")[0] != "torch" and len(getattr(a, "shape", ())) == 1


class ChoiceMap:
    """Immutable trie: an optional value at this node plus named children."""

    __slots__ = ("_value", "_children", "_plate", "_pdepth")

    def __init__(self, value=_NOVALUE, children=None, plate=None, pdepth=1):
        self._value = value
        self._children = children or {}
        # (plates nest — `sample_pixel.vmap().vmap()`, iterating_computation.ipynb c11: `chm[0, 0, "new_pixel"]` — : how many
        #  plate levels sit at this node; taking an element of the outer one leaves the next at the same axis)
        self._pdepth = int(pdepth) if plate is not None else 1
        # the choices of a plate / scan trace: every value below carries the plate axis at position `plate` (after the
        # particle axes), and an INTEGER address component here reads element j of all of them — the reference's
        # `chm[j, "x"]` on a vmapped trace (choice_map.py:1453-1531 `Indexed.get_inner_map`)
        self._plate = plate

    def with_plate(self, axis: int, depth: int = 1) -> "ChoiceMap":
        nested = self._plate is not None and self._plate > int(axis)          # a plate INSIDE the one being marked
        return ChoiceMap(self._value, self._children, int(axis), max(int(depth), self._pdepth + 1 if nested else 1))

    # -- builders ---------------------------------------------------------------
    @staticmethod
    def empty() -> "ChoiceMap":
        return _EMPTY

    @staticmethod
    def choice(v) -> "ChoiceMap":
        """a value-only map (choice_map.py:1396-1450).  A Mask whose flag is a concrete bool resolves now (False: the
        empty map; True: the bare value); an array with no elements is the empty map."""
        from .mask import Mask
        if isinstance(v, Mask) and isinstance(v.flag, bool):
            return ChoiceMap(value=v.value) if v.flag else _EMPTY
        shape = getattr(v, "shape", None)
        if shape is not None and len(shape) >= 1 and 0 in tuple(shape):
            return _EMPTY
        return ChoiceMap(value=v)

    value = choice
    v = choice

    @staticmethod
    def n() -> "ChoiceMap":
        return _EMPTY

    @staticmethod
    def d(mapping) -> "ChoiceMap":
        """from a dict (or a list of (address, value) pairs); dict values nest (choice_map.py:800-845)"""
        out = _EMPTY
        for a, val in (mapping.items() if isinstance(mapping, dict) else mapping):
            out = out.set(a if isinstance(a, tuple) else (a,), val)
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
        if isinstance(v, dict):
            v = ChoiceMap.d(v)
        raw = addr if isinstance(addr, tuple) else (addr,)
        for k, a in enumerate(raw):
            if isinstance(a, slice):
                if a != slice(None):
                    raise ValueError("partial slices are not allowed when setting (choice_map.py:699-749); use `:`")
                # C[prefix, :, rest].set(v): a PLATE — the values below carry the plate axis in front
                below = _EMPTY.set(tuple(x for x in raw[k + 1:] if not (isinstance(x, slice) and x == slice(None))), v)
                below = below.with_plate(0) if not below.static_is_empty() else below
                return self.set(raw[:k], below) if raw[:k] else self.merge_over(below)
        addr = _norm(addr) if not (isinstance(addr, tuple) and not addr) else ()
        if not addr:
            return v if isinstance(v, ChoiceMap) else ChoiceMap(value=v)
        head, rest = addr[0], addr[1:]
        if isinstance(head, _Slice):
            raise ValueError("partial slices are not allowed when setting (choice_map.py:699-749); use `:`")
        if isinstance(head, DynamicIndex):
            # a run-time index: the constraint sits at the plate's own address as Indexed(value, idx); the plate turns
            # it into Mask(value, idx == j) for its element j
            from .mask import Indexed
            if any(isinstance(a, (DynamicIndex, _IndexArray)) for a in rest):
                raise ValueError("an address may hold at most one array of indices (choice_map.py:699-749)")
            wrap = (lambda x: Indexed(x, head.idx))
            return self.set(rest, v.map_values(wrap) if isinstance(v, ChoiceMap) else Indexed(v, head.idx)) if rest else \
                self.merge_over(v.map_values(wrap) if isinstance(v, ChoiceMap) else ChoiceMap(value=Indexed(v, head.idx)))
        kids = dict(self._children)
        if isinstance(head, _IndexArray):
            # Indexed (choice_map.py:1453-1531) with a static index array: element idx[k] takes v[k]
            if len(set(head.idx)) != len(head.idx):
                raise ValueError("an array of indices in an address must not repeat an index")
            for k, j in enumerate(head.idx):
                kids[j] = kids.get(j, _EMPTY).set(rest, _take_indexed(v, k, len(head.idx)))
            return ChoiceMap(self._value, kids)
        kids[head] = kids.get(head, _EMPTY).set(rest, v)
        return ChoiceMap(self._value, kids, self._plate, self._pdepth)

    def merge_over(self, new: "ChoiceMap") -> "ChoiceMap":
        """`new` laid over this map (new entries win), keeping new's plate marker"""
        out = new.merge(self)
        return ChoiceMap(out._value, out._children, new._plate if new._plate is not None else self._plate)

    def extend(self, *addr) -> "ChoiceMap":
        if self.static_is_empty():
            return self
        out = self
        for a in reversed(addr):
            if isinstance(a, slice) and a == slice(None):
                out = out.with_plate(0)
                continue
            for c in reversed(_norm(a)):
                out = ChoiceMap(children={c: out})
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
            if isinstance(a, _Slice):
                cm = cm._take_plate(a.sl)
                continue
            if isinstance(a, int) and not isinstance(a, bool) and a not in cm._children and \
                    (cm._plate is not None or (cm.has_value() and len(getattr(cm._value, "shape", ())) >= 1)):
                cm = cm._take_plate(a)          # element a of a plate (or of a value-only array)
                continue
            if isinstance(a, _IndexArray) and a not in cm._children:
                import numpy as _np
                a = _np.asarray(a.idx, dtype=_np.int64)
            if _is_index_array(a) and (cm._plate is not None or (cm.has_value() and len(getattr(cm._value, "shape", ())) >= 1)):
                cm = cm._take_plate(a)          # elements a[0], a[1], ... of the plate, still a plate (`chm["obs", idxs]`)
                continue
            cm = cm._children.get(a, _EMPTY)
        return cm

    def _take_plate(self, j: int) -> "ChoiceMap":
        """element j along this node's plate axis, for every value below; deeper plates move up one axis"""
        ax = self._plate if self._plate is not None else 0
        is_slice = isinstance(j, slice) or _is_index_array(j)

        def take(v):
            shape = getattr(v, "shape", None)
            if shape is None or len(shape) <= ax:
                return v
            if _is_index_array(j):
                import numpy as _np
                idx = _np.asarray(j).astype(_np.int64)
                if idx.size and not (-shape[ax] <= idx.min() and idx.max() < shape[ax]):
                    raise IndexError(f"plate indices out of range for an axis of length {shape[ax]}")
                if hasattr(v, "index_select"):
                    import torch
                    return v.index_select(ax, torch.as_tensor(idx % shape[ax], device=v.device))
                return v.take(idx, axis=ax)
            if is_slice:
                return v[(slice(None),) * ax + (j,)]
            if not -shape[ax] <= j < shape[ax]:
                raise IndexError(f"plate index {j} out of range for an axis of length {shape[ax]}")
            return v.select(ax, j) if hasattr(v, "select") else v.take(j, axis=ax)

        def go(cm, top):
            depth = cm._pdepth
            if is_slice:
                plate = cm._plate
            elif top:
                plate, depth = (ax, cm._pdepth - 1) if (cm._plate is not None and cm._pdepth > 1) else (None, 1)
            else:
                plate = cm._plate - 1 if (cm._plate is not None and cm._plate > ax) else cm._plate
            return ChoiceMap(cm._value if cm._value is _NOVALUE else take(cm._value),
                             {a: go(c, False) for a, c in cm._children.items()}, plate, depth)
        return go(self, True)

    def __call__(self, *addr) -> "ChoiceMap":
        return self.get_submap(*addr)

    def __getitem__(self, addr):
        sub = self.get_submap(addr)
        if not sub.has_value():
            raise ChoiceMapNoValueAtAddress(addr)
        return sub._value

    @property
    def attributes(self):
        return {"value": self.get_value(), "children": dict(self._children)}

    def __contains__(self, addr) -> bool:
        return self.get_submap(addr).has_value()

    def get_selection(self) -> Selection:
        return _Chm(self) if not self.static_is_empty() else _NONE

    def _without(self, addr: tuple) -> "ChoiceMap":
        """this map with everything at and below addr removed"""
        if not addr:
            return _EMPTY
        kids = dict(self._children)
        if addr[0] in kids:
            sub = kids[addr[0]]._without(addr[1:])
            if sub.static_is_empty():
                del kids[addr[0]]
            else:
                kids[addr[0]] = sub
        return ChoiceMap(self._value, kids, self._plate, self._pdepth)

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
        if (self.has_value() and other._children) or (other.has_value() and self._children):
            raise Exception("Choice and non-Choice in Or: a value and a sub-map at one address (choice_map.py:1699-1733)")
        value = self._value if self._value is not _NOVALUE else other._value
        if self._value is not _NOVALUE and other._value is not _NOVALUE:
            value = _or_values(self._value, other._value)
        kids = dict(self._children)
        for a, c in other._children.items():
            kids[a] = kids[a].merge(c) if a in kids else c
        if other._plate is not None and any(isinstance(a, int) for a in self._children):
            kids = _overlay_elements(self, other, kids)
        return ChoiceMap(value, kids, self._plate if self._plate is not None else other._plate)

    def __or__(self, other): return self.merge(other)
    def __xor__(self, other): return self.merge(other)

    def __and__(self, other: "ChoiceMap") -> "ChoiceMap":
        """the addresses both maps hold, values from the RIGHT operand (choice_map.py `And`)"""
        value = other._value if (self.has_value() and other.has_value()) else _NOVALUE
        kids = {}
        for a, c in self._children.items():
            if a in other._children:
                sub = c & other._children[a]
                if not sub.static_is_empty():
                    kids[a] = sub
        return ChoiceMap(value, kids)

    def invalid_subset(self, gen_fn, args):
        """The part of this map `gen_fn(*args)` never visits, or None (choice_map.py `invalid_subset`): the model is
        run once (assess-free: simulate with a dummy key) and the addresses of its trace are compared — an index layer
        over a plate's addresses is optional, a MISSING address is fine, an EXTRA one is reported."""
        from ..random import key as _key
        shape = gen_fn.simulate(_key(0), tuple(args)).get_choices()

        def extra(mine: "ChoiceMap", theirs: "ChoiceMap") -> "ChoiceMap":
            value = mine._value if (mine.has_value() and not theirs.has_value()) else _NOVALUE
            kids = {}
            for a, c in mine._children.items():
                if a in theirs._children:
                    sub = extra(c, theirs._children[a])
                elif isinstance(a, int) and theirs._plate is not None:
                    sub = extra(c, theirs)            # an explicit element of a plate
                else:
                    sub = c
                if not sub.static_is_empty():
                    kids[a] = sub
            return ChoiceMap(value, kids, mine._plate)
        bad = extra(self, shape)
        return None if bad.static_is_empty() else bad

    def filter(self, selection: Selection) -> "ChoiceMap":
        value = self._value if (self._value is not _NOVALUE and selection.check()) else _NOVALUE
        kids = {}
        for a, c in self._children.items():
            f = c.filter(selection.get_subselection(a))
            if not f.static_is_empty():
                kids[a] = f
        return ChoiceMap(value, kids, self._plate, self._pdepth)

    def map_values(self, fn: Callable[[Any], Any]) -> "ChoiceMap":
        return ChoiceMap(self._value if self._value is _NOVALUE else fn(self._value),
                         {a: c.map_values(fn) for a, c in self._children.items()}, self._plate, self._pdepth)

    def mask(self, flag) -> "ChoiceMap":
        """every value wrapped in Mask(value, flag); a concrete flag resolves now (True: this map; False: empty)"""
        from .mask import Mask
        if isinstance(flag, bool):
            return self if flag else _EMPTY
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


def _or_values(a, b):
    """`Choice(a) | Choice(b)` = `Choice.build(Mask.build(a) | Mask.build(b))` (choice_map.py:1714-1717): a first operand
    that is a Mask with a run-time flag holds where its flag does, the second operand elsewhere (`Mask.__or__`,
    functional_types.py:309-319); anything else: the first operand."""
    from .mask import Mask
    if isinstance(a, Mask) and a.flag is False:
        return b                 # `Mask.__or__` (functional_types.py:312-316): a first operand whose flag is statically False gives way
    if not isinstance(a, Mask) or isinstance(a.flag, bool):
        return a
    import torch
    if not all(isinstance(x, torch.Tensor) for x in (a.flag, a.value)):
        return a
    bv, bf = (b.value, b.flag) if isinstance(b, Mask) else (b, True)
    if not isinstance(bv, torch.Tensor):
        bv = torch.as_tensor(bv, dtype=a.value.dtype, device=a.value.device)
    fl = a.flag.reshape(tuple(a.flag.shape) + (1,) * (max(a.value.ndim, bv.ndim) - a.flag.ndim)) if a.flag.ndim else a.flag
    value = torch.where(fl, a.value, bv.to(a.value.device))
    if isinstance(bf, bool):
        return value if bf else Mask(value, a.flag)
    return Mask(value, a.flag | bf.to(a.flag.device))


def _overlay_elements(first: "ChoiceMap", second: "ChoiceMap", kids: dict) -> dict:
    """`first | second` at the node of a plate / scan where FIRST constrains single elements (`C[name, idx, site]`, stored
    as integer sub-addresses) and SECOND the whole axis (a plate trace's choices handed on as latents by `ChangeTarget`,
    smc.py:378-384): element j of the merged constraint is `Choice(Mask(v[k], j == idx[k])) | Choice(vals[j])`
    (Or.get_inner_map, choice_map.py:1740-1743; Indexed.get_inner_map :1508-1531) — the listed elements take the first
    operand's values, every other element the second's.  The overlaid values are written into a copy of the second
    operand's leaf (data movement: no arithmetic); the integer entries that went into it are dropped."""
    import numpy as np
    import torch
    from .mask import Mask
    ax = int(second._plate)
    rest = ChoiceMap(_NOVALUE, {a: c for a, c in kids.items() if not isinstance(a, int)}, second._plate)
    own = ChoiceMap(_NOVALUE, {a: c for a, c in first._children.items() if not isinstance(a, int)})
    out = dict(kids)
    for i in sorted(a for a in first._children if isinstance(a, int)):
        sub = first._children[i]
        for addr in sub.addresses():
            if not addr or own.get_submap(addr).has_value():
                continue                 # the first operand constrains the whole axis itself: that one wins as before
            base = rest.get_submap(addr)
            v = sub[addr]
            if not base.has_value() or isinstance(base._value, Mask) or isinstance(v, Mask):
                continue
            b = base._value
            if isinstance(b, torch.Tensor):
                if b.ndim <= ax or not -b.shape[ax] <= i < b.shape[ax]:
                    continue
                vt = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v), device=b.device)
                vt = vt.to(device=b.device, dtype=b.dtype)
                new = b.clone()
                new.select(ax, i).copy_(vt)
            elif isinstance(b, np.ndarray) and not isinstance(v, torch.Tensor):
                if b.ndim <= ax or not -b.shape[ax] <= i < b.shape[ax]:
                    continue
                new = np.array(b, copy=True)
                new[(slice(None),) * ax + (i,)] = np.asarray(v, dtype=b.dtype)
                new = new.view(type(b)) if type(b) is not np.ndarray else new
            else:
                continue
            rest = rest.set(addr, new)
            out[i] = out[i]._without(addr)
            if out[i].static_is_empty():
                del out[i]
    for a, c in rest._children.items():
        out[a] = c
    return out


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
    """`chm.at["a", "b"].set(v)` / `C["a"].set(v)` / `.update(fn)` / `.v / .d / .kw / .from_mapping` / `.get()`
    (choice_map.py:70-121, 752-845).  Address components are kept RAW until used: `:` marks a plate when setting."""

    def __init__(self, chm, addr=()):
        self.chm, self.addr = chm, tuple(addr)

    def __getitem__(self, addr):
        return _AddressIndex(self.chm, self.addr + (addr if isinstance(addr, tuple) else (addr,)))

    def set(self, v) -> ChoiceMap:
        if isinstance(v, dict):
            v = ChoiceMap.d(v)
        if not self.addr:
            return v if isinstance(v, ChoiceMap) else ChoiceMap.choice(v)
        return self.chm.set(self.addr, v)

    def v(self, val) -> ChoiceMap:
        """like set, but a ChoiceMap argument is stored AS A VALUE (choice_map.py `.v`: "not advisable")"""
        if isinstance(val, ChoiceMap):
            leaf = ChoiceMap(value=val)
            return self.chm.set(self.addr, leaf) if self.addr else leaf
        return self.set(val)

    def update(self, fn) -> ChoiceMap:
        """replace what sits at the address by fn(it): fn gets the VALUE if there is one, else the sub-map"""
        sub = self.chm.get_submap(self.addr)
        new = fn(sub.get_value() if sub.has_value() else sub)
        if isinstance(new, dict):
            new = ChoiceMap.d(new)
        cleared = self.chm._without(_norm(self.addr))
        return cleared.set(self.addr, new) if not (isinstance(new, ChoiceMap) and new.static_is_empty()) else cleared

    def d(self, mapping) -> ChoiceMap:
        return self.set(ChoiceMap.d(mapping))

    from_mapping = d

    def kw(self, **kwargs) -> ChoiceMap:
        return self.set(ChoiceMap.d(kwargs))

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
    set = staticmethod(ChoiceMap.choice)
    v = staticmethod(ChoiceMap.choice)
    d = staticmethod(ChoiceMap.d)
    kw = staticmethod(ChoiceMap.kw)
    choice = staticmethod(ChoiceMap.choice)
    empty = staticmethod(ChoiceMap.empty)


ChoiceMapBuilder = _ChoiceMapBuilder()
ChoiceMap.builder = ChoiceMapBuilder
