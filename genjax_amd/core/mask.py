"""Mask: a value paired with a validity flag (functional_types.py:42-368).

As a CONSTRAINT leaf (`C["x"].set(Mask(value, flag))`) it is the reference's runtime-conditional constraint:
`Distribution.generate_choice_map` runs `lax.cond(flag, importance, simulate)` (distribution.py:129-142) and
`edit_update_with_constraint` `FlagOp.cond(flag, new value, old value)` (:189-224).  Here the flag may be a Python
bool (decided while the program is traced) or a launch value — one bool per particle — in which case the leaf's
site program computes both branches' value and selects (OP_SEL): static._leaf_call, engine.Flat."""
from __future__ import annotations


class Mask:
    __slots__ = ("value", "flag")

    def __init__(self, value, flag=True):
        _check_prefix(value, flag)
        self.value, self.flag = value, flag

    @staticmethod
    def build(value, flag=True):
        """functional_types.py:148-173: a Mask of a Mask is ONE Mask whose flag is the conjunction"""
        if isinstance(value, Mask):
            fs, gs = _shape(flag), _shape(value.flag)
            if fs != () and fs != gs and gs[:len(fs)] == fs:
                # a flag per step / element over an inner mask per (step, inner element): the reference builds this
                # mask INSIDE the scan / plate, a scalar flag against the inner flags, and stacks; here the stacked
                # leaves meet, so the outer flag is spread over the inner axes
                flag = flag.reshape(fs + (1,) * (len(gs) - len(fs)))
                fs = gs
            assert fs == () or fs == gs, f"Can't build a Mask with non-matching Flag shapes {fs} and {gs}"
            return Mask(value.value, _and(flag, value.flag))
        return Mask(value, flag)

    @staticmethod
    def maybe_mask(value, flag):
        """functional_types.py:175-192: the value itself under a Python True, None under a Python False, a Mask
        otherwise"""
        return Mask.build(value, flag).flatten()

    def flatten(self):
        """functional_types.py:220-243"""
        if self.flag is False:
            return None
        if self.flag is True:
            return self.value
        return self

    def primal_flag(self):
        return self.flag

    def unmask(self, default=None):
        """functional_types.py:245-275: without a default the flag must hold (everywhere, for a vectorised mask);
        with one, the default stands in wherever it does not"""
        if default is None:
            if not _all(self.flag):
                raise ValueError("Attempted to unmask when a mask flag (or some flag in a vectorized mask) is False: "
                                 "the unmasked value is invalid.")
            return self.value
        if isinstance(self.flag, bool):
            return self.value if self.flag else default
        return _tree_map2(lambda a, b: _where(self.flag, a, b), self.value, default)

    def __getitem__(self, path):
        """functional_types.py:196-218: the whole path indexes the value; a vectorised flag takes only as many
        components of it as it has axes"""
        path = path if isinstance(path, tuple) else (path,)
        f = self.flag
        if _shape(f) != ():
            f = f[path[:len(_shape(f))]]
        return Mask.build(_tree_map1(lambda v: v[path], self.value), f)

    def _check_same_form(self, other):
        if _structure(self.value) != _structure(other.value):
            raise ValueError("Cannot combine masks with different tree structures!")

        def chk(a, b):
            if _shape(a) != _shape(b):
                raise ValueError(f"Cannot combine masks with different array shapes: {_shape(a)} vs {_shape(b)}")
        _tree_map2(chk, self.value, other.value)
        chk(self.flag, other.flag)

    def __or__(self, other):
        """functional_types.py:316-326: the first valid side, element by element for vectorised flags"""
        self._check_same_form(other)
        a, b = self.flag, other.flag
        if a is True:
            return self
        if a is False:
            return other
        take_b = _and(_not(a), b)
        return Mask(_tree_map2(lambda x, y: _where(take_b, y, x), self.value, other.value), _or(a, b))

    def __xor__(self, other):
        """functional_types.py:328-346: valid where exactly one side is"""
        self._check_same_form(other)
        a, b = self.flag, other.flag
        if isinstance(a, bool) and isinstance(b, bool):
            if a == b:
                return Mask.build(self, False)
            return self if a else other
        take_b = _and(_not(a), b)
        return Mask(_tree_map2(lambda x, y: _where(take_b, y, x), self.value, other.value), _xor(a, b))

    def __invert__(self):
        return Mask(self.value, _not(self.flag))

    @staticmethod
    def or_n(mask, *masks):
        for m in masks:
            mask = mask | m
        return mask

    @staticmethod
    def xor_n(mask, *masks):
        for m in masks:
            mask = mask ^ m
        return mask

    def __repr__(self):
        return f"Mask({self.value!r}, {self.flag!r})"

    def __eq__(self, other):
        if not isinstance(other, Mask):
            return NotImplemented
        return _same(self.value, other.value) and _same(self.flag, other.flag)

    __hash__ = object.__hash__


class Indexed:
    """A constraint on ONE element of a plate chosen at run time: `C["ys", idx, "y"].set(v)` with `idx` a per-particle
    index (the reference's `Indexed` choice map with a traced index, choice_map.py:1453-1531): element idx of the
    plate's "y" takes the value v.  Inside the plate it is the masked constraint `Mask(v, idx == j)` for element j
    (`Indexed.get_inner_map`, :1508-1531) — resolved in combinators._index_chm / _loop_step_constraint."""
    __slots__ = ("value", "idx")

    def __init__(self, value, idx):
        self.value, self.idx = value, idx

    def __repr__(self):
        return f"Indexed({self.value!r} at {self.idx!r})"


def _same(a, b):
    try:
        import numpy as np
        import torch
        ta = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        tb = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
        return bool(np.array_equal(np.asarray(ta), np.asarray(tb)))
    except Exception:
        return a == b


def _and(a, b):
    if isinstance(a, bool) and isinstance(b, bool):
        return a and b
    return a & b


def _or(a, b):
    if isinstance(a, bool) and isinstance(b, bool):
        return a or b
    return a | b


def _xor(a, b):
    if isinstance(a, bool) and isinstance(b, bool):
        return a != b
    return a ^ b


def _not(a):
    return (not a) if isinstance(a, bool) else ~a


def _all(a):
    if isinstance(a, bool):
        return a
    try:
        return bool(a.all())
    except Exception:           # a traced flag: nothing to check while the program is traced
        return True


def _shape(x):
    """the array shape of a concrete leaf; () for Python scalars and for anything traced"""
    sh = getattr(x, "shape", None)
    try:
        return tuple(int(d) for d in sh) if sh is not None else ()
    except Exception:
        return ()


def _where(flag, a, b):
    import torch
    if isinstance(flag, bool):
        return a if flag else b
    if isinstance(flag, torch.Tensor):
        ta = torch.as_tensor(a, device=flag.device)
        f = flag.reshape(tuple(flag.shape) + (1,) * max(0, ta.dim() - flag.dim()))
        return torch.where(f, ta, torch.as_tensor(b, device=flag.device))
    import numpy as np
    return np.where(flag, a, b)


def _children(v):
    if isinstance(v, dict):
        return list(v.keys()), list(v.values())
    if isinstance(v, (list, tuple)):
        return list(range(len(v))), list(v)
    return None, None


def _structure(v):
    keys, kids = _children(v)
    if keys is None:
        return "*"
    return (type(v).__name__, tuple((k, _structure(c)) for k, c in zip(keys, kids)))


def _tree_map1(fn, v):
    keys, kids = _children(v)
    if keys is None:
        return fn(v)
    out = [_tree_map1(fn, c) for c in kids]
    return dict(zip(keys, out)) if isinstance(v, dict) else type(v)(out)


def _tree_map2(fn, v, w):
    keys, kids = _children(v)
    if keys is None:
        return fn(v, w)
    _, kw = _children(w)
    out = [_tree_map2(fn, c, d) for c, d in zip(kids, kw)]
    return dict(zip(keys, out)) if isinstance(v, dict) else type(v)(out)


def _leaves(v):
    keys, kids = _children(v)
    if keys is None:
        return [v]
    return [x for c in kids for x in _leaves(c)]


def _check_prefix(value, flag):
    """functional_types.py:75-103: a flag with a shape marks a vectorised mask, and every array leaf of the value must
    carry that shape as a prefix of its own.  Scalars (Python numbers, 0-d arrays) are exempt here: in this package a
    scalar next to a per-particle flag is a launch-uniform value, not a mis-shaped one."""
    fs = _shape(flag)
    if fs == ():
        return
    for leaf in _leaves(value):
        ls = _shape(leaf)
        if ls == ():
            continue
        if ls[:len(fs)] != fs:
            raise ValueError(f"Vectorized flag's shape {fs} must be a prefix of all leaf shapes. Found {ls}.")
