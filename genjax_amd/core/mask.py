"""Mask: a value paired with a validity flag (functional_types.py:42-368).

As a CONSTRAINT leaf (`C["x"].set(Mask(value, flag))`) it is the reference's runtime-conditional constraint:
`Distribution.generate_choice_map` runs `lax.cond(flag, importance, simulate)` (distribution.py:129-142) and
`edit_update_with_constraint` `FlagOp.cond(flag, new value, old value)` (:189-224).  Here the flag may be a Python
bool (decided while the program is traced) or a launch value — one bool per particle — in which case the leaf's
site program computes both branches' value and selects (OP_SEL): static._leaf_call, engine.Flat."""
from __future__ import annotations


class Mask:
    __slots__ = ("value", "flag")

    def __init__(self, value, flag):
        self.value, self.flag = value, flag

    @staticmethod
    def build(value, flag=True):
        if isinstance(value, Mask):
            return Mask(value.value, _and(flag, value.flag))
        return Mask(value, flag)

    def primal_flag(self):
        return self.flag

    def unmask(self):
        if self.flag is False:
            raise ValueError("Attempted to unmask when a mask flag is False: the masked value is invalid.")
        return self.value

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
