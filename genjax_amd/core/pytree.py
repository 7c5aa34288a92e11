"""The slice of `genjax.core.pytree` (src/genjax/_src/core/pytree.py:40-334) models and scripts touch:
`Const` (a static value riding in an argument tuple), `Pytree` (dataclass sugar), `Closure`, `nth`.
There is no jax pytree machinery underneath: launch values are classified by engine.leaf_spec."""
from __future__ import annotations

import dataclasses
from typing import Any, Generic, TypeVar

R = TypeVar("R")


class Const(Generic[R]):
    """A host-side constant passed where an argument is expected (`Const(40)`; `.unwrap()` /
    `.val` inside the model) — never a launch value (pytree.py:271-306)."""
    __gmx_static__ = True

    def __init__(self, val):
        self.val = val

    def unwrap(self):
        """`c.unwrap()`; `Const.unwrap(v)` on a plain value hands it back (pytree.py:290-306)"""
        return self.val if isinstance(self, Const) else self

    def __class_getitem__(cls, item):
        return cls

    def __eq__(self, other):
        return isinstance(other, Const) and self.val == other.val

    def __hash__(self):
        return hash(("Const", self.val)) if not isinstance(self.val, (list, dict)) else id(self)

    def __repr__(self):
        return f"Const({self.val!r})"


class Pytree:
    """`@Pytree.dataclass` / `Pytree.static()` / `Pytree.field()` (pytree.py:40-205) as plain dataclasses."""

    @staticmethod
    def dataclass(cls=None, /, **kwargs):
        kwargs.pop("match_args", None)
        def wrap(c):
            return dataclasses.dataclass(c, **kwargs)
        return wrap if cls is None else wrap(cls)

    @staticmethod
    def static(**kwargs):
        md = dict(kwargs.pop("metadata", None) or {})
        md["static"] = True                   # rides in the structure, not among the leaves (engine.Flat)
        return dataclasses.field(metadata=md, **kwargs)

    @staticmethod
    def field(**kwargs):
        return dataclasses.field(**kwargs)

    @staticmethod
    def const(v):
        return Const(v)


class PythonicPytree(Pytree):
    """Bracket indexing, len, iteration, `+` (leaf-wise concatenation) and `prepend` for dataclass pytrees whose leaves
    share a leading axis (pytree.py:342-376)."""

    def __getitem__(self, idx):
        return nth(self, idx)

    def __len__(self):
        return len(_leaves(self)[0])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __add__(self, other):
        if not isinstance(other, type(self)):
            raise TypeError(f"Cannot add {type(self)} and {type(other)}")
        import torch
        return _map2(lambda a, b: torch.cat([torch.as_tensor(a), torch.as_tensor(b)]), self, other)

    def prepend(self, child):
        import torch
        return _map2(lambda a, b: torch.as_tensor(a)[None], child, child) + self


class Closure:
    """A function closed over dynamic arguments (pytree.py:308-334): Closure(args, fn)(x) = fn(*args, x)."""

    def __init__(self, dyn_args: tuple, fn):
        self.dyn_args, self.fn = tuple(dyn_args), fn

    def __call__(self, *args, **kwargs):
        return self.fn(*self.dyn_args, *args, **kwargs)


def nth(x: Any, idx):
    """tree_map(lambda v: v[idx], x) over nested tuples / lists / dicts of arrays (pytree.py `nth`)."""
    if isinstance(x, tuple):
        return tuple(nth(v, idx) for v in x)
    if isinstance(x, list):
        return [nth(v, idx) for v in x]
    if isinstance(x, dict):
        return {k: nth(v, idx) for k, v in x.items()}
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return _map2(lambda a, _b: a[idx], x, x)
    return x[idx]


def _dyn_fields(x):
    return [f for f in dataclasses.fields(x) if not f.metadata.get("static")]


def _leaves(x):
    if isinstance(x, (tuple, list)):
        return [l for v in x for l in _leaves(v)]
    if isinstance(x, dict):
        return [l for v in x.values() for l in _leaves(v)]
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return [l for f in _dyn_fields(x) for l in _leaves(getattr(x, f.name))]
    return [x]


def _map2(fn, x, y):
    if isinstance(x, (tuple, list)):
        return type(x)(_map2(fn, a, b) for a, b in zip(x, y))
    if isinstance(x, dict):
        return {k: _map2(fn, v, y[k]) for k, v in x.items()}
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return dataclasses.replace(x, **{f.name: _map2(fn, getattr(x, f.name), getattr(y, f.name))
                                         for f in _dyn_fields(x)})
    return fn(x, y)
