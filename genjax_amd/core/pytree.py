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
        return self.val

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


PythonicPytree = Pytree


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
    return x[idx]
