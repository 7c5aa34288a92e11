"""Mask: a value paired with a validity flag (functional_types.py:42-368).
Only the host-side container is provided on the round-1 hot path; masked
constraints (`Indexed` / `Switch` choice maps) are next-tier."""
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


def _and(a, b):
    if isinstance(a, bool) and isinstance(b, bool):
        return a and b
    return a & b
