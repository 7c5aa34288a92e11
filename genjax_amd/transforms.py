"""`jit` / `vmap` shims for inference scripts written against JAX.

Every generative-function-interface method in this package is already batch
polymorphic: a batched `Key` (and tensors whose leading axes match it) run as
one fused launch over all particles.  `vmap(f, in_axes)` therefore only checks
the mapped axis sizes and calls `f` on the batched values; `jit` is the
identity (programs are compiled and cached per call signature on first use).
Replaces `jax.jit(jax.vmap(...))` in e.g. README.md:95-118 of the reference.
"""
from __future__ import annotations

import torch

from .random import Key


def jit(f=None, **_kw):
    if f is None:
        return lambda g: g
    return f


def vmap(f, in_axes=0, out_axes=0):
    def wrapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        if len(axes) != len(args):
            raise ValueError("vmap: in_axes does not match the number of arguments")
        size = None
        for a, ax in zip(args, axes):
            if ax is None:
                continue
            if ax != 0:
                raise NotImplementedError("vmap: only axis 0 (or None) is supported")
            n = a.shape[0] if isinstance(a, (Key, torch.Tensor)) else None
            if n is not None:
                if size is not None and n != size:
                    raise ValueError(f"vmap: mapped axis sizes differ ({size} vs {n})")
                size = n
        return f(*args)
    return wrapped
