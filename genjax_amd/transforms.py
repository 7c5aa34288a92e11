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


def _map_leaves(v, fn):
    from .core.choice_map import ChoiceMap
    if isinstance(v, torch.Tensor):
        return fn(v)
    if isinstance(v, tuple):
        return tuple(_map_leaves(x, fn) for x in v)
    if isinstance(v, list):
        return [_map_leaves(x, fn) for x in v]
    if isinstance(v, dict):
        return {k: _map_leaves(x, fn) for k, x in v.items()}
    if isinstance(v, ChoiceMap):
        return v.map_values(lambda x: _map_leaves(x, fn))
    return v


def _loop_instances(f, args, axes, n, first_error):
    from .core.choice_map import ChoiceMap
    from .engine import Broadcast, Mapped

    def at(a, ax, i):
        if ax is None:
            return _map_leaves(a, lambda t: t.plain if isinstance(t, Broadcast) else t)
        if isinstance(a, Key):
            return a[i]
        if not isinstance(a, torch.Tensor) and hasattr(a, "shape") and hasattr(a, "dtype") and len(a.shape) >= 1 \
                and getattr(a.dtype, "kind", "O") in "fiub":
            import numpy as _np
            v = _np.asarray(a)[i]
            return v.item() if _np.ndim(v) == 0 else v
        return _map_leaves(a, lambda t: (t.plain if isinstance(t, Mapped) else t)[i])
    outs = []
    for i in range(int(n)):
        try:
            outs.append(f(*[at(a, ax, i) for a, ax in zip(args, axes)]))
        except Exception:
            raise first_error from None          # not a batching problem: the function fails per instance too

    def stack(vs):
        v0 = vs[0]
        if isinstance(v0, torch.Tensor):
            return torch.stack([v.plain if isinstance(v, (Broadcast, Mapped)) else v for v in vs])
        if isinstance(v0, (int, float, bool)):
            return torch.tensor(vs)
        if isinstance(v0, tuple):
            return tuple(stack([v[k] for v in vs]) for k in range(len(v0)))
        if isinstance(v0, list):
            return [stack([v[k] for v in vs]) for k in range(len(v0))]
        if isinstance(v0, dict):
            return {k: stack([v[k] for v in vs]) for k in v0}
        if isinstance(v0, ChoiceMap):
            out = ChoiceMap.empty()
            for a in v0.addresses():
                out = out.set(a, stack([v[a] for v in vs])) if a else ChoiceMap.choice(stack([v.get_value() for v in vs]))
            return out
        return v0
    return stack(outs)


def vmap(f=None, in_axes=0, out_axes=0):
    """Two spellings share this name.  `genjax.vmap(in_axes=...)` with no function (or a generative function) is the
    reference's COMBINATOR decorator (combinators/vmap.py `vmap`): `@genjax.vmap(in_axes=(0,))` above `@genjax.gen`.
    `vmap(f, in_axes, out_axes)` with an ordinary function stands in for `jax.vmap` in inference scripts:

    `jax.vmap(f, in_axes, out_axes)` for functions built from this package's operations.  Those are batch
    polymorphic already — a batched `Key` and tensors whose LEADING axis matches it run as one fused launch — so
    mapping = arranging the arguments that way:
      * an argument mapped along axis k != 0 has that axis moved to the front (every tensor leaf of a tuple / dict /
        ChoiceMap argument);
      * an argument with in_axes None is shared by all instances: its tensors are marked launch-uniform
        (engine.Broadcast), so a vector whose length happens to equal the batch is not mistaken for per-instance data;
      * mapped axis sizes are checked against each other;
      * out_axes != 0 moves the leading axis of every tensor result there."""
    from .core.generative import GenerativeFunction
    if f is None:
        from .combinators import vmap as _combinator
        return _combinator(in_axes=in_axes)
    if isinstance(f, GenerativeFunction):
        from .combinators import Vmap
        return Vmap(f, in_axes)
    from .engine import Broadcast, Mapped

    def wrapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        if len(axes) != len(args):
            raise ValueError("vmap: in_axes does not match the number of arguments")
        sizes = set()

        def front(ax):
            def go(t):
                if t.ndim == 0:
                    raise ValueError("vmap: cannot map over a 0-d tensor (use in_axes=None)")
                sizes.add(int(t.shape[ax]))
                t = torch.movedim(t, ax, 0) if ax != 0 else t
                # integer tensors keep a tag: as an address component they are one RUN-TIME index per instance
                return Mapped(t) if t.dtype in (torch.int32, torch.int64) else t
            return go
        new = []
        for a, ax in zip(args, axes):
            if ax is None:
                new.append(_map_leaves(a, lambda t: Broadcast(t) if t.ndim >= 1 else t))
            elif isinstance(a, Key):
                if ax != 0:
                    raise NotImplementedError("vmap: a Key is mapped along its leading axis")
                sizes.add(int(a.shape[0]))
                new.append(a)
            else:
                new.append(_map_leaves(a, front(int(ax))))
        if len(sizes) > 1:
            raise ValueError(f"vmap: mapped axis sizes differ ({sorted(sizes)})")
        try:
            out = f(*new)
        except (RuntimeError, TypeError, IndexError) as e:      # (a ValueError / AssertionError is the function's own refusal)
            # `f` is not batch polymorphic as written — host glue that compares a mapped index with unmapped data
            # (`jax.vmap(lambda i: jnp.sum(jnp.where(idx == i, xs, 0)))(jnp.arange(k))`, the mixture notebook's c10): the
            # instances one by one, results stacked along a new leading axis (what jax.vmap returns)
            n_inst = set(sizes)
            for a, ax in zip(args, axes):            # (a HOST array mapped over — `jax.vmap(f)(jnp.arange(k))` — counts here)
                if ax is not None and not isinstance(a, (Key, torch.Tensor)) and hasattr(a, "shape") and len(a.shape) >= 1 \
                        and getattr(getattr(a, "dtype", None), "kind", "O") in "fiub":
                    n_inst.add(int(a.shape[0]))
            if len(n_inst) != 1:
                raise
            out = _loop_instances(f, new, axes, n_inst.pop(), e)
        if out_axes not in (0, None):
            out = _map_leaves(out, lambda t: torch.movedim(t, 0, int(out_axes)) if t.ndim > int(out_axes) else t)
        return out
    return wrapped
