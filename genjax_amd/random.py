"""PRNG keys with jax.random semantics (Threefry-2x32, `threefry_partitionable`
mode of jax 0.5.2: SURVEY.md App. A.2), without materialising per-particle key
arrays: `split(key, 1_000_000)` is a lazy object that the kernels expand in
registers from the global particle index.

Replaces `jax.random.key / split / fold_in` as used at
src/genjax/_src/inference/smc.py:154,171,299-300,386 and
src/genjax/_src/generative_functions/static.py:261,350.
"""
from __future__ import annotations

from ctypes import c_uint32

import numpy as np
import torch

from . import _lib

_HOST_LIMIT = 1 << 14     # batches up to this many keys stay on the host as numpy


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


_M32 = 0xFFFFFFFF
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _threefry_int(k0: int, k1: int, c0: int, c1: int):
    """One block on the host.  Through the library's own gmx_threefry2x32_host (include/genmi.h) when a backend is
    active — ~1 us — else in Python integers (~10 us; the numpy form costs ~100 us of dispatch for a handful of keys,
    and a functional SMC step derives three or four keys on the host)."""
    be = _lib._backend
    if be is not None:
        out = (c_uint32 * 2)()
        be.c.gmx_threefry2x32_host(k0, k1, c0, c1, out)
        return int(out[0]), int(out[1])
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (c0 + ks[0]) & _M32, (c1 + ks[1]) & _M32
    for g in range(5):
        for r in _ROT[g & 1]:
            x0 = (x0 + x1) & _M32
            x1 = ((x1 << r) | (x1 >> (32 - r))) & _M32
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M32
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & _M32
    return x0, x1


def threefry2x32(k0, k1, c0, c1):
    """Host Threefry-2x32-20: Python integers for up to 16 blocks, vectorised numpy uint32 beyond."""
    k0, k1, c0, c1 = (np.asarray(v, dtype=np.uint32) for v in (k0, k1, c0, c1))
    shape = np.broadcast_shapes(k0.shape, k1.shape, c0.shape, c1.shape)
    size = int(np.prod(shape, dtype=np.int64))
    if size <= 16:
        b = [np.broadcast_to(v, shape).reshape(-1).tolist() for v in (k0, k1, c0, c1)]
        o = [_threefry_int(*(int(v[j]) for v in b)) for j in range(size)]
        return (np.array([p[0] for p in o], dtype=np.uint32).reshape(shape),
                np.array([p[1] for p in o], dtype=np.uint32).reshape(shape))
    with np.errstate(over="ignore"):
        ks = (k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA))
        x0 = c0 + ks[0]
        x1 = c1 + ks[1]
        rot = ((13, 15, 26, 6), (17, 29, 16, 24))
        for g in range(5):
            for r in rot[g & 1]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + np.uint32(g + 1)
    return x0, x1


def _derive_host(keys: np.ndarray, ctr) -> np.ndarray:
    if keys.size == 2 and np.size(ctr) <= 16:
        # ONE key, a few children (fold_in, split(key, 3)): straight Python integers, one array at the end
        k0, k1 = int(keys.reshape(-1)[0]), int(keys.reshape(-1)[1])
        cs = [int(c) for c in np.asarray(ctr).reshape(-1)]
        out = np.array([_threefry_int(k0, k1, c >> 32, c & _M32) for c in cs], dtype=np.uint32)
        lead = np.broadcast_shapes(keys.shape[:-1], np.shape(ctr))
        return out.reshape(lead + (2,))
    ctr = np.asarray(ctr, dtype=np.uint64)
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    lo = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    o0, o1 = threefry2x32(keys[..., 0], keys[..., 1], hi, lo)
    return np.stack(np.broadcast_arrays(o0, o1), axis=-1).astype(np.uint32)


class Key:
    """A batch of PRNG keys.  Exactly one of the representations is set:

    host      numpy uint32 [*batch, 2]
    dev       torch int32  [*batch, 2]  (raw bits)
    lazy      ("split", base Key (batch ()), n)            -> shape (n,)
              ("rowsplit", rows Key (batch (B,)), inner)   -> shape (B, inner)
    """

    def __init__(self, host=None, dev=None, lazy=None, split_last=False, offset=0):
        if dev is not None and not dev.is_contiguous():
            dev = dev.contiguous()      # (a slice of a split over a batch — `split(keys)[0]`: the kernels read [n, 2] rows)
        if host is not None and not host.flags["C_CONTIGUOUS"]:
            host = np.ascontiguousarray(host)
        self._host, self._dev, self._lazy = host, dev, lazy
        self._offset = int(offset)      # lazy "split" only: this is children [offset, offset + n) of the base key
        # result of split() over a BATCH of keys: `a, b = split(keys)` and
        # `split(keys)[i]` address the split axis (the last one), which is what
        # the same code sees per instance under jax.vmap
        self._split_last = split_last

    # -- shape -----------------------------------------------------------
    @property
    def shape(self):
        if self._host is not None:
            return tuple(self._host.shape[:-1])
        if self._dev is not None:
            return tuple(self._dev.shape[:-1])
        kind, base, n = self._lazy
        return (n,) if kind == "split" else base.shape + (n,)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"Key(shape={self.shape})"

    # -- materialisation -----------------------------------------------------
    def host(self) -> np.ndarray:
        """numpy uint32 [*batch, 2] (computes / downloads if necessary)."""
        if self._host is not None:
            return self._host
        if self._dev is not None:
            return self._dev.cpu().numpy().view(np.uint32)
        kind, base, n = self._lazy
        bh = base.host()
        return _derive_host(bh[..., None, :], np.arange(n, dtype=np.uint64) + np.uint64(self._offset))

    def data(self) -> torch.Tensor:
        """device int32 [size, 2] (flattened batch), materialising lazily."""
        be = _lib.get()
        if self._dev is not None:
            return self._dev.reshape(-1, 2)
        if self._host is not None:
            t = torch.from_numpy(np.ascontiguousarray(self._host).view(np.int32).reshape(-1, 2))
            return t.to(be.device)
        kind, base, n = self._lazy
        if kind == "split":
            bh = base.host()
            out = torch.empty((n, 2), dtype=torch.int32, device=be.device)
            kk = (c_uint32 * 2)(int(bh[0]), int(bh[1]))
            be.check(be.c.gmx_split(kk, n, self._offset, be.ptr(out), be.stream()), "gmx_split")
            return out
        rows = base.data()
        out = torch.empty((rows.shape[0] * n, 2), dtype=torch.int32, device=be.device)
        be.check(be.c.gmx_split_rows(be.ptr(rows), rows.shape[0], n, be.ptr(out), be.stream()),
                 "gmx_split_rows")
        return out

    def materialize(self) -> "Key":
        if self._lazy is None:
            return self
        if self.size <= _HOST_LIMIT:
            return Key(host=self.host(), split_last=self._split_last)
        return Key(dev=self.data().reshape(self.shape + (2,)), split_last=self._split_last)

    def reshape(self, *shape) -> "Key":
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        m = self.materialize()
        if m._host is not None:
            return Key(host=m._host.reshape(shape + (2,)))
        return Key(dev=m._dev.reshape(shape + (2,)))

    # -- indexing ----------------------------------------------------------
    def __getitem__(self, idx):
        if self._split_last and isinstance(idx, (int, np.integer)) and len(self.shape) > 1:
            m = self.materialize()
            if m._host is not None:
                return Key(host=m._host[..., int(idx), :])
            return Key(dev=m._dev[..., int(idx), :])
        if self._lazy is not None:
            kind, base, n = self._lazy
            if kind == "split" and isinstance(idx, (int, np.integer)):
                i = int(idx) % n
                return Key(host=_derive_host(base.host(), np.uint64(i + self._offset)))
            return self.materialize()[idx]
        if not isinstance(idx, tuple):
            idx = (idx,)
        if self._host is not None:
            return Key(host=self._host[idx + (slice(None),)])
        return Key(dev=self._dev[idx + (slice(None),)])

    def __iter__(self):
        n = self.shape[-1] if self._split_last else self.shape[0]
        for i in range(n):
            yield self[i]

    # -- binding for gmx_program_run ----------------------------------------
    def binding(self):
        """(key_mode, key0, key1, keys tensor or None, inner) for the flattened batch."""
        if self._lazy is not None:
            kind, base, n = self._lazy
            if kind == "split":
                bh = base.host()
                return _lib.KEY_SPLIT, int(bh[0]), int(bh[1]), None, 0
            return _lib.KEY_ROWSPLIT, 0, 0, base.data(), n
        if self.shape == ():
            h = self.host()
            return _lib.KEY_BCAST, int(h[0]), int(h[1]), None, 0
        return _lib.KEY_ARRAY, 0, 0, self.data(), 0


def key(seed: int) -> Key:
    """jax.random.key(seed): key data (0, seed) for 0 <= seed < 2**32."""
    seed = int(seed)
    return Key(host=np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32))


PRNGKey = key


def split(k: Key, num: int = 2) -> Key:
    """jax.random.split: child i = threefry(key, counter i); shape k.shape + (num,)."""
    num = int(num)
    if k._lazy is not None:
        k = k.materialize()
    if k.shape == ():
        if num <= _HOST_LIMIT:
            return Key(host=_derive_host(k.host()[None, :], np.arange(num, dtype=np.uint64)))
        return Key(lazy=("split", k, num))
    if k.size * num <= _HOST_LIMIT and k._host is not None:
        return Key(host=_derive_host(k._host[..., None, :], np.arange(num, dtype=np.uint64)), split_last=True)
    flat = k.reshape((k.size,))
    out = Key(lazy=("rowsplit", flat, num), split_last=True)
    if len(k.shape) == 1:
        return out
    out = out.reshape(k.shape + (num,))
    out._split_last = True
    return out


def stack_keys(keys) -> Key:
    """a list of single keys (what iterating / unpacking a split gives) as ONE batch of keys: `jnp.array(sub_keys)`"""
    rows = [np.asarray(k.host(), dtype=np.uint32).reshape(-1, 2) for k in keys]
    return Key(host=np.concatenate(rows, axis=0))


def lazy_split(k: Key, num: int, offset: int = 0) -> Key:
    """split(k, num) for a single key, never materialised: kernels derive child
    i in registers from the global particle index (GMX_KEY_SPLIT).  With `offset`, the `num` keys are
    children offset .. offset+num-1 (a rank's block of a larger split)."""
    if k.shape != ():
        raise ValueError("lazy_split needs a single key")
    return Key(lazy=("split", k.materialize(), int(num)), offset=offset)


def fold_in(k: Key, data: int) -> Key:
    """jax.random.fold_in(key, data) = threefry(key, counter data)."""
    data = int(data) & 0xFFFFFFFF
    if k._lazy is not None:
        k = k.materialize()
    if k._host is not None:
        return Key(host=_derive_host(k._host, np.uint64(data)))
    be = _lib.get()
    flat = k.data()
    out = torch.empty_like(flat)
    be.check(be.c.gmx_fold_in(be.ptr(flat), data, flat.shape[0], be.ptr(out), be.stream()), "gmx_fold_in")
    return Key(dev=out.reshape(k.shape + (2,)))


def key_data(k: Key) -> np.ndarray:
    return k.host()


def _draw(dist_name: str, k: Key, shape, a, b):
    """`jax.random.uniform / normal(key, shape)`: what the distribution site of that name draws under `key` — element j of
    a vector on counter j (the same kernels the `@gen` sites run, never a host generator)."""
    from . import distributions as D
    from .static import run_gfi
    dist = getattr(D, dist_name)
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,))) if shape not in ((), None) else ()
    if shape == ():
        return run_gfi(dist, "simulate", k, (a, b)).get_retval()
    if len(shape) != 1:
        raise NotImplementedError("random.uniform / random.normal: shape () or (n,)")
    import torch
    n = shape[0]
    dev = _lib.get().device
    lo = torch.full((n,), float(a), dtype=torch.float32, device=dev)
    hi = torch.full((n,), float(b), dtype=torch.float32, device=dev)
    from .engine import Broadcast
    return run_gfi(dist, "simulate", k, (Broadcast(lo), Broadcast(hi))).get_retval()


def uniform(k: Key, shape=(), minval: float = 0.0, maxval: float = 1.0):
    """jax.random.uniform(key, shape, minval=0, maxval=1) — `jnp.log(jax.random.uniform(subkey)) < alpha`, the accept test
    of docs/cookbook/inactive/inference/mcmc.ipynb c8 and 3_speed_gains.ipynb c15"""
    return _draw("uniform", k, shape, minval, maxval)


def normal(k: Key, shape=()):
    """jax.random.normal(key, shape)"""
    return _draw("normal", k, shape, 0.0, 1.0)
