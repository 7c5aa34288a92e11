"""Enumerative Gibbs update of a discrete address (BASELINE config 5).

Two entry points:

  enumerative_gibbs(key, plate_trace, addr, K)   the Gibbs move on the trace of a PLATE (`generate_datapoint.vmap()` /
      `.repeat(n=N)` called directly: its elements run on the launch axis, combinators.Vmap._launch_axis): every
      element's `addr` is redrawn from its exact conditional and the trace is updated — the notebook's
      `update_datapoint_assignment` for a model written with the Vmap combinator;
  gibbs_categorical(key, gen_fn, args, choices, addr, K)   the fused draw underneath it: the lowering of
      `categorical.simulate(key, vmap(vmap(assess)))` to ONE launch.

Reference idiom (docs/cookbook/inactive/update/7_application_dirichlet_mixture_model.ipynb,
cell 10, `update_datapoint_assignment`):

    local_densities = vmap(lambda x: vmap(lambda i: gen_fn.assess(chm(x, i), args)[0])(arange(K)))(arange(N))
    new_idx = genjax.categorical.simulate(key, (local_densities,)).get_choices()

i.e. ONE key for the whole [N, K] matrix of logits: row i, category k draws its Gumbel from
counter i*K + k, first maximum wins.  `gibbs_categorical` computes exactly
that arithmetic in ONE launch without materialising the [N, K] matrix (256 MB at N = 1e6,
K = 64): each thread scores its datapoint under every category — the model's own `assess`,
traced K times with the category fixed, common sub-expressions shared — and keeps a running
Gumbel-max.
"""
from __future__ import annotations

import torch

from .. import _lib
from .. import tracer as T
from ..core.choice_map import ChoiceMap
from ..engine import Compiled, Flat, Tracing, leaf_spec, resolve, unflatten
from ..random import Key
from ..tracer import Expr
from ..engine import new_cache as _new_program_cache

_CACHE = _new_program_cache()


def gibbs_categorical(key: Key, gen_fn, args, choices: ChoiceMap, addr, n_categories: int, batch_shape=None,
                      index_offset: int = 0):
    """idx[i] = argmax_k( gen_fn.assess(choices_i with addr := k, args)[0] + gumbel(bits(key, i*K + k)) ).

    `choices` / `args` hold per-datapoint tensors (leading shape = the batch) and launch-uniform
    values; `key` is one (unbatched) key.  Returns an int32 tensor of shape `batch`.

    index_offset: the GLOBAL index of this call's first datapoint — datapoints are independent, so a
    dataset sharded over ranks (or processed in chunks) gives the single-call result when every shard
    passes its offset (the Gumbel counter is (index_offset + i)*K + k); no collective is involved."""
    from ..static import _Ctx, _gfkey, _infer_batch, _sym_constraint, call_gen_fn
    be = _lib.get()
    if tuple(key.shape) != ():
        raise ValueError("gibbs_categorical takes ONE key for the whole batch (categorical.simulate semantics)")
    K = int(n_categories)
    flat = Flat()
    atree = flat.add(tuple(args))
    n_args = len(flat.leaves)
    ctree = flat.add(choices)
    # the batch is the datapoint axis of the per-datapoint choices (args are usually launch-uniform)
    batch = tuple(batch_shape) if batch_shape is not None else _infer_batch(flat.leaves[n_args:])
    if len(batch) != 1:
        raise NotImplementedError("gibbs_categorical: one batch axis")
    specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
    addr_t = addr if isinstance(addr, tuple) else (addr,)
    ck = (_gfkey(gen_fn), atree, ctree, specs, addr_t, K)
    ent = _CACHE.get(ck)
    if ent is None:
        tr = Tracing(len(batch))
        ctx = _Ctx(tr)
        ctx.store_sites = False
        g = tr.graph
        with T.tracing(g):
            syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
            sargs = unflatten(atree, lambda j: syms[j].value)
            scon = _sym_constraint(ctree, syms)
            kx = g.add("LDKEY", dtype="key")
            base = Expr(g.add("LDIDX", dtype="i32")) * K            # row counter base i*K
            state = None
            for k in range(K):
                _, _, _, s = call_gen_fn(ctx, "assess", gen_fn, None, sargs, scon.set(addr_t, k), None, None, None, ())
                state = g.add("S_CATSTEP", (state, kx, T.as_float(s).node, (base + k).node), imm=k, dtype="cat")
            out = tr.emit_output(Expr(g.add("CATIDX", (state,), dtype="i32")))
        ent = (Compiled(tr), out)
        _CACHE[ck] = ent
    comp, out = ent
    outs = comp.run(flat.leaves, batch, key, index_offset=int(index_offset))
    return resolve(out, outs, flat.leaves)


def enumerative_gibbs(key: Key, trace, addr, n_categories: int):
    """One enumerative Gibbs move on `addr` of EVERY element of a plate, through the GFI (the reference's
    `update_datapoint_assignment`, 7_application_dirichlet_mixture_model.ipynb c10, for a model whose datapoints are a
    `Vmap` plate): `trace` is the trace of `inner.vmap(...)` / `inner.repeat(n=...)` called directly (one key: its
    elements are the launch axis).  As the notebook does:

        key, subkey = split(key)
        local_densities[j, k] = inner.assess(choices_j with addr := k, args_j)[0]        (never materialised)
        new = categorical.simulate(key, (local_densities,))      ONE key: (j, k) draws its Gumbel from counter j K + k
        new_trace = trace.update(subkey, C[addr].set(new))

    Returns (new trace, the new values of `addr`, the update's weight)."""
    from ..combinators import Vmap
    from ..core.generative import Diff
    from ..random import split
    from ..static import VmapTrace
    vm = trace.get_gen_fn()
    if not isinstance(vm, Vmap) or not isinstance(trace, VmapTrace) or tuple(trace.batch_shape) != ():
        raise TypeError("enumerative_gibbs: the trace of a Vmap / repeat plate called directly under one key")
    args = tuple(trace.get_args() or ())
    la = vm._launch_axis(None, args, None, ())
    if la is None:
        raise NotImplementedError("enumerative_gibbs: the plate is too small for the launch-axis form "
                                  "(fewer elements than combinators.VMAP_LAUNCH_MIN)")
    n, inner_args = la
    key, subkey = split(key)
    new = gibbs_categorical(key, vm.gen_fn, inner_args, trace.get_choices(), addr, n_categories, batch_shape=(n,))
    addr_t = addr if isinstance(addr, tuple) else (addr,)
    new_trace, w, _, _ = trace.update(subkey, ChoiceMap.empty().set(addr_t, new), Diff.no_change(args))
    return new_trace, new, w
