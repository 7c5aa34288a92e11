"""Sequential Monte Carlo combinators.

Part 1 restates src/genjax/_src/inference/smc.py: ParticleCollection :76-109,
SMCAlgorithm :117-225, Importance :233-279, ImportanceK :282-351,
ChangeTarget :359-465 — same key plumbing, same weight algebra — with the
`jax.vmap` over particles replaced by one fused launch per GFI call.

Part 2 is BUILD-DEFINED (the reference has no resampling strategies, no
extend step, no MH accept: SURVEY.md §0, App. B): scalable resampling
(systematic / stratified / multinomial over an exact integer CDF), a bootstrap
`extend`, an MH `rejuvenate`, and `BootstrapSweep`, the graph-captured
particle-filter sweep bench.py times.  PARITY UNPINNED against the reference
for Part 2; pinned against oracle/ and closed-form Kalman answers.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from ctypes import c_uint32

import numpy as np
import torch

from .. import _lib, engine
from ..core.choice_map import ChoiceMap
from ..core.generative import Diff
from ..engine import Gathered
from ..random import Key, fold_in, lazy_split, split
from ..static import DistributionTrace, StaticTrace, VmapTrace
from .sp import Algorithm, Target


# the float algebra between launches, as traced one-launch programs (engine.elementwise)
def _sub(a, b): return a - b
def _add(a, b): return a + b
def _sub_const(a, c): return a - c
def _reweight(new_w, old_score, w): return new_w - old_score + w      # smc.py:383


# ---------------------------------------------------------------------------
# helpers over batched traces
# ---------------------------------------------------------------------------
def trace_map(tr, fn):
    """Apply fn to every device leaf of a (batched) trace."""
    def leaf(v):
        if isinstance(v, (torch.Tensor, Gathered)):
            return fn(v)
        if isinstance(v, tuple):
            return tuple(leaf(x) for x in v)
        if isinstance(v, list):
            return [leaf(x) for x in v]
        if isinstance(v, dict):
            return {k: leaf(x) for k, x in v.items()}
        return v
    if isinstance(tr, DistributionTrace):
        return DistributionTrace(tr.gen_fn, leaf(tr.args), leaf(engine.materialize(tr.value)), leaf(engine.materialize(tr.score)))
    if isinstance(tr, VmapTrace):          # a plate / scan among the particle's sites: its own score and return value too
        return VmapTrace(tr.gen_fn, trace_map(tr.inner, fn), leaf(engine.materialize(tr.score)),
                         leaf(_materialize_tree(tr.retval)), leaf(tr.args))
    return StaticTrace(tr.gen_fn, leaf(tr.args), leaf(_materialize_tree(tr.retval)),
                       OrderedDict((a, trace_map(s, fn)) for a, s in tr.subtraces.items()))


def _materialize_tree(v):
    from ..static import _tree_materialize
    return _tree_materialize(v)


def stack_traces(trs, tr_one, axis: int = 0):
    """tree_map(stack_to_first_dim, trs, tr_one) (smc.py:56-68, :343-345): append ONE trace after the K-1 particles
    of a batched trace along the particle axis — axis 0 for one key, axis `len(key.shape)` under a batch of keys
    (the reference's `vmap` of run_csmc: leaves [*keys, K-1, ...] and [*keys, ...])."""
    def cat(a, b):
        a = engine.materialize(a)
        b = engine.materialize(b)
        if not isinstance(a, torch.Tensor):
            return a                                   # static argument (same in both traces)
        # (a retained trace built from HOST tensors joins device-resident particles: same device, same element type)
        b = torch.as_tensor(b, dtype=a.dtype, device=a.device) if not isinstance(b, torch.Tensor) else b.to(device=a.device, dtype=a.dtype)
        if a.ndim <= axis:
            return a                                   # a launch-uniform leaf (the same in both traces)
        return torch.cat([a, b.reshape(tuple(a.shape[:axis]) + (1,) + tuple(a.shape[axis + 1:]))], dim=axis)

    def leaf(a, b):
        if isinstance(a, (torch.Tensor, Gathered)):
            return cat(a, b)
        if isinstance(a, (tuple, list)):
            return type(a)(leaf(x, y) for x, y in zip(a, b))
        if isinstance(a, dict):
            return {k: leaf(a[k], b[k]) for k in a}
        return a
    if isinstance(trs, DistributionTrace):
        return DistributionTrace(trs.gen_fn, leaf(trs.args, tr_one.args), leaf(trs.value, tr_one.value),
                                 leaf(trs.score, tr_one.score))
    if isinstance(trs, VmapTrace):
        return VmapTrace(trs.gen_fn, stack_traces(trs.inner, tr_one.inner, axis), leaf(engine.materialize(trs.score), tr_one.score),
                         leaf(_materialize_tree(trs.retval), tr_one.retval), leaf(trs.args, tr_one.args))
    return StaticTrace(trs.gen_fn, leaf(trs.args, tr_one.args), leaf(trs.retval, tr_one.retval),
                       OrderedDict((a, stack_traces(s, tr_one.subtraces[a], axis)) for a, s in trs.subtraces.items()))


def trace_leaves(tr) -> list:
    out = []
    trace_map(tr, lambda v: (out.append(v), v)[1])
    return out


# ---------------------------------------------------------------------------
# Part 1: the reference's combinators
# ---------------------------------------------------------------------------
class ParticleCollection:
    """A weighted collection of particles (smc.py:76-109): `particles` is ONE
    batched trace (struct-of-arrays), `log_weights` has the same batch shape.
    `log_ml_offset` (build addition) carries the evidence accumulated by
    earlier resampling steps; it is 0 for the reference's algorithms."""

    def __init__(self, particles, log_weights, is_valid=True, log_ml_offset=None, n_zero=None):
        """log_weights: a tensor, or None with n_zero = N for "all zero" (a freshly resampled collection): the zeros
        are only materialised if somebody asks for them, and `extend` then adds nothing."""
        self.particles, self._lw, self.is_valid = particles, log_weights, is_valid
        self._n_zero = n_zero
        self.log_ml_offset = log_ml_offset

    @property
    def log_weights(self):
        if self._lw is None:
            self._lw = torch.zeros((self._n_zero,), dtype=torch.float32, device=_lib.get().device)
        return self._lw

    @log_weights.setter
    def log_weights(self, v):
        self._lw = v

    def weights_are_zero(self) -> bool:
        return self._lw is None

    def get_particles(self):
        return self.particles

    def get_log_weights(self):
        return self.log_weights

    def get_particle(self, idx):
        return trace_map(self.particles, lambda v: engine.materialize(v)[idx])

    def __getitem__(self, idx):
        return self.get_particle(idx), self.log_weights[idx]

    def get_log_marginal_likelihood_estimate(self):
        """logsumexp(lw) - log N (smc.py:96-97) [+ accumulated offset]."""
        n = self.log_weights.shape[-1]
        est = engine.elementwise(_sub_const, engine.logsumexp_rows(self.log_weights), math.log(n))
        if self.log_ml_offset is not None:
            est = engine.elementwise(_add, est, self.log_ml_offset.value())
        return est

    def sample_index(self, key: Key):
        """Categorical draw proportional to the weights (smc.py:102-108): Gumbel-max
        over logits = lw - logsumexp(lw), gumbel counter = particle index."""
        lw = self.log_weights
        lse = engine.logsumexp_rows(lw)
        logits = engine.elementwise(_sub, lw, lse) if lw.ndim > 1 else engine.elementwise(_sub, lw, lse.reshape(()))
        return engine.categorical_rows(key, logits)

    def sample_particle(self, key: Key):
        idx = self.sample_index(key)
        nb = self.log_weights.ndim

        def take(v):
            v = engine.materialize(v)
            if v.ndim < nb:
                return v
            ix = idx.long().reshape(idx.shape + (1,) * (v.ndim - nb + 1))
            ix = ix.expand(idx.shape + (1,) + tuple(v.shape[nb:]))
            return torch.gather(v, nb - 1, ix).squeeze(nb - 1)
        return trace_map(self.particles, take)


class SMCAlgorithm(Algorithm):
    """smc.py:117-225"""

    def get_num_particles(self): raise NotImplementedError
    def get_final_target(self): raise NotImplementedError
    def run_smc(self, key): raise NotImplementedError

    def run_csmc(self, key, retained): raise NotImplementedError

    def log_marginal_likelihood_estimate(self, key, target=None):
        algorithm = ChangeTarget(self, target) if target else self       # smc.py:150-153
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    def random_weighted(self, key, *args):
        """smc.py:162-179"""
        target = args[0]
        assert isinstance(target, Target)
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        collection = algorithm.run_smc(key)
        particle = collection.sample_particle(sub_key)
        estimate = engine.elementwise(_sub, particle.get_score(), collection.get_log_marginal_likelihood_estimate())
        chm = target.filter_to_unconstrained(particle.get_choices())
        return estimate, chm

    def estimate_logpdf(self, key, v, *args):
        """smc.py:181-198: conditional SMC with `v` retained in the last slot."""
        target = args[0]
        assert isinstance(target, Target)
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        collection = algorithm.run_csmc(key, v)
        particle = collection.sample_particle(sub_key)
        return engine.elementwise(_sub, particle.get_score(), collection.get_log_marginal_likelihood_estimate())

    def estimate_normalizing_constant(self, key, target):
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    def estimate_reciprocal_normalizing_constant(self, key, target, latent_choices, w):
        """smc.py:214-225: `w` (with `latent_choices`) is already properly weighted for `target`, so the
        conditional run skips the redundant re-weighting of the retained particle."""
        return ChangeTarget(self, target).run_csmc_for_normalizing_constant(key, latent_choices, w)


class Importance(SMCAlgorithm):
    """One-particle importance sampling (smc.py:233-266): the particle is
    generated with `key` itself (child 0 of the split)."""

    def __init__(self, target: Target, q=None):
        self.target, self.q = target, q

    def get_num_particles(self): return 1
    def get_final_target(self): return self.target

    def run_smc(self, key):
        key, sub_key = split(key)
        k1 = key.reshape(tuple(key.shape) + (1,))
        if self.q is not None:                                           # smc.py:256-258
            log_weight, choice = self.q.random_weighted(sub_key, self.target)
            tr, score = self.target.importance(k1, _expand(choice))
            return ParticleCollection(tr, engine.elementwise(_sub, score, _expand_leaf(log_weight)), True)
        tr, score = self.target.importance(k1, ChoiceMap.empty())
        return ParticleCollection(tr, score, True)

    def run_csmc(self, key, retained):
        """smc.py:268-279"""
        nb = len(key.shape)                  # a batch of keys (the reference's vmap of run_csmc): particle axis nb
        key, sub_key = split(key)
        q_score = self.q.estimate_logpdf(sub_key, retained, self.target) if self.q is not None else 0.0
        k1 = key.reshape(tuple(key.shape) + (1,))
        tgt = _target_over(self.target, key.shape, 1)
        tr, score = tgt.importance(k1, _expand(retained, nb))
        return ParticleCollection(tr, engine.elementwise(_sub, score, _expand_leaf(q_score, nb)), True)


class ImportanceK(SMCAlgorithm):
    """K-particle importance sampling (smc.py:282-315):
    key, sub = split(key); keys = split(sub, K); one fused launch over K."""

    def __init__(self, target: Target, q=None, k_particles: int = 2):
        self.target, self.q, self.k_particles = target, q, int(k_particles)

    def get_num_particles(self): return self.k_particles
    def get_final_target(self): return self.target

    def run_smc(self, key):
        key, sub_key = split(key)
        sub_keys = split(sub_key, self.k_particles)
        if self.q is not None:                                           # smc.py:301-305
            log_weights, choices = self.q.random_weighted(sub_keys, self.target)
            trs, target_scores = self.target.importance(sub_keys, choices)
            return ParticleCollection(trs, engine.elementwise(_sub, target_scores, log_weights), True)
        trs, target_scores = self.target.importance(sub_keys, ChoiceMap.empty())
        return ParticleCollection(trs, target_scores, True)         # log_weights = scores - 0.0

    def run_csmc(self, key, retained):
        """smc.py:317-351: K-1 fresh particles, the retained choices in slot K-1."""
        K = self.k_particles
        nb = len(key.shape)                  # under a batch of keys everything gains the leading key axes: ONE launch
        key, sub_key = split(key)            # set over [*keys, K] with the retained particle in slot K-1 of every row
        sub_keys = split(sub_key, K - 1)
        tgt_k1 = _target_over(self.target, key.shape, K - 1)
        if self.q is not None:
            log_scores, choices = self.q.random_weighted(sub_keys, tgt_k1)
            retained_score = self.q.estimate_logpdf(key, retained, self.target)
            stacked = _stack_chm(choices, retained, nb)
            stacked_scores = torch.cat([log_scores, _expand_leaf(retained_score, nb).to(log_scores.device)], dim=nb)
            trs, target_scores = _target_over(self.target, key.shape, K).importance(split(key, K), stacked)
            return ParticleCollection(trs, engine.elementwise(_sub, target_scores, stacked_scores), True)
        ignored, ignored_scores = tgt_k1.importance(sub_keys, ChoiceMap.empty())
        retained_tr, retained_score = self.target.importance(key, retained)
        scores = torch.cat([ignored_scores, retained_score.reshape(tuple(ignored_scores.shape[:nb]) + (1,))], dim=nb)
        return ParticleCollection(stack_traces(ignored, retained_tr, nb), scores, True)


class ChangeTarget(SMCAlgorithm):
    """Re-weight every particle for a new target (smc.py:359-396); the incoming
    key is used both for prev.run_smc(key) and for split(key, K) (:374, :386)."""

    def __init__(self, prev: SMCAlgorithm, target: Target):
        self.prev, self.target = prev, target

    def get_num_particles(self): return self.prev.get_num_particles()
    def get_final_target(self): return self.target

    def run_smc(self, key):
        collection = self.prev.run_smc(key)
        particles = collection.get_particles()
        latents = self.prev.get_final_target().filter_to_unconstrained(particles.get_choices())
        sub_keys = split(key, self.get_num_particles())
        tgt = _target_over(self.target, key.shape, self.get_num_particles())
        new_particles, new_weight = tgt.importance(sub_keys, latents)
        this_weight = engine.elementwise(_reweight, new_weight, particles.get_score(), collection.get_log_weights())   # smc.py:383
        return ParticleCollection(new_particles, this_weight, True, collection.log_ml_offset)

    def run_csmc(self, key, retained):
        """smc.py:398-425: prev.run_csmc(key, retained), then the same re-weighting."""
        collection = self.prev.run_csmc(key, retained)
        particles = collection.get_particles()
        latents = self.prev.get_final_target().filter_to_unconstrained(particles.get_choices())
        sub_keys = split(key, self.get_num_particles())
        tgt = _target_over(self.target, key.shape, self.get_num_particles())
        new_particles, new_score = tgt.importance(sub_keys, latents)
        this_weight = engine.elementwise(_reweight, new_score, particles.get_score(), collection.get_log_weights())
        return ParticleCollection(new_particles, this_weight, True)

    def run_csmc_for_normalizing_constant(self, key, latent_choices, w):
        return _csmc_normalizing_constant(self, key, latent_choices, w)


def _csmc_normalizing_constant(self, key, latent_choices, w):
    """ChangeTarget.run_csmc_for_normalizing_constant (smc.py:432-465): conditional SMC under the previous
    target with `latent_choices` retained in slot K-1; the K-1 rejected particles are re-weighted for the new
    target (keys split(key, K-1)), the retained one takes `w - retained_score + retained_weight`;
    returns retained_score - (logsumexp(all weights) - log K)."""
    nb = len(key.shape)                  # a batch of keys: every row of [*keys, K] is one conditional run
    key, sub_key = split(key)
    collection = self.prev.run_csmc(sub_key, latent_choices)
    K = self.get_num_particles()
    particles, lw = collection.get_particles(), collection.get_log_weights()
    scores = particles.get_score()
    last_i = (slice(None),) * nb + (-1,)
    head_i = (slice(None),) * nb + (slice(None, -1),)
    retained_score, retained_weight = scores[last_i], lw[last_i]
    be = _lib.get()
    w_t = w if isinstance(w, torch.Tensor) else torch.tensor(float(w), dtype=torch.float32, device=be.device)
    one = tuple(retained_score.shape) + (1,)
    last = engine.elementwise(_reweight, w_t.to(be.device).expand(retained_score.shape).reshape(one).contiguous(),
                              retained_score.reshape(one).contiguous(), retained_weight.reshape(one).contiguous())
    if K > 1:
        head = trace_map(particles, lambda v: engine.materialize(v)[head_i].contiguous()
                         if tuple(v.shape[nb:nb + 1]) == (K,) else v)
        latents = self.prev.get_final_target().filter_to_unconstrained(head.get_choices())
        _, new_score = _target_over(self.target, key.shape, K - 1).importance(split(key, K - 1), latents)
        rejected = engine.elementwise(_reweight, new_score, scores[head_i].contiguous(), lw[head_i].contiguous())
        all_weights = torch.cat([rejected.reshape(one[:-1] + (K - 1,)), last.reshape(one)], dim=nb)
    else:
        all_weights = last.reshape(one)
    total = engine.logsumexp_rows(all_weights)
    out = engine.elementwise(_recip_z, retained_score.reshape(one).contiguous(), total.reshape(one).contiguous(), math.log(K))
    return out.reshape(tuple(retained_score.shape))


def _recip_z(retained_score, total, log_k): return retained_score - (total - log_k)


def _unbatched(key: Key, who: str):
    if tuple(key.shape) != ():
        raise NotImplementedError(f"{who} under a batch of keys")


def _expand_leaf(v, nb: int = 0):
    """jnp.expand_dims(v, nb) for a score (tensor or Python number): the one-particle axis after `nb` key axes."""
    be = _lib.get()
    t = v if isinstance(v, torch.Tensor) else torch.tensor(float(v), dtype=torch.float32, device=be.device)
    return t.reshape(tuple(t.shape[:nb]) + (1,) + tuple(t.shape[nb:]))


def _expand(chm: ChoiceMap, nb: int = 0) -> ChoiceMap:
    """A one-particle batch of a choice map (its values carry `nb` leading key axes)."""
    be = _lib.get()

    def one(v):
        t = torch.as_tensor(v, device=be.device) if not isinstance(v, torch.Tensor) else v
        if t.dtype == torch.float64:
            t = t.float()
        return t.reshape(tuple(t.shape[:nb]) + (1,) + tuple(t.shape[nb:]))
    return chm.map_values(one)


def _stack_chm(choices: ChoiceMap, one: ChoiceMap, nb: int = 0) -> ChoiceMap:
    """tree_map(stack_to_first_dim, choices, retained) over choice maps with the same addresses (particle axis nb)."""
    out = ChoiceMap.empty()
    for a in choices.addresses():
        v = engine.materialize(choices[a])
        r = torch.as_tensor(one[a], dtype=v.dtype, device=v.device).reshape(tuple(v.shape[:nb]) + (1,) + tuple(v.shape[nb + 1:]))
        out = out.set(a, torch.cat([v, r], dim=nb))
    return out


def _target_over(target: Target, batch: tuple, k: int) -> Target:
    """Under a batch of keys a Target's own observations may be batched over the keys too ([*keys, ...]: a nested
    Marginal's latent choices, one set per key): for a launch over [*keys, k] particles they are repeated along the
    particle axis.  One key, or launch-uniform observations (leading shape != the key batch): left alone."""
    batch = tuple(batch)
    nb = len(batch)
    if nb == 0:
        return target

    def rep(v):
        if isinstance(v, torch.Tensor) and tuple(v.shape[:nb]) == batch:
            return v.unsqueeze(nb).expand(batch + (k,) + tuple(v.shape[nb:])).contiguous()
        return v
    return Target(target.p, tuple(rep(a) for a in target.args), target.constraint.map_values(rep))


# ---------------------------------------------------------------------------
# Part 2: build-defined scalable SMC moves
# ---------------------------------------------------------------------------
MULTINOMIAL_GUIDED_MIN = 8192      # below this the per-slot search of gmx_ancestors is as fast as building a guide table
SYSTEMATIC, STRATIFIED, MULTINOMIAL = (_lib.RESAMPLE_SYSTEMATIC, _lib.RESAMPLE_STRATIFIED,
                                       _lib.RESAMPLE_MULTINOMIAL)
# two-stage multinomial (gmx_multinomial_tiled): the same offspring law, the output ordered by the ancestor's CDF tile
MULTINOMIAL_TILED = _lib.RESAMPLE_MULTINOMIAL_TILED
# multinomial with SORTED uniforms (gmx_resample_sorted): the same offspring law, the output ordered by ancestor — an
# ordered scheme on the systematic resampler's kernel; its order-statistics table comes from the background stream
MULTINOMIAL_SORTED = _lib.RESAMPLE_MULTINOMIAL_SORTED
_KINDS = {"systematic": SYSTEMATIC, "stratified": STRATIFIED, "multinomial": MULTINOMIAL,
          "multinomial_tiled": MULTINOMIAL_TILED, "multinomial_sorted": MULTINOMIAL_SORTED}
_TILE_KINDS = (SYSTEMATIC, STRATIFIED, MULTINOMIAL_TILED, MULTINOMIAL_SORTED)      # from log-weights + tile statistics, n <= 2^21


def cdf_reference(m) -> float:
    """The log-weight an integer CDF total is relative to: total * 2^-shift = sum_i exp(lw_i - cdf_reference(max lw)).
    = ceil(max lw / ln 2) * ln 2 in float32 arithmetic (gmx_tile_exp / gmx_tile_ref in csrc/gmx_math.h: the
    exponent K of the block-floating-point CDF), evaluated on the host for the evidence terms."""
    lim = 1 << 29
    inv_ln2 = np.frombuffer(np.uint32(0x3FB8AA3B).tobytes(), np.float32)[0]
    ln2 = np.frombuffer(np.uint32(0x3F317218).tobytes(), np.float32)[0]
    with np.errstate(invalid="ignore", over="ignore"):
        t = np.float32(m) * inv_ln2
    if not (t > -np.float32(lim)):
        k = -lim
    elif t > np.float32(lim):
        k = lim
    else:
        k = int(t)
        if np.float32(k) < t:
            k += 1
    return float(np.float32(k) * ln2)


def cdf_shift(n_total: int) -> int:
    """Fixed-point exponent: a sum of n_total terms <= 2^shift stays below 2^62."""
    need = 0
    while (1 << need) < n_total:
        need += 1
    return 62 - need


class LogMLOffset:
    """Evidence accumulated by resampling steps, kept on the device as exact
    integers and finished in float64 on the host when asked:
      sum_t [ M_t + log(total_t * 2^-shift) - log N ]."""

    def __init__(self, terms=()):
        self.terms = list(terms)      # (max_d f32[1], total_d u64-as-i64[1], shift, n)

    def plus(self, max_d, total_d, shift, n):
        return LogMLOffset(self.terms + [(max_d, total_d, shift, n)])

    def value(self) -> float:
        acc = 0.0
        for max_d, total_d, shift, n in self.terms:
            m = float(max_d.reshape(-1)[0].item())
            tot = int(total_d.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
            if tot == 0:                 # no particle carried any mass: the evidence estimate is 0
                return -math.inf
            acc += cdf_reference(m) + math.log(tot) - shift * math.log(2.0) - math.log(n)
        return acc


def weight_cdf(lw: torch.Tensor, n_total=None, max_partials=None):
    """gmx_weight_cdf: returns (cdf int64-bits [n], total [1], max [1], shift)."""
    be = _lib.get()
    lw = lw.reshape(-1)
    if lw.dtype != torch.float32:
        lw = lw.float()
    lw = lw.contiguous()
    n = lw.numel()
    shift = cdf_shift(n if n_total is None else n_total)
    cdf = torch.empty((n,), dtype=torch.int64, device=lw.device)
    total = torch.empty((1,), dtype=torch.int64, device=lw.device)
    mx = torch.empty((1,), dtype=torch.float32, device=lw.device)
    ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=lw.device)
    if max_partials is None:
        # max via the deterministic LSE kernel's max output
        rows_ws = torch.empty(((be.c.gmx_logsumexp_workspace(1, n) + 3) // 4,), dtype=torch.int32, device=lw.device)
        dummy = torch.empty((1,), dtype=torch.float32, device=lw.device)
        be.check(be.c.gmx_logsumexp(be.ptr(lw), 1, n, be.ptr(dummy), be.ptr(mx), be.ptr(rows_ws), be.stream()),
                 "gmx_logsumexp")
        be.check(be.c.gmx_weight_cdf(be.ptr(lw), n, shift, None, 0, be.ptr(mx), be.ptr(cdf), be.ptr(total),
                                     be.ptr(ws), be.stream()), "gmx_weight_cdf")
    else:
        be.check(be.c.gmx_weight_cdf(be.ptr(lw), n, shift, be.ptr(max_partials), max_partials.shape[-1],
                                     be.ptr(mx), be.ptr(cdf), be.ptr(total), be.ptr(ws), be.stream()),
                 "gmx_weight_cdf")
    return cdf, total, mx, shift


FUSE_RESAMPLE_MAX = 1 << 20      # the resample-first launch: all its workgroups resident at once (1024 tiles)
FUSE_RESAMPLE_LOOP_MAX = 1 << 24 # ... or, LOOPED, 1024 workgroups walking up to 16 tiles each (engine.Compiled.set_fuse_resample(loop=True))
FUSED_RESAMPLE_MAX = 2048 * 1024        # RS_MAX_TILES tiles of 1024 particles (csrc/gmx_kernels.hip)


def resample_fused(kind, key: Key, lw: torch.Tensor):
    """gmx_resample: log-weights -> (ancestors int32 [n], total [1], max [1], shift) in two launches with no CDF
    array (tile statistics + k_offspring_tile); the same integers as weight_cdf + ancestors_from_cdf."""
    be = _lib.get()
    stats = getattr(lw, "_gmx_tile_stats", None)       # left by the program that computed these weights (run_gfi)
    if stats is not None and (len(stats) < 5 or stats[4] != lw._version):
        stats = None                       # the weights were changed in place since: the statistics are stale
    lw = lw.reshape(-1).float().contiguous()
    if lw.data_ptr() % 16:
        lw = lw.clone()                    # a view into the middle of a buffer: the kernels load float4
        stats = None
    n = lw.numel()
    shift = cdf_shift(n)
    anc = torch.empty((n,), dtype=torch.int32, device=lw.device)
    total = torch.empty((1,), dtype=torch.int64, device=lw.device)
    mx = torch.empty((1,), dtype=torch.float32, device=lw.device)
    if int(kind) in (MULTINOMIAL_TILED, MULTINOMIAL_SORTED):     # tile statistics, then the kind's own entry point
        kh = key.host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        if not (stats is not None and stats[2] == shift and stats[3] == n):
            tiles = (n + 1023) // 1024
            stats = (torch.empty((tiles,), dtype=torch.float32, device=lw.device),
                     torch.empty((tiles,), dtype=torch.int64, device=lw.device))
            be.check(be.c.gmx_tile_stats(be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(stats[1]), be.stream()), "gmx_tile_stats")
        if int(kind) == MULTINOMIAL_SORTED and n > FUSED_RESAMPLE_MAX:
            # past 2048 tiles (config 4's k = 1e7): the tile prefixes from one workgroup, the table kernels in chunks
            table = torch.empty((int(be.c.gmx_sorted_uniforms_words(n)),), dtype=torch.int32, device=lw.device)
            pref = torch.empty((int(be.c.gmx_tile_prefix_words(n)),), dtype=torch.int64, device=lw.device)
            be.check(be.c.gmx_tile_prefix(be.ptr(stats[0]), be.ptr(stats[1]), n, be.ptr(pref), be.stream()), "gmx_tile_prefix")
            be.check(be.c.gmx_resample_sorted_p(kk, be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(pref), be.ptr(table), 0,
                                                be.ptr(mx), be.ptr(total), be.ptr(anc), be.stream()), "gmx_resample_sorted_p")
            return anc, total, mx, shift
        if int(kind) == MULTINOMIAL_SORTED:
            table = torch.empty((int(be.c.gmx_sorted_uniforms_words(n)),), dtype=torch.int32, device=lw.device)
            be.check(be.c.gmx_resample_sorted(kk, be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(stats[1]), be.ptr(table), 0,
                                              be.ptr(mx), be.ptr(total), be.ptr(anc), be.stream()), "gmx_resample_sorted")
            return anc, total, mx, shift
        # count buffers zeroed HERE (a fill kernel; phase 0: "buffer 0 is zero") rather than by the call's own memset
        # (phase -1): inside `smc.capture` the memset node of a one-off workspace did not hold across replays (replay 0
        # right, every later one wrong: tools/experiments/capture_kinds_dbg.py) — a sweep's persistent workspace is fine
        ws = torch.zeros(((be.c.gmx_multinomial_tiled_workspace(n) + 3) // 4,), dtype=torch.int32, device=lw.device)
        be.check(be.c.gmx_multinomial_tiled(kk, be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(stats[1]), None, be.ptr(mx),
                                            be.ptr(total), be.ptr(anc), be.ptr(ws), 0, be.stream()), "gmx_multinomial_tiled")
        return anc, total, mx, shift
    if n > FUSED_RESAMPLE_MAX:
        # more than 2048 tiles (BASELINE config 4: k = 1e7): the same kernel reading tile PREFIXES that one workgroup
        # computes (gmx_tile_prefix) instead of every workgroup reducing the whole statistics table — still no CDF array
        kh = key.host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        if not (stats is not None and stats[2] == shift and stats[3] == n):
            tiles = (n + 1023) // 1024
            stats = (torch.empty((tiles,), dtype=torch.float32, device=lw.device),
                     torch.empty((tiles,), dtype=torch.int64, device=lw.device))
            be.check(be.c.gmx_tile_stats(be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(stats[1]), be.stream()), "gmx_tile_stats")
        pref = torch.empty((int(be.c.gmx_tile_prefix_words(n)),), dtype=torch.int64, device=lw.device)
        be.check(be.c.gmx_tile_prefix(be.ptr(stats[0]), be.ptr(stats[1]), n, be.ptr(pref), be.stream()), "gmx_tile_prefix")
        be.check(be.c.gmx_resample_tiles_p(int(kind), kk, be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(pref), be.ptr(mx),
                                           be.ptr(total), be.ptr(anc), be.stream()), "gmx_resample_tiles_p")
        return anc, total, mx, shift
    if stats is not None and stats[2] == shift and stats[3] == n:
        kh = key.host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        be.check(be.c.gmx_resample_tiles(int(kind), kk, be.ptr(lw), n, shift, be.ptr(stats[0]), be.ptr(stats[1]),
                                         be.ptr(mx), be.ptr(total), be.ptr(anc), be.stream()), "gmx_resample_tiles")
        return anc, total, mx, shift
    ws = torch.empty(((be.c.gmx_resample_workspace(n) + 7) // 8,), dtype=torch.int64, device=lw.device)
    kh = key.host()
    kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
    be.check(be.c.gmx_resample(int(kind), kk, be.ptr(lw), n, shift, None, 0, be.ptr(mx), be.ptr(total), be.ptr(anc),
                               be.ptr(ws), be.stream()), "gmx_resample")
    return anc, total, mx, shift


def ancestors_from_cdf(kind, key: Key, cdf, total, n_out=None) -> torch.Tensor:
    be = _lib.get()
    n_in = cdf.numel()
    n_out = n_in if n_out is None else int(n_out)
    kh = key.host()
    kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
    anc = torch.empty((n_out,), dtype=torch.int32, device=cdf.device)
    if int(kind) == MULTINOMIAL and n_in >= MULTINOMIAL_GUIDED_MIN:
        # unordered slots: through the guide table (two table reads + a search over ~3 entries per slot instead of a
        # binary search over n_in) — the same ancestors
        ws = torch.empty(((be.c.gmx_multinomial_workspace(n_in) + 3) // 4,), dtype=torch.int32, device=cdf.device)
        be.check(be.c.gmx_multinomial(kk, be.ptr(cdf), n_in, be.ptr(total), n_out, be.ptr(anc), be.ptr(ws), be.stream()),
                 "gmx_multinomial")
        return anc
    be.check(be.c.gmx_ancestors(int(kind), kk, be.ptr(cdf), n_in, 0, be.ptr(total), n_out, 0, n_out,
                                be.ptr(anc), be.stream()), "gmx_ancestors")
    return anc


def resample(key: Key, collection: ParticleCollection, kind="systematic", n_out=None) -> ParticleCollection:
    """Resample a 1-D particle collection.  The result's leaves are lazy
    gathers (`engine.Gathered`), so a following `extend` fuses the gather into
    its kernel; weights reset to 0 and the evidence moves into log_ml_offset."""
    lw = collection.get_log_weights()
    if lw.ndim != 1:
        raise NotImplementedError("resample: batched collections")
    kind = _KINDS[kind] if isinstance(kind, str) else int(kind)
    n = lw.numel()
    if (kind in _TILE_KINDS and n_out in (None, n) and 0 < n <= FUSED_RESAMPLE_MAX) or \
            (kind in (SYSTEMATIC, STRATIFIED, MULTINOMIAL_SORTED) and n_out in (None, n) and FUSED_RESAMPLE_MAX < n < 2 ** 31 - 8192):
        anc, total, mx, shift = resample_fused(kind, key, lw)        # no CDF in memory (gmx_resample[_tiles[_p]])
    elif kind in (MULTINOMIAL_TILED, MULTINOMIAL_SORTED):
        raise NotImplementedError("resample(kind='multinomial_tiled'): n_out = n <= 2^21; 'multinomial_sorted': n_out = n "
                                  "(use 'multinomial')")
    else:
        cdf, total, mx, shift = weight_cdf(lw)
        anc = ancestors_from_cdf(kind, key, cdf, total, n_out)
    particles = trace_map(collection.get_particles(),
                          lambda v: Gathered(engine.materialize(v), anc) if tuple(v.shape[:1]) == (n,) else v)
    off = (collection.log_ml_offset or LogMLOffset()).plus(mx, total, shift, n)
    out = ParticleCollection(particles, None, True, off, n_zero=anc.numel())     # weights reset to 0, lazily
    out.ancestors = anc
    return out


def extend(key: Key, collection: ParticleCollection, step, step_args, observations: ChoiceMap) -> ParticleCollection:
    """Bootstrap extension by one step: every particle i runs
    `step.importance(split(key, N)[i], observations, step_args_i)` and
    lw_i += weight_i (same algebra as ChangeTarget._reweight, smc.py:378-384,
    restricted to the new step's sites).  `step_args` is a tuple, or a callable
    mapping the previous particles' trace to the tuple."""
    zero = collection.weights_are_zero()
    n = collection._n_zero if zero else collection.get_log_weights().shape[0]
    args = step_args(collection.get_particles()) if callable(step_args) else tuple(step_args)
    keys = split(key, n)
    from ..static import StaticGenerativeFunction, run_gfi
    if isinstance(step, StaticGenerativeFunction):
        # the step's program also leaves the resampler's tile statistics of its weight when it can
        tr, w = run_gfi(step, "generate", keys, args, constraint=observations, weight_stats=zero)
    else:
        tr, w = step.importance(keys, observations, args)
    if zero:                           # lw + w with lw = 0 exactly (0.0f + w == w for every w but -0.0, which a
        return ParticleCollection(tr, w, True, collection.log_ml_offset)          # sum of log-densities is not)
    return ParticleCollection(tr, engine.elementwise(_add, collection.get_log_weights(), w), True, collection.log_ml_offset)


def rejuvenate(key: Key, collection: ParticleCollection, request, argdiffs=None) -> ParticleCollection:
    """One MH sweep over all particles as ONE fused launch (static.run_mh):
    particle i uses key_i = split(key, N)[i], (k_edit, k_acc) = split(key_i);
    propose with `request.edit(k_edit, ...)`, accept iff log U(k_acc) < weight,
    keep the old trace otherwise (tests/inference/test_requests.py:131-137 idiom).
    Weights are unchanged (an MH kernel leaves the target invariant)."""
    from ..static import run_mh
    tr = collection.get_particles()
    n = collection._n_zero if collection.weights_are_zero() else collection.get_log_weights().shape[0]
    if argdiffs is None:
        argdiffs = Diff.no_change(tr.get_args() or ())
    new_tr, accept, _ = run_mh(tr.get_gen_fn(), lazy_split(key, n), tr, request, argdiffs)
    res = ParticleCollection(new_tr, collection._lw, True, collection.log_ml_offset, n_zero=collection._n_zero)
    res.accept = accept
    return res


class CapturedLoop:
    NOISE_ARENA_MB = 16384     # every draw of a captured loop lives in one arena for the graph's lifetime: its bound

    """A Python inference loop over this module's functional API (resample -> rejuvenate -> extend ...) captured ONCE
    into a hipGraph and replayed without the interpreter in the loop: every C-ABI launch goes to torch's current stream,
    so one capture records them all; tensors made during the capture come from the graph's private pool and stay
    valid (and are overwritten in place) across replays.  `result` is whatever the loop returned at capture time —
    its tensors hold the latest replay's values."""

    def __init__(self, loop_fn, *args, warmup: int = 1, noise_ahead=None):
        be = _lib.get()
        if not be.uses_streams:
            raise _lib.GenmiError("capture needs the HIP backend")
        if noise_ahead is None:
            noise_ahead = os.environ.get("GENMI_NOISE_AHEAD", "1") != "0"
        from contextlib import nullcontext
        from ..static import NoiseAheadContext
        # noise ahead: the draws of the loop's extend / rejuvenate launches by background programs on a
        # second stream; in the captured graph they depend on nothing but each other and run ahead of the chain
        self.noise = NoiseAheadContext(torch.cuda.Stream(device=be.device)) if noise_ahead else None
        scope = self.noise if self.noise is not None else nullcontext()
        side = torch.cuda.Stream(device=be.device)
        side.wait_stream(torch.cuda.current_stream(be.device))
        with torch.cuda.stream(side):           # programs are traced / specialised outside the capture
            for _ in range(max(1, warmup)):
                with scope:
                    loop_fn(*args)
            if self.noise is not None and self.noise.settle():
                with scope:                      # once more, with the launches that keep their draws as they will run
                    loop_fn(*args)
            if self.noise is not None:
                side.wait_stream(self.noise.stream)
                # every draw of the loop lives in one arena for the graph's lifetime: bounded (NOISE_ARENA_MB)
                if 4 * self.noise.demand > int(self.NOISE_ARENA_MB) << 20:
                    self.noise = None
                    loop_fn(*args)               # the plain programs must exist before the capture too
        torch.cuda.current_stream(be.device).wait_stream(side)
        torch.cuda.synchronize(be.device)
        if self.noise is not None:
            self.noise.reserve(be.device)
        self.graph = torch.cuda.CUDAGraph()
        # the graph's kernel nodes point into the site programs' code: hold every program launched during the capture
        # for as long as the graph lives (the program caches are bounded LRUs; genjax_amd.clear_caches() is public)
        with engine.holding_captured_programs() as self.programs, torch.cuda.graph(self.graph):
            if self.noise is not None:
                cur = torch.cuda.current_stream(be.device)
                self.noise.stream.wait_stream(cur)          # the background stream joins the capture ...
                self.noise.issue_ahead(be.device)           # ... every recorded noise launch goes out first ...
                with self.noise:
                    self.result = loop_fn(*args)
                cur.wait_stream(self.noise.stream)          # ... and the stream is joined back before the capture ends
            else:
                self.result = loop_fn(*args)

    def replay(self):
        self.graph.replay()
        return self.result


def capture(loop_fn, *args, warmup: int = 1, noise_ahead=None) -> CapturedLoop:
    """`smc.capture(loop_fn, *args)`: see CapturedLoop.  Host-side reads of device values inside `loop_fn`
    (`.item()`, `float(tensor)`, LogMLOffset.value()) are not capturable — return tensors and read them after
    `replay()`.  noise_ahead (default: on; GENMI_NOISE_AHEAD=0): launches over >= 2^18 particles take their
    launch-keyed normal / uniform draws from background programs on a second stream (same values)."""
    return CapturedLoop(loop_fn, *args, warmup=warmup, noise_ahead=noise_ahead)


class _NoiseAhead:
    """The noise-ahead machinery shared by BootstrapSweep and sharded.ShardedBootstrapSweep: background
    programs that draw the chain programs' hoisted normal / uniform values on a second stream, a group of steps ahead.
    The host class provides n, T, step_keys, specialize, fuse_mh, `_chain_prog(t)` (the site program of step t) and
    `_chain_step(t, skip_vm)` (everything step t launches on the chain); `_noise_total` / `_noise_offset`: the keys of
    this shard are children offset .. offset + n - 1 of split(k, total) (a single GPU: total = n, offset = 0)."""

    NOISE_LDS_PAD = 56000      # bytes of unused LDS per noise workgroup: two of them per CU (160 KB)
    NOISE_GROUP = 10           # steps per group of noise launches (the noise runs one group ahead of the chain)
    NOISE_RING = 3             # groups of noise buffers (the background stream runs at most NOISE_RING - 1 groups ahead)
    _noise_offset = 0
    _noise_total = None

    def _noise_split(self, k):
        total = self._noise_total or self.n
        return lazy_split(k, total, offset=self._noise_offset) if (self._noise_offset or total != self.n) else lazy_split(k, self.n)

    def _noise_setup(self, chain_progs, n, T, dev):
        from ..static import NoiseProgram
        be = _lib.get()
        self.noise_ahead = True
        # one background program per (chain program, root key): the draws that hang off the launch key (the
        # step's own sites; with rejuvenate=, the move's proposal and accept draws: key k_mh) and those that hang
        # off the chained extension's key (KSPLITU: k_prop)
        for P in chain_progs:
            if id(P) in self._noise_progs:
                continue
            by_root = {}
            for k_, d in enumerate(P.noise):
                by_root.setdefault(d[0], []).append(k_)
            self._noise_progs[id(P)] = [(root, NoiseProgram([P.noise[k_] for k_ in idx], (n,)), idx)
                                        for root, idx in by_root.items()]
        self.noise_group = max(1, min(int(os.environ.get("GENMI_NOISE_GROUP", self.NOISE_GROUP)), T))
        # the ring of noise buffers: NOISE_RING groups, so the background stream may run NOISE_RING - 1 groups ahead
        # (measured on MI355X, config 2: 14.50 -> 14.32 us/step with 3, 14.27 with 4; the sorted multinomial, whose
        # background stream also builds tables, is better off with 2: 21.2 vs 21.9 — profiles/r03cd_ring_*.json)
        ring = 2 if getattr(self, "kind", None) == MULTINOMIAL_SORTED else self.NOISE_RING
        self.noise_ring = max(2, int(ring))
        # groups of steps [start, end): the noise of group g + 1 is issued before the chain of group g.  The chain
        # can only start once the FIRST group's noise is there, so the groups grow 1, 2, 4, ... up to noise_group
        self.noise_groups, self.noise_slot = [], []
        t0, size = 0, 1
        while t0 < T:
            t1 = min(T, t0 + min(size, self.noise_group))
            for t in range(t0, t1):
                self.noise_slot.append((len(self.noise_groups) % self.noise_ring, t - t0))
            self.noise_groups.append((t0, t1))
            t0, size = t1, size * 2
        S = max(len(P.noise) for P in chain_progs)
        # two groups of noise buffers: the background stream fills one while the chain reads the other.
        # [half, draw, row of the group, n]: a draw's rows are contiguous, so ONE launch can fill several steps
        self.zbuf = torch.zeros((self.noise_ring, S, self.noise_group, n), dtype=torch.float32, device=dev)
        self._noise_stream = torch.cuda.Stream(device=dev) if be.uses_streams else None
        pad = int(self.NOISE_LDS_PAD)
        for plist in self._noise_progs.values():
            for _, q, _ in plist:
                if self.specialize:
                    q.comp.set_background(pad)
                    q.comp.specialize()

    def _noise_views(self, t, count):
        """the [1, n] buffers of step t's draws: half (t // group) % 2 of the ring, row t % group"""
        half, row = self.noise_slot[t]
        return [self.zbuf[half, k, row:row + 1] for k in range(count)]

    def _noise_leaves(self, t, prog):
        return [v.reshape(self.n) for v in self._noise_views(t, len(prog.noise))]

    def _noise_runs(self, g):
        """The background launches of group g: the steps of a group that share a chain program get their draws from ONE
        launch per key root — a 2-D grid, one row of keys per step (GMX_KEY_ROWSPLIT; gmx_program_run) — instead of
        one launch per step: fewer nodes in the graph (the HIP runtime walks a two-stream graph node by node on the
        host) and no launch boundary between the steps' noise."""
        cache = self.__dict__.setdefault("_noise_run_cache", {})
        if g in cache:
            return cache[g]
        n = self.n
        t0, t1 = self.noise_groups[g]
        runs, ta = [], t0
        while ta < t1:
            tb = ta + 1
            while tb < t1 and self._chain_prog(tb) is self._chain_prog(ta):
                tb += 1
            runs.append((ta, tb))
            ta = tb
        out = []
        dev = self.zbuf.device
        for ta, tb in runs:
            P = self._chain_prog(ta)
            half, row_a = self.noise_slot[ta]
            rows = tb - ta
            mh = self.fuse_mh and ta >= 1
            for root, q, idx in self._noise_progs[id(P)]:
                ks = [self.step_keys[t][2] if (mh and root == "LDKEY") else self.step_keys[t][0] for t in range(ta, tb)]
                if rows == 1 or rows * n >= 2 ** 31 - 4096:
                    for r, k in enumerate(ks):
                        out.append((q, (n,), self._noise_split(k), [self.zbuf[half, k_, row_a + r:row_a + r + 1] for k_ in idx]))
                    continue
                kd = torch.from_numpy(np.stack([k.host() for k in ks]).astype(np.uint32).view(np.int32)).to(dev)
                # row r's keys: children offset .. offset + n - 1 of split(ks[r], total) (GMX_KEY_ROWSPLIT + index_offset)
                key = Key(lazy=("rowsplit", Key(dev=kd), n), split_last=True, offset=self._noise_offset)
                out.append((q, (rows * n,), key, [self.zbuf[half, k_, row_a:row_a + rows].reshape(1, rows * n) for k_ in idx]))
        cache[g] = out
        return out

    def _launch_noise_group(self, g):
        self._launch_group_extras(g)
        for q, batch, key, outs in self._noise_runs(g):
            q.run(batch, key, outs)

    def _launch_group_extras(self, g):
        """other key-only work of group g's steps that belongs on the background stream (BootstrapSweep: the stratified
        resampler's slot uniforms)"""

    def _launch_noise(self, t):
        """the draws step t's chain program reads, by the background programs: root LDKEY from the program's launch key
        (k_prop; the chained MH + extension program: k_mh), root KSPLITU from the extension's key k_prop"""
        P = self._chain_prog(t)
        views = self._noise_views(t, len(P.noise))
        mh = self.fuse_mh and t >= 1
        for root, q, idx in self._noise_progs[id(P)]:
            k = self.step_keys[t][2] if (mh and root == "LDKEY") else self.step_keys[t][0]
            q.run((self.n,), self._noise_split(k), [views[k_] for k_ in idx])

    def _enqueue_noise_ahead(self, skip_vm=False, skip_noise=False):
        """The sweep on TWO streams: the chain [site program' -> resampler] per step on the current one, the noise
        programs on the background stream, one group of steps ahead (group g + 1's noise is issued before group g's
        chain; with a ring of R groups of buffers, group g + R - 1's before group g's chain: it may overwrite slot (g - 1) % R
        once the chain of group g - 1 has read it; the groups
        grow 1, 2, 4, ... steps up to noise_group, so the chain starts after ONE noise launch).  Capturable:
        the background stream joins the capture through the first event wait and is joined back at the end.
        Without streams (the CPU mirror of the C-ABI) the same launches run in issue order."""
        be = _lib.get()
        spans = self.noise_groups
        groups = len(spans)
        two = be.uses_streams and self._noise_stream is not None
        if two:
            A, Bs = torch.cuda.current_stream(be.device), self._noise_stream
            Bs.wait_stream(A)
        done, ready = [None] * groups, [None] * groups

        def noise_group(g):
            if skip_noise:
                return
            if two:
                with torch.cuda.stream(Bs):
                    if g >= ring:
                        Bs.wait_event(done[g - ring])
                    self._launch_noise_group(g)
                    ready[g] = torch.cuda.Event()
                    ready[g].record(Bs)
            else:
                self._launch_noise_group(g)

        ring = getattr(self, "noise_ring", 2)
        for g in range(min(ring - 1, groups)):
            noise_group(g)
        for g in range(groups):
            if g + ring - 1 < groups:
                noise_group(g + ring - 1)
            if two and not skip_noise:
                A.wait_event(ready[g])
            for t in range(*spans[g]):
                self._chain_step(t, skip_vm)
            if two and not skip_noise:
                done[g] = torch.cuda.Event()
                done[g].record(A)
        if two:
            A.wait_stream(Bs)


class BootstrapSweep(_NoiseAhead):
    """A whole bootstrap particle filter (T steps, resampling every step) as a
    fixed sequence of launches on one stream, capturable into a hipGraph — two launches per step:

      per step t:  [site program]  x_t[i] ~ step(x_{t-1}[anc[i]]), lw[i] = log p(y_t | x_t[i]); specialised with 4
                                   particles per thread a workgroup is one 1024-particle tile of the integer CDF
                                   and also leaves the tile statistics (max, fixed-point weight sum)
                   [offspring]     k_offspring_tile: global exponent + tile prefixes from the statistics, the
                                   tile's CDF rebuilt in registers, exact systematic / stratified ancestors
      (programs that cannot write the statistics: + gmx_tile_stats; multinomial or n > 2^21: gmx_weight_cdf +
       gmx_ancestors)

    Key schedule (build-defined): step key = fold_in(run_key, t);
    (k_prop, k_res, k_mh) = split(step key, 3); particle i uses split(k_prop, N)[i].
    The evidence is accumulated from the integer CDF totals in float64 on the host.

    NOISE AHEAD (`noise_ahead`, default: on when it applies; GENMI_NOISE_AHEAD=0 switches it off).  A step is bound by
    vector-instruction issue, and two thirds of its instructions are the Threefry blocks and the `erf_inv` of the
    step's normal draws — which depend on keys and particle indices only, not on anything the chain
    [site program -> resampler -> site program ...] produces.  So the step's `normal` sites take their standard-normal
    draws from memory (static.MinimalGenerate(hoist_noise=True)) and a BACKGROUND program (static.NoiseProgram:
    priority 0, a capped number of workgroups per CU) draws them on a second stream, a group of steps ahead of the
    chain, filling the issue slots the chain's launch boundaries and memory round trips leave idle.  Same keys, same
    operations in the same order: every particle, weight and ancestor is the one the one-stream form computes.
    Measured on MI355X (config 2): 17.9 -> 15.8 us/step.

    `resample`: "systematic" (default), "stratified", and three multinomial forms with the same offspring law —
    "multinomial" (iid slot order: a random particle per slot), "multinomial_tiled" (ordered by the ancestor's CDF tile)
    and "multinomial_sorted" (the uniforms drawn sorted: ordered by ancestor, on the systematic resampler's kernel; the
    fastest of the three: DESIGN.md §4).  The background stream also draws what the resampler needs from its key alone:
    the stratified slot uniforms, the sorted multinomial's order-statistics table.
    """

    NOISE_LDS_PAD = 56000      # bytes of unused LDS per noise workgroup: two of them per CU (160 KB)
    NOISE_GROUP = 10           # steps per group of noise launches (the noise runs one group ahead of the chain)
    NOISE_ROOTS_MH = "LDKEY"   # with rejuvenate=: which keys' draws the background programs take (see prepare)
    SORTED_LDS_PAD = 16000     # residency cap of the sorted multinomial's table kernels (they wait on memory)

    def __init__(self, init, step, n_particles: int, T: int, obs_addr="y", resample="systematic",
                 step_extra=None, specialize=True, rejuvenate=None, state_addr="x", noise_ahead=None, chain_mh=True,
                 noise_roots=None, fuse_resample=None):
        """chain_mh=False keeps the MH move and the extension as two launches (the form a chained program too large
        for the tile statistics falls back to); noise_roots: which keys' draws of the chained program the background
        stream takes ("LDKEY" = the move's proposal + accept draws, the default; "KSPLITU" = the extension's; "all").
        rejuvenate: an edit request (e.g. StaticRequest({"x": Rejuvenate(...)})) applied as one fused
        MH move per particle after every resampling, before the next extension (BASELINE config 3; the
        graph-captured form of smc.resample -> smc.rejuvenate -> smc.extend, same keys, same results).
        Supported for models whose trace is {state_addr: the return value, obs_addr: the observation}."""
        self.init, self.step, self.n, self.T = init, step, int(n_particles), int(T)
        self.obs_addr, self.state_addr, self.rejuvenate = obs_addr, state_addr, rejuvenate
        self.kind = _KINDS[resample] if isinstance(resample, str) else int(resample)
        self.step_extra = step_extra or (lambda t: ())
        self.specialize = specialize
        self.graph = None
        self.noise_ahead_req = noise_ahead
        self.chain_mh = bool(chain_mh)
        self.noise_roots = noise_roots or self.NOISE_ROOTS_MH
        # ONE launch per step (None: when it applies): the program that gathers the resampled state first resamples the
        # previous step itself — its workgroup's tile of k_offspring_tile, ancestors as tagged words its neighbours poll
        # (include/genmi.h: gmx_run_args.rs) — so only the grid-wide dependency (the tile statistics) still needs a
        # launch boundary.  Systematic resampling, n <= 2^20, specialised programs.
        self.fuse_req = fuse_resample

    def prepare(self, key: Key, ys: torch.Tensor):
        from ..static import MinimalGenerate as _MG, NoiseProgram
        be = _lib.get()
        # a sweep object may be prepared again (another key, other observations): nothing bound for the previous run
        # survives — the background launches' key rows in particular (found by the unbiasedness test on the GPU: a
        # second prepare() replayed the first run's noise launches)
        self.__dict__.pop("_noise_run_cache", None)
        if self.graph is not None:           # a graph captured for the previous run holds its launch arguments
            if be.uses_streams:
                torch.cuda.synchronize()
            be.c.gmx_graph_destroy(self.graph)
            self.graph = None
        # noise ahead: asked for explicitly, or by default on a device with streams on the fast path (specialised
        # programs; with rejuvenate=, the MH move chained into the extension)
        fuse_mh_ok = self.chain_mh
        want_na = self.noise_ahead_req
        if want_na is None:
            want_na = (os.environ.get("GENMI_NOISE_AHEAD", "1") != "0" and be.uses_streams and self.specialize
                       and (self.rejuvenate is None or fuse_mh_ok))
        if want_na and self.rejuvenate is not None and not fuse_mh_ok:
            raise NotImplementedError("BootstrapSweep(noise_ahead=True, rejuvenate=...) needs the chained MH + extension "
                                      "program (chain_mh=False)")
        self.noise_ahead = False
        self._noise_progs = {}

        def MinimalGenerate(*a):
            return _MG(*a, hoist_noise=bool(want_na))
        n, T = self.n, self.T
        dev = be.device
        self.key = key
        self.ys = ys.to(dev).float().contiguous()
        assert self.ys.numel() >= T
        self.lw = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.cdf = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.anc = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.maxs = torch.zeros((T,), dtype=torch.float32, device=dev)
        self.totals = torch.zeros((T,), dtype=torch.int64, device=dev)
        self.shift = cdf_shift(n)
        self.ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
        self.fused = self.kind in _TILE_KINDS and n <= (512 * 4096)
        self.fuse = False
        if self.kind in (MULTINOMIAL_TILED, MULTINOMIAL_SORTED) and not self.fused:
            raise NotImplementedError("resample='multinomial_tiled' / 'multinomial_sorted': n <= 2^21 per GPU (use 'multinomial')")
        self.sorted_ws = torch.zeros((int(be.c.gmx_sorted_uniforms_words(n)),), dtype=torch.int32, device=dev) \
            if self.kind == MULTINOMIAL_SORTED else None
        self.mnt_ws = torch.zeros(((be.c.gmx_multinomial_tiled_workspace(n) + 3) // 4,), dtype=torch.int32, device=dev) \
            if self.kind == MULTINOMIAL_TILED else None
        self.mn_ws = torch.zeros(((be.c.gmx_multinomial_workspace(n) + 3) // 4,), dtype=torch.int32, device=dev) \
            if self.kind == MULTINOMIAL else None
        self.rs_ws = torch.zeros(((be.c.gmx_resample_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev) \
            if self.fused else None
        obs0 = ChoiceMap.empty().set(self.obs_addr, self.ys[0])
        self.p_init = MinimalGenerate(self.init, (), obs0, (n,))
        # the state is the model's return value: a scalar, or ONE vector of D floats per particle, stored
        # struct-of-arrays as [D, n] and seen by models (and by state()) as the [n, D] view
        ro = self.p_init.ro
        outs_ = self.p_init.comp.outputs
        self.tuple_state = None
        if ro[0] in ("tuple", "list") and ro[1] and all(o[0] == "out" and outs_[o[1]][0] == "f32" and outs_[o[1]][1] == ()
                                                          for o in ro[1]):
            # a TUPLE of float scalars (the latent sites a step hands to the next one): D rows of one [D, n] store,
            # every row its own output of the site program and its own gathered input of the next step
            self.tuple_state = type(()) if ro[0] == "tuple" else type([])
            if self.rejuvenate is not None:
                raise NotImplementedError("BootstrapSweep(rejuvenate=...): a tuple state is not supported (return one "
                                          "array, or use smc.resample / rejuvenate / extend under smc.capture)")
            event, D = (), len(ro[1])
        else:
            if ro[0] != "out":
                raise NotImplementedError("BootstrapSweep: the step model must return one array (scalar or vector "
                                          "state) or a tuple of float scalars")
            dt, event, _slots = outs_[ro[1]]
            if dt != "f32" or len(event) > 1:
                raise NotImplementedError("BootstrapSweep: the state must be a float scalar or a float vector")
            D = int(np.prod(event, dtype=np.int64)) if event else 1
        self.event = tuple(event)
        self.x_store = [torch.zeros((D, n), dtype=torch.float32, device=dev) for _ in range(2)]
        if self.tuple_state is not None:
            self.x = [self.tuple_state(s_[d] for d in range(D)) for s_ in self.x_store]
        else:
            self.x = [s_.reshape(n) if not event else s_.t() for s_ in self.x_store]
        g = self._gathered(0)
        if self.rejuvenate is None:
            self.p_step = MinimalGenerate(self.step, (g,) + tuple(self.step_extra(1)), obs0, (n,))
        else:
            from ..static import MinimalMH
            # xm[t % 2]: the MH-moved, resampled state the extension of step t starts from
            self.xm_store = [torch.zeros((D, n), dtype=torch.float32, device=dev) for _ in range(2)]
            self.xm = [s_.reshape(n) if not event else s_.t() for s_ in self.xm_store]
            self.accept = torch.zeros((n,), dtype=torch.bool, device=dev)
            self.p_step = _MG(self.step, (self.xm[0],) + tuple(self.step_extra(1)), obs0, (n,))
            ch = obs0.set(self.state_addr, g)
            self.p_mh_init = MinimalMH(self.init, (), ch, self.rejuvenate, (n,))
            self.p_mh_step = MinimalMH(self.step, (Gathered(self.xm[0], self.anc),) + tuple(self.step_extra(1)), ch,
                                       self.rejuvenate, (n,))
        # the MH move and the extension that follows it as ONE program / one launch per step (static.MinimalMHGenerate:
        # same keys, same draws, same bits; one launch boundary and one trip of the moved state through memory less).
        self.p_mhvm_init = self.p_mhvm_step = None
        if self.rejuvenate is not None and self.chain_mh:
            from ..static import MinimalMHGenerate
            ex = tuple(self.step_extra(1))
            # which draws of the chained program go to the background stream: the two streams should carry about the
            # same vector work (noise_roots: "all", "LDKEY" = the move's proposal + accept draws, "KSPLITU" = the
            # extension's draw)
            roots = self.noise_roots
            hn = False if not want_na else (True if roots == "all" else tuple(roots.split(",")))
            self.p_mhvm_init = MinimalMHGenerate(self.init, (), ch, self.rejuvenate, self.step, ex, obs0, (n,),
                                                 hoist_noise=hn)
            self.p_mhvm_step = MinimalMHGenerate(self.step, (Gathered(self.xm[0], self.anc),) + ex, ch, self.rejuvenate,
                                                 self.step, ex, obs0, (n,), hoist_noise=hn)
        # the chain's programs by step: t = 0, t = 1, t >= 2
        chain_progs = (self.p_init, self.p_step, self.p_step) if self.rejuvenate is None else \
            (self.p_init, self.p_mhvm_init, self.p_mhvm_step)
        if want_na and chain_progs[2] is not None and chain_progs[2].noise:
            self._noise_setup(chain_progs, n, T, dev)
        elif want_na and any(P is not None and P.noise for P in chain_progs):
            # the steady-state program draws nothing ahead although another one would: the plain programs throughout
            self.noise_ahead_req = False
            return self.prepare(key, ys)
        self.fused = self.kind in _TILE_KINDS and n <= (512 * 4096)
        want_fuse = self.fuse_req
        if want_fuse is None:
            want_fuse = be.uses_streams
        # past 2^21 particles the standalone tile-form resamplers do not apply (RS_MAX_TILES) — the LAST step of a sweep is
        # resampled through the CDF array there — but the looped resample-first launch does (up to 2^24): the steps in
        # between stay ONE launch each
        self.big = bool(self.kind == SYSTEMATIC and not self.fused and n <= FUSE_RESAMPLE_LOOP_MAX)
        want_fuse = bool(want_fuse and self.specialize and self.kind == SYSTEMATIC and (self.fused or self.big)
                         and n <= FUSE_RESAMPLE_LOOP_MAX)
        if self.rejuvenate is None:
            gatherers = (self.p_step,)
        elif self.p_mhvm_init is not None:
            gatherers = (self.p_mhvm_init, self.p_mhvm_step)
        else:
            gatherers = ()
        if want_fuse:
            for p_ in gatherers:
                if not p_.comp.is_specialized():
                    # (past 2^20 particles the workgroups of ONE launch are not all resident: each walks several tiles)
                    # (the looped form at 1e6 particles, by workgroup count, was measured and is slower than one tile per
                    #  workgroup: profiles/r06d_loop_grid.txt — 14.2 us/step plain, 14.9 looped at 1024, 17.4 at 512)
                    p_.comp.set_fuse_resample(loop=n > FUSE_RESAMPLE_MAX)
        if self.specialize:
            self.p_init.comp.specialize()
            self.p_step.comp.specialize()
            if self.rejuvenate is not None:
                self.p_mh_init.comp.specialize()
                self.p_mh_step.comp.specialize()
            if self.p_mhvm_init is not None:
                self.p_mhvm_init.comp.specialize()
                self.p_mhvm_step.comp.specialize()
        # block partials: sized for the interpreter's one row per 256 particles; a specialised kernel
        # writes fewer rows (gmx_program_grid), asked per launch in _rows()
        self.partials = torch.zeros((2, (n + 255) // 256), dtype=torch.float32, device=dev)
        # two launches per step: when the site programs can leave the CDF tile statistics themselves (specialised,
        # 4 particles per thread: a workgroup is one 1024-particle tile) the resampler needs no pass of its own
        # over the log-weights (gmx_resample_tiles); programs that cannot (interpreted) get a gmx_tile_stats launch
        self.tile_agg = torch.zeros(((n + 1023) // 1024,), dtype=torch.int64, device=dev)
        self.tile_stats = bool((self.fused or (self.big and want_fuse)) and self.p_init.comp.writes_tile_stats()
                               and self.p_step.comp.writes_tile_stats())
        if self.p_mhvm_init is not None and self.tile_stats and not (self.p_mhvm_init.comp.writes_tile_stats()
                                                                     and self.p_mhvm_step.comp.writes_tile_stats()):
            self.p_mhvm_init = self.p_mhvm_step = None      # the chained programs are too large for the tile form
        self.fuse_mh = self.p_mhvm_init is not None
        if self.noise_ahead and self.rejuvenate is not None and not self.fuse_mh:
            # the noise-ahead form needs the chained program: start over with the plain ones
            if self.noise_ahead_req:
                raise NotImplementedError("BootstrapSweep(noise_ahead=True, rejuvenate=...): the chained MH + extension "
                                          "program does not fit the tile form")
            self.noise_ahead_req = False
            return self.prepare(key, ys)
        # ONE launch per step where the gathering programs can resample first: two sets of log-weights / statistics (a
        # launch reads step t-1's while it writes step t's)
        # ... and where the device holds every workgroup of such a launch at once (they wait for each other): asked of
        # the runtime per code object (engine.Compiled.resident_particles), two launches per step otherwise
        self.fuse = bool(want_fuse and self.tile_stats and gatherers and (self.rejuvenate is None or self.fuse_mh)
                         and all(p_.comp.fuses_resample() and p_.comp.resident_particles() >= n for p_ in gatherers))
        if self.fuse_req and not self.fuse:
            raise NotImplementedError("BootstrapSweep(fuse_resample=True): needs specialised 4-particles-per-thread programs "
                                      "that gather, systematic resampling and n <= 2^20")
        if self.fuse:
            self.lw_pp = [self.lw, torch.zeros_like(self.lw)]
            self.partials_pp = [self.partials, torch.zeros_like(self.partials)]
            self.tile_agg_pp = [self.tile_agg, torch.zeros_like(self.tile_agg)]
            self.rs_status = torch.zeros((1,), dtype=torch.int64, device=dev)
        else:
            self.lw_pp, self.partials_pp, self.tile_agg_pp = [self.lw] * 2, [self.partials] * 2, [self.tile_agg] * 2
        # per-step keys on the host
        self.step_keys = []
        for t in range(T):
            ks = split(fold_in(key, t), 3)
            self.step_keys.append((ks[0], ks[1], ks[2]))
        self._slot_uniforms_setup()
        return self

    def _gathered(self, which):
        """the resampled state x[which][anc] as the step model's first argument (lazy: the gather is fused)"""
        if self.tuple_state is not None:
            return self.tuple_state(Gathered(row, self.anc) for row in self.x[which])
        return Gathered(self.x[which], self.anc)

    def _launch_vm(self, t):
        n = self.n
        k_prop = self.step_keys[t][0]
        obs = ChoiceMap.empty().set(self.obs_addr, self.ys[t])
        if t == 0:
            prog = self.p_init
            leaves = prog.leaves((), obs, self._noise_leaves(t, prog)) if self.noise_ahead else prog.leaves((), obs)
        else:
            g = self._gathered((t - 1) % 2) if self.rejuvenate is None else self.xm[t % 2]
            prog = self.p_step
            a = (g,) + tuple(self.step_extra(t))
            leaves = prog.leaves(a, obs, self._noise_leaves(t, prog)) if self.noise_ahead else prog.leaves(a, obs)
        bufs = [None] * len(prog.comp.outputs)
        if self.tuple_state is not None:
            for d, o in enumerate(prog.ro[1]):
                bufs[o[1]] = self.x_store[t % 2][d:d + 1]
        else:
            bufs[prog.ro[1]] = self.x_store[t % 2]
        w = t % 2 if self.fuse else 0
        bufs[prog.wo[1]] = self.lw_pp[w].reshape(1, n)
        prog.comp.run(leaves, (n,), lazy_split(k_prop, n), red_out=self.partials_pp[w], out_buffers=bufs,
                      tile_stats=(self.tile_agg_pp[w], self.shift) if self.tile_stats else None,
                      resample_in=self._resample_in(t) if (self.fuse and t >= 1 and self.rejuvenate is None) else None)

    def _resample_in(self, t):
        """gmx_run_args.rs of the launch of step t (>= 1) that gathers: resample step t-1's weights first"""
        kh = self.step_keys[t - 1][1].host()
        w = (t - 1) % 2
        return dict(lw=self.lw_pp[w], tile_max=self.partials_pp[w], tile_agg=self.tile_agg_pp[w], shift=self.shift,
                    key=(int(kh[0]), int(kh[1])), tag=1 + (t - 1) % _lib.ANC_TAG_MAX, max_out=self.maxs[t - 1:t],
                    total_out=self.totals[t - 1:t], status=self.rs_status)

    def _chain_prog(self, t):
        if t == 0:
            return self.p_init
        if self.fuse_mh:
            return self.p_mhvm_init if t == 1 else self.p_mhvm_step
        return self.p_step

    def _launch_mh(self, t):
        """The MH move on the resampled particles of step t-1 (t >= 1): reads x_{t-1}[anc] and, for
        t >= 2, the state xm[(t-1) % 2][anc] that x_{t-1} was extended from; writes xm[t % 2]."""
        n = self.n
        k_mh = self.step_keys[t][2]
        ch = ChoiceMap.empty().set(self.obs_addr, self.ys[t - 1]).set(self.state_addr,
                                                                      Gathered(self.x[(t - 1) % 2], self.anc))
        if t == 1:
            prog, leaves = self.p_mh_init, self.p_mh_init.leaves((), ch, self.rejuvenate)
        else:
            prog = self.p_mh_step
            a = Gathered(self.xm[(t - 1) % 2], self.anc)
            leaves = prog.leaves((a,) + tuple(self.step_extra(t - 1)), ch, self.rejuvenate)
        bufs = [None] * len(prog.comp.outputs)
        bufs[prog.ro[1]] = self.xm_store[t % 2]
        bufs[prog.ao[1]] = self.accept.reshape(1, n)
        prog.comp.run(leaves, (n,), lazy_split(k_mh, n), out_buffers=bufs)

    def _launch_mhvm(self, t):
        """step t >= 1 as ONE launch: the MH move on the resampled particles of step t-1 (as _launch_mh), then the
        extension to step t (as _launch_vm) from the moved state"""
        n = self.n
        k_mh, k_prop = self.step_keys[t][2], self.step_keys[t][0]
        ch = ChoiceMap.empty().set(self.obs_addr, self.ys[t - 1]).set(self.state_addr,
                                                                      Gathered(self.x[(t - 1) % 2], self.anc))
        obs = ChoiceMap.empty().set(self.obs_addr, self.ys[t])
        ex = tuple(self.step_extra(t))
        kw = k_prop.host()
        if t == 1:
            prog = self.p_mhvm_init
            leaves = prog.leaves((), ch, self.rejuvenate, ex, obs, (int(kw[0]), int(kw[1])),
                                 self._noise_leaves(t, prog) if self.noise_ahead else ())
        else:
            prog = self.p_mhvm_step
            a = Gathered(self.xm[(t - 1) % 2], self.anc)
            leaves = prog.leaves((a,) + tuple(self.step_extra(t - 1)), ch, self.rejuvenate, ex, obs,
                                 (int(kw[0]), int(kw[1])), self._noise_leaves(t, prog) if self.noise_ahead else ())
        bufs = [None] * len(prog.comp.outputs)
        bufs[prog.mo[1]] = self.xm_store[t % 2]
        bufs[prog.ao[1]] = self.accept.reshape(1, n)
        bufs[prog.ro[1]] = self.x_store[t % 2]
        w = t % 2 if self.fuse else 0
        bufs[prog.wo[1]] = self.lw_pp[w].reshape(1, n)
        prog.comp.run(leaves, (n,), lazy_split(k_mh, n), red_out=self.partials_pp[w], out_buffers=bufs,
                      tile_stats=(self.tile_agg_pp[w], self.shift) if self.tile_stats else None,
                      resample_in=self._resample_in(t) if self.fuse else None)

    def _rows(self, t) -> int:
        """partial rows the site program of step t wrote"""
        prog = self.p_init if t == 0 else (self.p_step if not self.fuse_mh else
                                           (self.p_mhvm_init if t == 1 else self.p_mhvm_step))
        return int(_lib.get().c.gmx_program_grid(prog.comp.handle, self.n))

    def _launch_cdf(self, t):
        be = _lib.get()
        w = t % 2 if self.fuse else 0          # (a one-launch sweep alternates two sets of log-weights / partials)
        be.check(be.c.gmx_weight_cdf(be.ptr(self.lw_pp[w]), self.n, self.shift, be.ptr(self.partials_pp[w]),
                                     self._rows(t), be.ptr(self.maxs[t:t + 1]), be.ptr(self.cdf),
                                     be.ptr(self.totals[t:t + 1]), be.ptr(self.ws), be.stream()),
                 "gmx_weight_cdf")

    def _launch_anc(self, t):
        be = _lib.get()
        kh = self.step_keys[t][1].host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        if self.kind == MULTINOMIAL and self.n >= MULTINOMIAL_GUIDED_MIN:
            if getattr(self, "mn_ws", None) is None:
                self.mn_ws = torch.zeros(((be.c.gmx_multinomial_workspace(self.n) + 3) // 4,), dtype=torch.int32,
                                         device=be.device)
            be.check(be.c.gmx_multinomial(kk, be.ptr(self.cdf), self.n, be.ptr(self.totals[t:t + 1]), self.n,
                                          be.ptr(self.anc), be.ptr(self.mn_ws), be.stream()), "gmx_multinomial")
            return
        be.check(be.c.gmx_ancestors(self.kind, kk, be.ptr(self.cdf), self.n, 0, be.ptr(self.totals[t:t + 1]),
                                    self.n, 0, self.n, be.ptr(self.anc), be.stream()), "gmx_ancestors")

    def _slot_uniforms_setup(self):
        """Stratified resampling draws one uniform per SLOT, keyed by the step's resampling key and the slot number —
        nothing the chain produces.  In the noise-ahead form they are drawn on the background stream with the steps'
        normals (gmx_slot_uniforms, one 2-D launch per group of steps) and the resampler reads them
        (gmx_resample_tiles_u): the same ancestors, one Threefry block per slot-edge evaluation less on the chain
        (the one-stream form draws them inside the resampler)."""
        self.ubuf = None
        if not (self.noise_ahead and self.kind in (STRATIFIED, MULTINOMIAL_TILED, MULTINOMIAL_SORTED) and self.fused
                and self.tile_stats):
            return
        dev = self.zbuf.device
        # (the sorted multinomial's row is its whole order-statistics table: gmx_sorted_uniforms_words(n) words)
        row_words = int(_lib.get().c.gmx_sorted_uniforms_words(self.n)) if self.kind == MULTINOMIAL_SORTED else self.n
        self.ubuf = torch.zeros((self.noise_ring, self.noise_group, row_words), dtype=torch.int32, device=dev)
        self._u_keys = []
        for t0, t1 in self.noise_groups:
            # stratified: the resampling key itself; the two-stage multinomial: its first child (stage 1's key)
            rk = (lambda t: split(self.step_keys[t][1], 2)[0]) if self.kind == MULTINOMIAL_TILED else (lambda t: self.step_keys[t][1])
            ks = np.stack([rk(t).host() for t in range(t0, t1)]).astype(np.uint32)
            self._u_keys.append(torch.from_numpy(ks.view(np.int32)).to(dev))
        self._u_pad = int(self.NOISE_LDS_PAD)
        if self.kind == MULTINOMIAL_SORTED:       # its two kernels wait on memory, not on the vector ALUs: more of them per CU
            self._u_pad = self.SORTED_LDS_PAD

    def _launch_group_extras(self, g):
        if getattr(self, "ubuf", None) is None:
            return
        be = _lib.get()
        t0, t1 = self.noise_groups[g]
        half, row = self.noise_slot[t0]
        if self.kind == MULTINOMIAL_SORTED:
            be.check(be.c.gmx_sorted_uniforms(be.ptr(self._u_keys[g]), t1 - t0, self.n,
                                              be.ptr(self.ubuf[half, row:row + (t1 - t0)]), self._u_pad, be.stream()),
                     "gmx_sorted_uniforms")
            return
        be.check(be.c.gmx_slot_uniforms(be.ptr(self._u_keys[g]), t1 - t0, self.n, be.ptr(self.ubuf[half, row:row + (t1 - t0)]),
                                        self._u_pad, be.stream()), "gmx_slot_uniforms")

    def _launch_resample(self, t):
        be = _lib.get()
        kh = self.step_keys[t][1].host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        if self.kind == MULTINOMIAL_TILED:
            if not self.tile_stats:
                be.check(be.c.gmx_tile_stats(be.ptr(self.lw), self.n, self.shift, be.ptr(self.partials),
                                             be.ptr(self.tile_agg), be.stream()), "gmx_tile_stats")
            u_d = None
            if getattr(self, "ubuf", None) is not None:
                half, row = self.noise_slot[t]
                u_d = be.ptr(self.ubuf[half, row])
            be.check(be.c.gmx_multinomial_tiled(kk, be.ptr(self.lw), self.n, self.shift, be.ptr(self.partials),
                                                be.ptr(self.tile_agg), u_d, be.ptr(self.maxs[t:t + 1]),
                                                be.ptr(self.totals[t:t + 1]), be.ptr(self.anc), be.ptr(self.mnt_ws),
                                                -1 if t == 0 else (t & 1),      # count buffers alternate: one memset per sweep
                                                be.stream()), "gmx_multinomial_tiled")
            return
        if self.kind == MULTINOMIAL_SORTED:
            if not self.tile_stats:
                be.check(be.c.gmx_tile_stats(be.ptr(self.lw), self.n, self.shift, be.ptr(self.partials),
                                             be.ptr(self.tile_agg), be.stream()), "gmx_tile_stats")
            table, ready = self.sorted_ws, 0
            if getattr(self, "ubuf", None) is not None:        # drawn ahead on the background stream
                half, row = self.noise_slot[t]
                table, ready = self.ubuf[half, row], 1
            be.check(be.c.gmx_resample_sorted(kk, be.ptr(self.lw), self.n, self.shift, be.ptr(self.partials),
                                              be.ptr(self.tile_agg), be.ptr(table), ready, be.ptr(self.maxs[t:t + 1]),
                                              be.ptr(self.totals[t:t + 1]), be.ptr(self.anc), be.stream()),
                     "gmx_resample_sorted")
            return
        if getattr(self, "ubuf", None) is not None:
            half, row = self.noise_slot[t]
            be.check(be.c.gmx_resample_tiles_u(self.kind, kk, be.ptr(self.lw), self.n, self.shift,
                                               be.ptr(self.partials), be.ptr(self.tile_agg),
                                               be.ptr(self.ubuf[half, row]), be.ptr(self.maxs[t:t + 1]),
                                               be.ptr(self.totals[t:t + 1]), be.ptr(self.anc), be.stream()),
                     "gmx_resample_tiles_u")
            return
        if self.tile_stats:        # tile maxima = the workgroup maxima the site program left in partials[0]
            w = t % 2 if self.fuse else 0
            be.check(be.c.gmx_resample_tiles(self.kind, kk, be.ptr(self.lw_pp[w]), self.n, self.shift,
                                             be.ptr(self.partials_pp[w]), be.ptr(self.tile_agg_pp[w]),
                                             be.ptr(self.maxs[t:t + 1]),
                                             be.ptr(self.totals[t:t + 1]), be.ptr(self.anc), be.stream()),
                     "gmx_resample_tiles")
            return
        be.check(be.c.gmx_resample(self.kind, kk, be.ptr(self.lw), self.n, self.shift, be.ptr(self.partials),
                                   self._rows(t), be.ptr(self.maxs[t:t + 1]),
                                   be.ptr(self.totals[t:t + 1]), be.ptr(self.anc), be.ptr(self.rs_ws),
                                   be.stream()), "gmx_resample")

    def _chain_step(self, t, skip_vm=False):
        """everything step t launches on the chain (noise-ahead form): site program', then the resampler"""
        if not skip_vm:
            if t >= 1 and self.fuse_mh:
                self._launch_mhvm(t)
            else:
                self._launch_vm(t)
        if self.fuse and t < self.T - 1:
            return                         # step t's weights are resampled by step t + 1's launch itself
        if self.fused:
            self._launch_resample(t)
        else:
            self._launch_cdf(t)
            self._launch_anc(t)

    def enqueue(self, skip_vm=False):
        """Issue every launch of the sweep on the current stream (no syncs, no allocations).
        skip_vm=True leaves the site-program launches out (the resampling kernels then run on the
        previous sweep's log-weights): bench.py times that variant to get the site program's cost
        IN the sweep as a difference."""
        if self.noise_ahead:
            return self._enqueue_noise_ahead(skip_vm)
        for t in range(self.T):
            if t >= 1 and self.fuse_mh:
                if not skip_vm:
                    self._launch_mhvm(t)
            else:
                if t >= 1 and self.rejuvenate is not None:
                    self._launch_mh(t)
                if not skip_vm:
                    self._launch_vm(t)
            if self.fuse and t < self.T - 1:
                continue                   # step t's weights are resampled by step t + 1's launch itself
            if self.fused:
                self._launch_resample(t)
            else:
                self._launch_cdf(t)
                self._launch_anc(t)

    def kernel_timers(self):
        """Representative single launches (a mid-sweep step) for per-kernel timing in bench.py."""
        t = max(1, self.T // 2)
        out = {"k_vm": lambda: self._launch_vm(t)}
        if self.noise_ahead:
            out["k_noise"] = lambda: self._launch_noise(t)
        if self.fuse:
            pass                           # (the resampler is the prologue of k_vm's launch)
        elif self.fused and self.tile_stats:
            out["k_offspring_tile"] = lambda: self._launch_resample(t)
        elif self.fused:
            out["resample(k_tile_stats+k_offspring_tile)"] = lambda: self._launch_resample(t)
        else:
            out["k_weight_cdf"] = lambda: self._launch_cdf(t)
            out["k_ancestors"] = lambda: self._launch_anc(t)
        return out

    def capture(self):
        """Capture enqueue() into a hipGraph (launch-bound: ~5 nodes per step)."""
        be = _lib.get()
        from ctypes import c_void_p
        s = torch.cuda.Stream(device=be.device)
        s.wait_stream(torch.cuda.current_stream(be.device))
        with torch.cuda.stream(s):
            self.enqueue()          # warm-up outside capture (program upload, lazy init)
            s.synchronize()
            be.check(be.c.gmx_capture_begin(be.stream()), "gmx_capture_begin")
            try:
                self.enqueue()
            finally:
                h = c_void_p()
                rc = be.c.gmx_capture_end(be.stream(), h)
            be.check(rc, "gmx_capture_end")
        torch.cuda.current_stream(be.device).wait_stream(s)
        self.graph = h
        return self

    def launch(self):
        be = _lib.get()
        if self.graph is None:
            self.enqueue()
        else:
            be.check(be.c.gmx_graph_launch(self.graph, be.stream()), "gmx_graph_launch")

    def _check_valid(self):
        """the fused resampling prologue's sticky timeout word: a workgroup that gave up waiting clamps stale words into
        indices, so nothing of such a sweep may be read.  The word is STICKY for the object's lifetime (nothing in the
        captured sweep clears it: one timed-out poll condemns every later replay until `reset_status()`), which is the
        safe side: a launch that was not resident once will not be the next time either"""
        if self.fuse and int(self.rs_status.item()) != 0:
            raise RuntimeError("BootstrapSweep: a workgroup's wait for its ancestors timed out (the launch was not "
                               "resident at once?): the sweep's results are not valid")

    def reset_status(self):
        """clear the sticky timeout word (after the cause — e.g. another process holding compute units — is gone)"""
        if getattr(self, "rs_status", None) is not None:
            self.rs_status.zero_()

    def log_ml(self) -> float:
        """sum_t [ ref(M_t) + log(total_t * 2^-shift) - log N ] in float64 (synchronises)."""
        self._check_valid()
        m = np.array([cdf_reference(v) for v in self.maxs.cpu().numpy()], dtype=np.float64)
        tot = self.totals.cpu().numpy().view(np.uint64).astype(np.float64)
        if np.any(tot == 0):             # a step where no particle carried any mass
            return -math.inf
        return float(np.sum(m + np.log(tot) - self.shift * math.log(2.0) - math.log(self.n)))

    def state(self):
        """(x_T particles before the last resampling, log-weights, last ancestors)."""
        self._check_valid()
        return self.x[(self.T - 1) % 2], self.lw_pp[(self.T - 1) % 2 if self.fuse else 0], self.anc
