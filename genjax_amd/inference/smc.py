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
from ..static import DistributionTrace, StaticTrace
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
        return DistributionTrace(tr.gen_fn, leaf(tr.args), leaf(tr.value), leaf(tr.score))
    return StaticTrace(tr.gen_fn, leaf(tr.args), leaf(tr.retval),
                       OrderedDict((a, trace_map(s, fn)) for a, s in tr.subtraces.items()))


def stack_traces(trs, tr_one, axis: int = 0):
    """tree_map(stack_to_first_dim, trs, tr_one) (smc.py:56-68, :343-345): append ONE trace after the K-1 particles
    of a batched trace along the particle axis — axis 0 for one key, axis `len(key.shape)` under a batch of keys
    (the reference's `vmap` of run_csmc: leaves [*keys, K-1, ...] and [*keys, ...])."""
    def cat(a, b):
        a = engine.materialize(a)
        b = engine.materialize(b)
        if not isinstance(a, torch.Tensor):
            return a                                   # static argument (same in both traces)
        b = torch.as_tensor(b, dtype=a.dtype, device=a.device) if not isinstance(b, torch.Tensor) else b.to(a.dtype)
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
