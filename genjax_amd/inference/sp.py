"""Posterior targets and the GenSP algorithm interface
(src/genjax/_src/inference/sp.py: Target :52-94, Algorithm :111-199,
Marginal :207-273)."""
from __future__ import annotations

from ..core.choice_map import ChoiceMap, Selection
from ..core.generative import GenerativeFunction


class Marginal(GenerativeFunction):
    """The marginal of a generative function over a selection (sp.py:207-252).
    `Target` rejects it as a model (tests/inference/test_smc.py:89-106); as a
    proposal `q` it is a SampleDistribution: `random_weighted` / `estimate_logpdf`."""

    def __init__(self, gen_fn, selection=None, algorithm=None):
        self.gen_fn = gen_fn
        self.selection = selection if selection is not None else Selection.all()
        self.algorithm = algorithm

    def random_weighted(self, key, *args):
        """sp.py:217-240, literally: simulate, keep the selected choices, weight =
        project(trace, ~selection); with an inner algorithm the weight is its estimate of the reciprocal
        normalising constant of Target(gen_fn, args, selected choices) — this is how algorithms nest."""
        from ..random import split
        key, sub_key = split(key)
        tr = self.gen_fn.simulate(sub_key, tuple(args))
        choices = tr.get_choices()
        latent_choices = choices.filter(self.selection)
        key, sub_key = split(key)
        weight = tr.project(sub_key, ~self.selection)
        if self.algorithm is None:
            return weight, latent_choices
        # sp.py:229-238: the inner algorithm estimates 1 / Z of the posterior over the marginalised choices
        target = Target(self.gen_fn, tuple(args), latent_choices)
        other_choices = choices.filter(~self.selection)
        Z = self.algorithm.estimate_reciprocal_normalizing_constant(key, target, other_choices, weight)
        return Z, latent_choices

    def estimate_logpdf(self, key, v, *args):
        """sp.py:242-254"""
        if self.algorithm is None:
            _, weight = self.gen_fn.importance(key, v, tuple(args))
            return weight
        return self.algorithm.estimate_normalizing_constant(key, Target(self.gen_fn, tuple(args), v))

    # -- the Distribution GFI over the two methods above (a Marginal is a SampleDistribution[ChoiceMap]: sp.py:207,
    #    distribution.py:108-147, 398-419): `marginal_model.simulate(key, ())`, `.importance(key, obs, ())` (ravi_stack.ipynb
    #    c4 / c5) — the VALUE of its trace is the choice map of the selected addresses
    def simulate(self, key, args):
        w, v = self.random_weighted(key, *tuple(args))
        return MarginalTrace(self, tuple(args), v, w)

    def generate(self, key, constraint, args):
        v = constraint.get_value() if constraint is not None else None
        if v is None:                                    # (distribution.py:123-127: nothing at the root: unconstrained)
            return self.simulate(key, args), 0.0
        if not isinstance(v, ChoiceMap):
            raise TypeError("a Marginal's value is the ChoiceMap of its selected addresses: constrain it with C.v(choice_map)")
        w = self.estimate_logpdf(key, v, *tuple(args))
        return MarginalTrace(self, tuple(args), v, w), w

    importance = generate

    def assess(self, sample, args):
        from ..random import key as _key
        v = sample.get_value()
        if not isinstance(v, ChoiceMap):
            raise TypeError("a Marginal's value is the ChoiceMap of its selected addresses: assess(C.v(choice_map), args)")
        return self.estimate_logpdf(_key(0), v, *tuple(args)), v          # (distribution.py:403: a dummy key)


class MarginalTrace:
    """DistributionTrace of a SampleDistribution (distribution.py:59-82): arguments, the sampled choice map, the weight"""

    def __init__(self, gen_fn, args, value, score):
        self.gen_fn, self.args, self.value, self.score = gen_fn, args, value, score

    def get_args(self): return self.args
    def get_retval(self): return self.value
    def get_gen_fn(self): return self.gen_fn
    def get_score(self): return self.score
    def get_choices(self): return self.value          # (ChoiceMap.choice of a ChoiceMap is that map)


def marginal(selection=None, algorithm=None):
    def decorator(gen_fn):
        return Marginal(gen_fn, selection, algorithm)
    return decorator


class Target:
    """An unnormalised posterior: (generative function, arguments, constraint)
    (sp.py:52-94)."""

    __gmx_static__ = True        # as an argument of a proposal it is a host object, not a launch value

    def __init__(self, p, args, constraint: ChoiceMap):
        if isinstance(p, Marginal):
            raise TypeError("Target does not support Marginal generative functions.")   # sp.py:46-49
        self.p, self.args, self.constraint = p, tuple(args), constraint

    def importance(self, key, constraint: ChoiceMap):
        merged = self.constraint.merge(constraint)          # target's own observations win (sp.py:86)
        return self.p.importance(key, merged, self.args)

    def filter_to_unconstrained(self, choice_map: ChoiceMap) -> ChoiceMap:
        return choice_map.filter(~self.constraint.get_selection())     # sp.py:89-91

    def __getitem__(self, addr):
        return self.constraint[addr]


class Algorithm:
    """sp.py:111-199: inference algorithms are samplers with density estimates."""

    def random_weighted(self, key, *args):
        raise NotImplementedError

    def estimate_logpdf(self, key, v, *args):
        raise NotImplementedError

    def __call__(self, target):
        alg = self

        class _Closure:
            def __call__(self_, key):
                return alg.random_weighted(key, target)[1]
        return _Closure()

    def simulate(self, key, args):
        score, chm = self.random_weighted(key, *args)

        class _Tr:
            def get_retval(self_): return chm
            def get_score(self_): return score
        return _Tr()


SampleDistribution = Algorithm
