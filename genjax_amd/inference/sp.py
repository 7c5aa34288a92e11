"""Posterior targets and the GenSP algorithm interface
(src/genjax/_src/inference/sp.py: Target :52-94, Algorithm :111-199,
Marginal :207-273)."""
from __future__ import annotations

from ..core.choice_map import ChoiceMap, Selection
from ..core.generative import GenerativeFunction


class Marginal(GenerativeFunction):
    """sp.py:207-252.  Container only on the round-1 hot path: `Target` must
    reject it (tests/inference/test_smc.py:89-106); its sampler is next-tier
    (SURVEY.md §8f item 4)."""

    def __init__(self, gen_fn, selection=None, algorithm=None):
        self.gen_fn = gen_fn
        self.selection = selection if selection is not None else Selection.all()
        self.algorithm = algorithm

    def random_weighted(self, key, *args):
        raise NotImplementedError("Marginal.random_weighted: SURVEY.md §8(f) item 4 (next tier)")

    def estimate_logpdf(self, key, v, *args):
        raise NotImplementedError("Marginal.estimate_logpdf: SURVEY.md §8(f) item 4 (next tier)")


def marginal(selection=None, algorithm=None):
    def decorator(gen_fn):
        return Marginal(gen_fn, selection, algorithm)
    return decorator


class Target:
    """An unnormalised posterior: (generative function, arguments, constraint)
    (sp.py:52-94)."""

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
