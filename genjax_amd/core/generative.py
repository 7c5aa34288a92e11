"""Generative function interface: abstract types kept from the reference so
models and inference code read the same
(src/genjax/_src/core/generative/generative_function.py:72-689, 1557-1689;
concepts.py:95-164; requests.py:48-95; interpreters/incremental.py Diff).
"""
from __future__ import annotations

from typing import Any

from .choice_map import ChoiceMap, Selection


# ---------------------------------------------------------------------------
# change tags (incremental.py:57-120)
# ---------------------------------------------------------------------------
class _ChangeTangent:
    def __repr__(self):
        return type(self).__name__


class _NoChange(_ChangeTangent):
    pass


class _UnknownChange(_ChangeTangent):
    pass


NoChange = _NoChange()
UnknownChange = _UnknownChange()


class Diff:
    """A value tagged with whether it changed since the trace was made."""
    __slots__ = ("primal", "tangent")

    def __init__(self, primal, tangent):
        self.primal, self.tangent = primal, tangent

    def get_primal(self): return self.primal
    def get_tangent(self): return self.tangent

    @staticmethod
    def tree_diff(tree, tangent_tree):
        """incremental.py:122-150: a primal tree and a tree of ChangeTangents of the same structure, zipped"""
        return _tree_map2(lambda p_, t_: Diff(p_, t_), tree, tangent_tree)

    @staticmethod
    def no_change(tree):
        """incremental.py:152-173: every leaf NoChange — an existing Diff's tangent is REPLACED"""
        return _tree_map(lambda v: Diff(v.primal, NoChange) if isinstance(v, Diff) else Diff(v, NoChange), tree)

    @staticmethod
    def unknown_change(tree):
        return _tree_map(lambda v: Diff(v.primal, UnknownChange) if isinstance(v, Diff) else Diff(v, UnknownChange), tree)

    @staticmethod
    def tree_primal(tree):
        return _tree_map(lambda v: v.primal if isinstance(v, Diff) else v, tree)

    @staticmethod
    def tree_tangent(tree):
        """incremental.py:218-236: a value that is not a Diff reads as NoChange (the docstring there says UnknownChange;
        the code and tests/core/interpreters/test_incremental.py:49-54 say NoChange).  An edit never sees one:
        check_argdiffs refuses a tree with bare leaves, as the reference's `Argdiffs` type does (concepts.py:66-81)."""
        return _tree_map(lambda v: v.tangent if isinstance(v, Diff) else NoChange, tree)

    @staticmethod
    def is_diff(v) -> bool:
        return isinstance(v, Diff)

    @staticmethod
    def is_change_tangent(v) -> bool:
        return isinstance(v, _ChangeTangent)

    @staticmethod
    def static_check_no_change(tree) -> bool:
        ok = True

        def visit(v):
            nonlocal ok
            if not (isinstance(v, Diff) and v.tangent is NoChange):
                ok = False
            return v
        _tree_map(visit, tree)
        return ok

    @staticmethod
    def static_check_tree_diff(tree) -> bool:
        ok = True

        def visit(v):
            nonlocal ok
            if not isinstance(v, Diff):
                ok = False
            return v
        _tree_map(visit, tree)
        return ok

    def __repr__(self):
        return f"Diff({self.primal!r}, {self.tangent!r})"


def check_argdiffs(argdiffs):
    """concepts.py:66-81: `Argdiffs` is a tree whose every leaf is a Diff (a runtime type check in the reference)"""
    if argdiffs is not None and not Diff.static_check_tree_diff(argdiffs):
        raise TypeError("argdiffs must be a tree of Diff values: wrap the arguments with Diff.no_change(...) / "
                        "Diff.unknown_change(...) (the reference's Argdiffs type, concepts.py:66-81)")
    return argdiffs


def _tree_map2(fn, a, b):
    if isinstance(a, tuple):
        if not isinstance(b, tuple) or len(a) != len(b):
            raise ValueError("Diff.tree_diff: the two trees differ in structure")
        return tuple(_tree_map2(fn, x, y) for x, y in zip(a, b))
    if isinstance(a, list):
        if not isinstance(b, list) or len(a) != len(b):
            raise ValueError("Diff.tree_diff: the two trees differ in structure")
        return [_tree_map2(fn, x, y) for x, y in zip(a, b)]
    if isinstance(a, dict):
        if not isinstance(b, dict) or a.keys() != b.keys():
            raise ValueError("Diff.tree_diff: the two trees differ in structure")
        return {k: _tree_map2(fn, a[k], b[k]) for k in a}
    return fn(a, b)


def _tree_map(fn, tree):
    if isinstance(tree, Diff):
        return fn(tree)
    if isinstance(tree, tuple):
        return tuple(_tree_map(fn, t) for t in tree)
    if isinstance(tree, list):
        return [_tree_map(fn, t) for t in tree]
    if isinstance(tree, dict):
        return {k: _tree_map(fn, t) for k, t in tree.items()}
    return fn(tree)


# ---------------------------------------------------------------------------
# traces
# ---------------------------------------------------------------------------
class Trace:
    """Trace ABC (generative_function.py:72-230)."""

    def get_args(self): raise NotImplementedError
    def get_retval(self): raise NotImplementedError
    def get_score(self): raise NotImplementedError
    def get_choices(self) -> ChoiceMap: raise NotImplementedError
    def get_gen_fn(self): raise NotImplementedError

    def get_sample(self):
        return self.get_choices()

    def edit(self, key, request, argdiffs=None):
        if argdiffs is None:
            argdiffs = Diff.no_change(self.get_args())
        return request.edit(key, self, check_argdiffs(argdiffs))

    def update(self, key, constraint, argdiffs=None):
        if argdiffs is None:
            argdiffs = Diff.no_change(self.get_args())
        return self.get_gen_fn().update(key, self, constraint, check_argdiffs(argdiffs))

    def project(self, key, selection):
        return self.get_gen_fn().project(key, self, selection)

    @property
    def batch_shape(self):
        s = self.get_score()
        return tuple(getattr(s, "shape", ()))


# ---------------------------------------------------------------------------
# edit requests
# ---------------------------------------------------------------------------
class NotSupportedEditRequest(Exception):
    pass


class EditRequest:
    """EditRequest / PrimitiveEditRequest (concepts.py:95-150): primitive
    requests are answered by the trace's generative function."""

    def edit(self, key, tr: Trace, argdiffs):
        return tr.get_gen_fn().edit(key, tr, self, argdiffs)

    def dimap(self, *, pre=lambda v: v, post=lambda v: v):
        return DiffAnnotate(self, argdiff_fn=pre, retdiff_fn=post)


PrimitiveEditRequest = EditRequest


class Update(EditRequest):
    """generative_function.py:1687-1689"""
    __match_args__ = ("constraint",)

    def __init__(self, constraint: ChoiceMap):
        self.constraint = constraint

    def __repr__(self):
        return f"Update({self.constraint!r})"


class Regenerate(EditRequest):
    """requests.py:63-65"""
    __match_args__ = ("selection",)

    def __init__(self, selection: Selection):
        self.selection = selection


class IndexRequest(EditRequest):
    """concepts.py:153-164: edit ONE index of a vector combinator's trace with a sub-request
    (`Vmap.edit_index`, vmap.py:277-332).  `idx` is a Python int (the unrolled plate picks the element
    at trace time) or an integer tensor with one index PER PARTICLE (every element is then edited
    in the program and selected where idx == j: n times the work, for small plates)."""
    __match_args__ = ("idx", "request")

    def __init__(self, idx, request: EditRequest):
        self.idx = idx if hasattr(idx, "shape") and tuple(getattr(idx, "shape", ())) != () else int(idx)
        self.request = request


class VectorRequest(EditRequest):
    """The backward request a vector combinator's `Regenerate` edit returns (scan.py:504): per-element
    sub-requests stacked along the leading axis.  Container only."""
    __match_args__ = ("request",)

    def __init__(self, request):
        self.request = request


class EmptyRequest(EditRequest):
    """requests.py:48-60: no change requested; re-scores only what changed args force."""

    def edit(self, key, tr: Trace, argdiffs):
        if Diff.static_check_no_change(argdiffs):
            return tr, 0.0, Diff.no_change(tr.get_retval()), EmptyRequest()
        return Update(ChoiceMap.empty()).edit(key, tr, argdiffs)


class DiffAnnotate(EditRequest):
    """requests.py:68-95"""

    def __init__(self, request, argdiff_fn=lambda v: v, retdiff_fn=lambda v: v):
        self.request, self.argdiff_fn, self.retdiff_fn = request, argdiff_fn, retdiff_fn

    def edit(self, key, tr, argdiffs):
        new_tr, w, retdiff, bwd = self.request.edit(key, tr, self.argdiff_fn(argdiffs))
        return new_tr, w, self.retdiff_fn(retdiff), bwd


# ---------------------------------------------------------------------------
# generative functions
# ---------------------------------------------------------------------------
class GenerativeFunction:
    """GenerativeFunction ABC (generative_function.py:232-689)."""

    # abstract interface ---------------------------------------------------------
    def simulate(self, key, args): raise NotImplementedError
    def assess(self, sample: ChoiceMap, args): raise NotImplementedError
    def generate(self, key, constraint: ChoiceMap, args): raise NotImplementedError
    def project(self, key, trace, selection): raise NotImplementedError
    def edit(self, key, trace, edit_request, argdiffs): raise NotImplementedError

    # derived (generative_function.py:611-689) -------------------------------------
    def update(self, key, trace, constraint: ChoiceMap, argdiffs):
        tr, w, rd, bwd = Update(constraint).edit(key, trace, check_argdiffs(argdiffs))
        assert isinstance(bwd, Update), type(bwd)
        return tr, w, rd, bwd.constraint

    def importance(self, key, constraint: ChoiceMap, args):
        return self.generate(key, constraint, args)

    def propose(self, key, args):
        tr = self.simulate(key, args)
        return tr.get_choices(), tr.get_score(), tr.get_retval()

    def handle_kwargs(self):
        raise NotImplementedError(f"{type(self).__name__} does not accept keyword arguments")

    def __call__(self, *args, **kwargs) -> "GenerativeFunctionClosure":
        return GenerativeFunctionClosure(self, args, kwargs)

    # sugar ---------------------------------------------------------------------------
    def marginal(self, *, selection=None, algorithm=None):
        from ..inference.sp import Marginal
        return Marginal(self, selection if selection is not None else Selection.all(), algorithm)

    def vmap(self, *, in_axes=0):
        from ..combinators import Vmap
        return Vmap(self, in_axes)

    def repeat(self, *, n: int):
        from ..combinators import repeat
        return repeat(n=n)(self)

    def scan(self, *, n=None):
        from ..combinators import Scan
        return Scan(self, n)

    def mask(self):
        from ..combinators import MaskCombinator
        return MaskCombinator(self)

    def masked_iterate(self):
        from ..combinators import masked_iterate
        return masked_iterate()(self)

    def masked_iterate_final(self):
        from ..combinators import masked_iterate_final
        return masked_iterate_final()(self)

    def iterate(self, *, n: int):
        from ..combinators import iterate
        return iterate(n=n)(self)

    def iterate_final(self, *, n: int):
        from ..combinators import iterate_final
        return iterate_final(n=n)(self)

    def accumulate(self):
        from ..combinators import accumulate
        return accumulate()(self)

    def reduce(self):
        from ..combinators import reduce
        return reduce()(self)


def _args_equal(a, b) -> bool:
    """structural equality of argument tuples / dicts that may hold tensors"""
    import torch
    if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)):
        return len(a) == len(b) and all(_args_equal(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_args_equal(a[k], b[k]) for k in a)
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        return isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.shape == b.shape and bool((a == b).all())
    return a == b


class GenerativeFunctionClosure(GenerativeFunction):
    """`gen_fn(*args)`: `closure @ addr` traces the callee at addr inside a
    `@gen` function; `closure(key)` simulates and returns the return value
    (generative_function.py:1557-1684)."""

    def __init__(self, gen_fn, args, kwargs):
        self.gen_fn, self.args, self.kwargs = gen_fn, tuple(args), dict(kwargs)

    def _target(self):
        """(callee, args) with keyword arguments folded the way the reference
        does: a kwarg-handling callee receives (args, kwargs)."""
        if self.kwargs:
            return self.gen_fn.handle_kwargs(), (self.args, self.kwargs)
        return self.gen_fn, self.args

    def __matmul__(self, addr):
        from ..static import trace
        gf, args = self._target()
        return trace(addr, gf, args)

    def handle_kwargs(self):
        """the closure in its (args, kwargs)-taking form (generative_function.py:1597-1611); idempotent"""
        if not self.kwargs and getattr(self, "_kwarged_form", False):
            return self
        out = GenerativeFunctionClosure(self.gen_fn.handle_kwargs(), (self.args, dict(self.kwargs)), {})
        out._kwarged_form = True
        return out

    def __eq__(self, other):
        return (isinstance(other, GenerativeFunctionClosure) and self.gen_fn is other.gen_fn
                and _args_equal(self.args, other.args) and _args_equal(self.kwargs, other.kwargs))

    __hash__ = object.__hash__

    def __call__(self, key, *args, **kwargs):
        if getattr(self, "_kwarged_form", False):      # (args, kwargs) payload: extra arguments join their own kind
            pos, kw = self.args
            return self.gen_fn.simulate(key, (tuple(pos) + args, {**kw, **kwargs})).get_retval()
        full = GenerativeFunctionClosure(self.gen_fn, self.args + args, {**self.kwargs, **kwargs})
        gf, a = full._target()
        return gf.simulate(key, a).get_retval()

    def simulate(self, key, args):
        gf, a = GenerativeFunctionClosure(self.gen_fn, self.args + tuple(args), self.kwargs)._target()
        return gf.simulate(key, a)

    def generate(self, key, constraint, args):
        gf, a = GenerativeFunctionClosure(self.gen_fn, self.args + tuple(args), self.kwargs)._target()
        return gf.generate(key, constraint, a)

    def assess(self, sample, args):
        gf, a = GenerativeFunctionClosure(self.gen_fn, self.args + tuple(args), self.kwargs)._target()
        return gf.assess(sample, a)

    def project(self, key, trace, selection):
        return self.gen_fn.project(key, trace, selection)

    def edit(self, key, trace, edit_request, argdiffs):
        return self.gen_fn.edit(key, trace, edit_request, argdiffs)


Arguments = tuple
Argdiffs = Any
Retdiff = Any
Score = Any
Weight = Any
