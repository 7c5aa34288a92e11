"""The reference's tests of the change-tangent algebra, restated against this package
(/root/reference/tests/core/interpreters/test_incremental.py, 6 tests: every one mirrored), plus the `Argdiffs` type rule of
/root/reference/src/genjax/_src/core/generative/concepts.py:66-81 (a tree whose every leaf is a Diff) as `edit` / `update`
enforce it here.  Host logic only: no launch."""
import pytest

import genjax_amd as genjax
from genjax_amd import Diff, NoChange, UnknownChange


class TestDiff:
    def test_no_nested_diffs(self):
        """test_incremental.py:24-30"""
        d1 = Diff.no_change(1.0)
        d2 = Diff.unknown_change(d1)
        assert not isinstance(d2.get_primal(), Diff)
        assert Diff.static_check_no_change(d1)
        assert not Diff.static_check_no_change(d2)
        # ... and the other way round: no_change REPLACES a tangent (incremental.py:152-173)
        d3 = Diff.no_change(d2)
        assert not isinstance(d3.get_primal(), Diff) and d3.get_tangent() is NoChange

    def test_tree_diff(self):
        """test_incremental.py:32-41"""
        primal_tree = {"a": 1, "b": [2, 3]}
        tangent_tree = {"a": NoChange, "b": [UnknownChange, NoChange]}
        result = Diff.tree_diff(primal_tree, tangent_tree)
        assert isinstance(result["a"], Diff)
        assert isinstance(result["b"][0], Diff)
        assert isinstance(result["b"][1], Diff)
        assert result["a"].get_tangent() == NoChange
        assert result["b"][0].get_tangent() == UnknownChange
        assert result["b"][1].get_tangent() == NoChange
        with pytest.raises(ValueError):
            Diff.tree_diff({"a": 1, "b": [2, 3]}, {"a": NoChange, "b": [NoChange]})

    def test_tree_primal(self):
        """test_incremental.py:43-46"""
        tree = {"a": Diff(1, NoChange), "b": [Diff(2, UnknownChange), 3]}
        assert Diff.tree_primal(tree) == {"a": 1, "b": [2, 3]}

    def test_tree_tangent(self):
        """test_incremental.py:48-54: a value that is not a Diff reads as NoChange"""
        tree = {"a": 1, "b": [Diff(2, UnknownChange), 3]}
        assert Diff.tree_tangent(tree) == {"a": NoChange, "b": [UnknownChange, NoChange]}

    def test_static_check_tree_diff(self):
        """test_incremental.py:56-60"""
        tree1 = {"a": Diff(1, NoChange), "b": [Diff(2, UnknownChange)]}
        tree2 = {"a": Diff(1, NoChange), "b": [2]}
        assert Diff.static_check_tree_diff(tree1)
        assert not Diff.static_check_tree_diff(tree2)

    def test_static_check_no_change(self):
        """test_incremental.py:62-66"""
        tree1 = {"a": Diff(1, NoChange), "b": [Diff(2, NoChange)]}
        tree2 = {"a": Diff(1, NoChange), "b": [Diff(2, UnknownChange)]}
        assert Diff.static_check_no_change(tree1)
        assert not Diff.static_check_no_change(tree2)

    def test_predicates(self):
        """incremental.py:242-266"""
        assert Diff.is_diff(Diff(1, NoChange)) and not Diff.is_diff(1)
        assert Diff.is_change_tangent(NoChange) and Diff.is_change_tangent(UnknownChange) and not Diff.is_change_tangent(0)


class TestArgdiffsType:
    def test_bare_arguments_are_refused(self, hostsim):
        """concepts.py:66-81: `Argdiffs` = a tree of Diffs; bare values are a type error in the reference (beartype), a
        TypeError here — never read as 'unchanged'"""
        @genjax.gen
        def model(mu):
            return genjax.normal(mu, 1.0) @ "x"

        tr = model.simulate(genjax.key(0), (0.5,))
        with pytest.raises(TypeError, match="Diff"):
            tr.update(genjax.key(1), genjax.ChoiceMap.empty(), (1.5,))
        with pytest.raises(TypeError, match="Diff"):
            model.update(genjax.key(1), tr, genjax.ChoiceMap.empty(), (1.5,))
        with pytest.raises(TypeError, match="Diff"):
            tr.edit(genjax.key(1), genjax.Update(genjax.ChoiceMap.empty()), (Diff.no_change(0.5), 2.0))
        new, w, rd, _ = tr.update(genjax.key(1), genjax.ChoiceMap.empty(), Diff.unknown_change((1.5,)))
        assert new.get_args() == (1.5,) or float(new.get_args()[0]) == 1.5
