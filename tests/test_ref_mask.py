"""The reference's `Mask` and `Pytree` tests, restated against this package
(/root/reference/tests/core/generative/test_functional_types.py, 13 tests; /root/reference/tests/core/test_pytree.py, 2;
line ranges in the docstrings).  Host only: a Mask is resolved while a site program is traced (static._leaf_call,
engine.Flat); what the device does with it is held by tests/parity.py::check_masked_constraints.

One difference, stated: the particle axis of a launch is implicit here, so `genjax.vmap(Mask.build)`
sees whole arrays and the shapes in the prefix error are the batched ones — the test matches the message's fixed part."""
import re

import numpy as np
import pytest
import torch

import genjax_amd as genjax
from genjax_amd import Mask
from genjax_amd import numpy as jnp


def same(a, b):
    return Mask.__eq__(a, b) is True


class TestMask:
    def test_mask_kwarg_constructor(self):
        """:28-37"""
        m = Mask(value=42, flag=True)
        assert m.value == 42 and m.flag is True
        m = Mask(value=42)
        assert m.value == 42 and m.flag is True

    def test_mask_unmask_without_default(self):
        """:39-46"""
        assert Mask(42, True).unmask() == 42
        with pytest.raises(Exception):
            Mask(42, False).unmask()
        with pytest.raises(Exception, match="some flag in a vectorized mask"):
            Mask(jnp.arange(3), jnp.array([True, False, True])).unmask()

    def test_mask_unmask_with_default(self):
        """:48-53"""
        assert Mask(42, True).unmask(default=0) == 42
        assert Mask(42, False).unmask(default=0) == 0
        got = Mask(jnp.array([1.0, 2.0]), jnp.array([True, False])).unmask(default=jnp.array([9.0, 9.0]))
        assert got.tolist() == [1.0, 9.0]

    def test_mask_unmask_pytree(self):
        """:55-63"""
        tree = {"a": 1, "b": [2, 3], "c": {"d": 4}}
        assert Mask(tree, True).unmask() == tree
        default = {"a": 0, "b": [0, 0], "c": {"d": 0}}
        assert Mask(tree, False).unmask(default=default) == default

    def test_build(self):
        """:65-103"""
        m = Mask.build(42, True)
        assert isinstance(m, Mask) and m.flag is True and m.value == 42
        nested = Mask.build(Mask.build(42, True), False)
        assert isinstance(nested, Mask) and nested.flag is False and nested.value == 42

        with pytest.raises(ValueError, match="must be a prefix of all leaf shapes"):
            genjax.vmap(Mask.build)(jnp.arange(2), jnp.array([[True], [False]], dtype=bool))

        v_mask = genjax.vmap(Mask.build)(jnp.arange(10), jnp.ones(10, dtype=bool))
        nested = Mask.build(v_mask, False)
        assert jnp.array_equal(nested.value, jnp.arange(10))
        assert jnp.array_equal(nested.primal_flag(), jnp.zeros(10, dtype=bool))
        assert same(nested, Mask.build(v_mask, jnp.array(False)))

        with pytest.raises(AssertionError,
                           match=re.escape("Can't build a Mask with non-matching Flag shapes (2,) and (10,)")):
            Mask.build(v_mask, jnp.array([False, True]))

    def test_scalar_flag_validation(self):
        """:105-137"""
        assert Mask.build(42, True).flag is True
        assert Mask.build([1, 2, 3], False).flag is False
        value = jnp.array([1.0, 2.0, 3.0])
        with pytest.raises(ValueError, match=re.escape("shape (1,) must be a prefix of all leaf shapes. Found (3,)")):
            Mask.build(value, jnp.array([True]))
        m = Mask.build(value, jnp.array(True))
        assert jnp.array_equal(m.primal_flag(), jnp.array(True))
        value = {"a": jnp.ones((3, 2)), "b": jnp.ones((3, 2))}
        flag = jnp.array(False)
        assert jnp.array_equal(Mask.build(value, flag).primal_flag(), flag)
        Mask.build({"a": jnp.ones((4, 8)), "b": jnp.ones((3, 2))}, jnp.array(True))

    def test_maybe_mask(self):
        """:139-153"""
        assert Mask.maybe_mask(42, True) == 42
        assert Mask.maybe_mask(42, False) is None
        m = Mask(42, True)
        assert Mask.maybe_mask(m, True) == 42
        assert Mask.maybe_mask(m, False) is None
        assert Mask.maybe_mask(None, jnp.asarray(True)) == Mask(None, jnp.asarray(True)), "None survives maybe_mask"

    def test_mask_or_concrete_flags(self):
        """:155-193"""
        r = Mask(42, True) | Mask(43, True)
        assert r.primal_flag() is True and r.value == 42
        r = Mask(42, True) | Mask(43, False)
        assert r.primal_flag() is True and r.value == 42
        r = Mask(42, False) | Mask(43, True)
        assert r.primal_flag() is True and r.value == 43
        assert (Mask(42, False) | Mask(43, False)).primal_flag() is False
        a = Mask(jnp.array([42, 42, 42, 42]), jnp.array([True, True, False, False]))
        b = Mask(jnp.array([43, 43, 43, 43]), jnp.array([False, True, False, True]))
        r = a | b
        assert r.primal_flag().tolist() == [True, True, False, True]
        assert [v for v, f in zip(r.value.tolist(), r.flag.tolist()) if f] == [42, 42, 43]

    def test_mask_xor_concrete_flags(self):
        """:195-233"""
        assert (Mask(42, True) ^ Mask(43, True)).primal_flag() is False
        r = Mask(42, True) ^ Mask(43, False)
        assert r.primal_flag() is True and r.value == 42
        r = Mask(42, False) ^ Mask(43, True)
        assert r.primal_flag() is True and r.value == 43
        assert (Mask(42, False) ^ Mask(43, False)).primal_flag() is False
        a = Mask(jnp.array([42, 42, 42, 42]), jnp.array([True, True, False, False]))
        b = Mask(jnp.array([43, 43, 43, 43]), jnp.array([False, True, False, True]))
        r = a ^ b
        assert r.primal_flag().tolist() == [True, False, False, True]
        assert r.value.tolist()[0] == 42 and r.value.tolist()[3] == 43

    def test_mask_combine_different_pytree_shapes(self):
        """:235-247"""
        a, b = Mask({"a": 1, "b": 2}, True), Mask({"a": 1}, True)
        with pytest.raises(ValueError, match="Cannot combine masks with different tree structures"):
            _ = a | b
        with pytest.raises(ValueError, match="Cannot combine masks with different tree structures"):
            _ = a ^ b

    def test_mask_combine_different_array_shapes(self):
        """:249-330"""
        msg = "Cannot combine masks with different array shapes"
        for a, b in [(Mask(jnp.ones((2, 3)), True), Mask(jnp.ones((2, 2)), True)),
                     (Mask(jnp.asarray(1.0), True), Mask(jnp.ones((2, 2)), True))]:
            with pytest.raises(ValueError, match=msg):
                _ = a | b
            with pytest.raises(ValueError, match=msg):
                _ = a ^ b
        m5, m6 = Mask(1.0, True), Mask(jnp.array(1.0), True)
        assert m5 | m6 == m6
        assert (m5 ^ m6).primal_flag() is False
        m7, m8 = Mask(1.0, True), Mask(2.0, False)
        assert m7 | m8 == m7
        assert m7 ^ m8 == m7
        m9 = Mask(jnp.array([1.0, 2.0]), jnp.array([True, False]))
        m10 = Mask(jnp.array([3.0, 4.0]), jnp.array([True, True]))
        assert same(m9 | m10, Mask(jnp.array([1.0, 4.0]), jnp.array([True, True])))
        assert same(m9 ^ m10, Mask(jnp.array([1.0, 4.0]), jnp.array([False, True])))
        m11 = Mask(jnp.array([[3.0, 4.0], [3.0, 4.0]]), jnp.array([True, True]))
        m12 = Mask(jnp.array([3.0, 4.0]), jnp.array(True))
        for other in (m11, m12):
            with pytest.raises(ValueError, match=msg):
                _ = m9 | other
            with pytest.raises(ValueError, match=msg):
                _ = m9 ^ other

    def test_mask_not(self):
        """:332-346"""
        assert ~Mask(1.0, True) == Mask(1.0, False)
        assert ~Mask(2.0, False) == Mask(2.0, True)
        m = Mask(jnp.array([1.0, 2.0]), jnp.array([True, False]))
        assert same(~m, Mask(jnp.array([1.0, 2.0]), jnp.array([False, True])))

    def test_mask_indexing(self):
        """:348-372"""
        m = Mask(jnp.array([[1, 2], [3, 4]]), True)
        assert m[0, 1].value == 2 and m[0, 1].primal_flag() is True
        v = Mask(jnp.array([[1, 2], [3, 4]]), jnp.array([True, False]))
        assert v[0, 1].value == 2 and bool(v[0, 1].primal_flag()) is True
        assert v[1, 0].value == 3 and bool(v[1, 0].primal_flag()) is False


class TestPytree:
    def test_unwrap(self):
        """test_pytree.py:8-11"""
        c = genjax.Pytree.const(5)
        assert c.unwrap() == 5
        assert genjax.Const.unwrap(10) == 10

    def test_pythonic(self):
        """test_pytree.py:15-58"""
        @genjax.Pytree.dataclass
        class Foo(genjax.PythonicPytree):
            x: torch.Tensor
            y: torch.Tensor

        x = jnp.array([1.0, 2.0, 3.0, 4.0, 5.0])
        y = jnp.array([10.0, 20.0, 30.0, 40.0, 50.0])
        f = Foo(x, y)
        assert f[1] == Foo(x[1], y[1])
        assert f[jnp.array(1, dtype=int)] == Foo(x[1], y[1])
        for sl in (slice(None, 2), slice(2, None), slice(1, None, 4)):
            assert jnp.all(f[sl].x == x[sl]) and jnp.all(f[sl].y == y[sl])
        assert len(f) == x.shape[0]
        it = iter(f)
        assert next(it) == Foo(x[0], y[0])
        assert next(it) == Foo(x[1], y[1])
        ff = f + f
        assert len(ff) == 2 * len(f)
        assert jnp.allclose(ff.x, jnp.concatenate((x, x)))
        p = Foo(jnp.array(-1.0), jnp.array(-10.0))
        fp = f.prepend(p)
        assert len(fp) == 1 + len(f)
        assert jnp.all(fp[0].x == p.x) and jnp.all(fp[0].y == p.y)
        assert fp[0] == p
