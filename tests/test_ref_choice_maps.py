"""The reference's choice-map / selection tests, restated against this package's host-side trie
(/root/reference/tests/core/test_choice_maps.py; line ranges in the docstrings).  Host only: a ChoiceMap never touches
the device.

Not mirrored (SURVEY §2 out of scope, or the reference's internal representation): `ChoiceMap.switch` / `Switch`
(test_switch :369-404, test_switch_chm :530-550, test_or_with_switch :552-596, test_choicemap_switch :978-1020), the
`Static.mapping` layout and `attributes_dict` round trip (test_nested_static_choicemap :628-672, test_static_extend
:674-676, test_chm_roundtrip :872-874), `simplify` of filter nodes (test_simplify :678-706: this trie has no lazy
filter nodes to push down)."""
import numpy as np
import pytest
import torch
from hypothesis import assume, given, settings
from hypothesis import strategies as st

import genjax_amd as genjax
from genjax_amd import ChoiceMap, ChoiceMapNoValueAtAddress, Mask, Selection
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import SelectionBuilder as S
from genjax_amd import numpy as jnp


class TestSelections:
    def test_selection(self):
        """:38-52: a selected address selects everything below it, nothing above it"""
        sel = S["x"] | S["z", "y"]
        assert sel["x"] and sel["z", "y"] and sel["z", "y", "tail"]
        sel = S["x"]
        assert sel["x"] and sel["x", "y"] and sel["x", "y", "z"]
        sel = S["x", "y", "z"]
        assert sel["x", "y", "z"] and not sel["x"] and not sel["x", "y"]

    def test_wildcard(self):
        """:54-59: S[..., "y"] matches "y" under any first component"""
        sel = S["x"] | S[..., "y"]
        assert sel["x"] and sel["any_address", "y"] and sel["rando", "y", "tail"]

    def test_all_and_none(self):
        """:61-79"""
        a, n = Selection.all(), Selection.none()
        assert a == ~~a and a["x"] and a["y", "z"] and a[()]
        assert n == ~~n and not n["x"] and not n["y", "z"] and not n[()]
        assert Selection.none().extend("a", "b") == Selection.none()

    def test_builder_properties(self):
        """:81-104: S.all / S.none / S.leaf, S[()]"""
        assert S.all() == Selection.all() and S.all()["x"] and S.all()[()]
        assert S.none() == Selection.none() and not S.none()["x"]
        leaf = S.leaf().extend("a", "b")
        assert leaf["a", "b"] and not leaf["a"] and not leaf["a", "b", "c"]
        assert S[()] == Selection.leaf() and () in S[()]

    def test_leaf(self):
        """:106-117: exact matches only; no wildcards against a leaf"""
        leaf = Selection.leaf().extend("x", "y")
        assert not leaf["x"] and leaf["x", "y"] and not leaf["x", "y", "z"]
        with pytest.raises(TypeError):
            leaf[..., "y"]

    def test_complement(self):
        """:119-135"""
        sel = S["x"] | S["y"]
        comp = ~sel
        assert not comp["x"] and not comp["y"] and comp["z"]
        assert ~~sel == sel
        assert ~Selection.all() == Selection.none() and ~Selection.none() == Selection.all()

    def test_and(self):
        """:137-162"""
        s1, s2 = S["x"] | S["y"], S["y"] | S["z"]
        both = s1 & s2
        assert not both["x"] and both["y"] and not both["z"]
        assert not both.check() and both.get_subselection("y").check()
        a, n = Selection.all(), Selection.none()
        assert (a & s1) == s1 and (s1 & a) == s1 and (n & s1) == n and (s1 & n) == n
        assert s1 & s1 == s1 and s2 & s2 == s2

    def test_or(self):
        """:164-187"""
        s1, s2 = S["x"], S["y"]
        either = s1 | s2
        assert either["x"] and either["y"] and either.get_subselection("y").check() and not either["z"]
        a, n = Selection.all(), Selection.none()
        assert (a | s1) == a and (s1 | a) == a and (n | s1) == s1 and (s1 | n) == s1
        assert s1 | s1 == s1 and s2 | s2 == s2

    def test_filter(self):
        """:189-223: Selection.filter(chm)"""
        chm = ChoiceMap.kw(x=1, y=2, z=3)
        got = (S["x"] | S["y"]).filter(chm)
        assert "x" in got and "y" in got and "z" not in got and got["x"] == 1 and got["y"] == 2
        assert Selection.none().filter(chm).static_is_empty()
        assert Selection.all().filter(chm) == chm
        nested = ChoiceMap.kw(a={"b": 1, "c": 2}, d=3)
        got = (S["a", "b"] | S["d"]).filter(nested)
        assert "d" in got and "b" in got("a") and "c" not in got("a")

    def test_combination(self):
        """:225-232"""
        sel = ((S["x"] | S["y"]) & (S["y"] | S["z"])) | S["w"]
        assert not sel["x"] and sel["y"] and not sel["z"] and sel["w"]

    def test_contains(self):
        """:234-260: `in` is `[]`; check() is membership of ()"""
        sel = S["x"] | S["y", "z"]
        assert "x" in sel and sel["x"] and ("y", "z") in sel and sel["y", "z"]
        assert "y" not in sel and not sel["y"] and "w" not in sel and not sel["w"]
        nested = S["c"].extend("a", "b")
        assert ("a", "b", "c") in nested and nested["a", "b", "c"]
        assert ("a", "b") not in nested and not nested["a", "b"]
        assert not nested("a")("b").check() and nested("a")("b")("c").check()

    def test_ellipsis_only_in_front(self):
        """:262-267"""
        sel = S["a", "b", "c"] | S["x", "y", "z"]
        with pytest.raises(TypeError):
            sel["a", ..., ...]

    def test_static_sel(self):
        """:269-278"""
        xy = Selection.at["x", "y"]
        assert not xy[()] and xy["x", "y"] and not xy["other_address"]
        nested = Selection.at["x"].extend("y")
        assert nested["y", "x"] and not nested["y"]

    def test_chm_sel(self):
        """:280-298: the selection of a choice map's addresses"""
        sel = (C["x", "y"].set(3.0) | C["z"].set(5.0)).get_selection()
        assert sel["x", "y"] and sel["z"] and not sel["w"] and sel("x")["y"]
        assert ChoiceMap.empty().get_selection() == Selection.none()


class TestChoiceMapBuilder:
    def test_set(self):
        """:302-311: membership is true for the actual path only"""
        assert ChoiceMap.builder.set(1.0) == C[()].set(1.0)
        chm = C["a", "b"].set(1)
        assert chm["a", "b"] == 1 and ("a", "b") in chm and "a" not in chm and "b" in chm("a")

    def test_nested_set(self):
        """:313-317"""
        chm = C["x"].set(C["y"].set(2))
        assert chm["x", "y"] == 2 and ("x", "y") in chm and "y" not in chm

    def test_update(self):
        """:319-335: at[addr].update(fn) maps the sub-map (or the value) at addr"""
        chm = C["x", "y"].set(2)
        assert chm.at["x"].update(lambda m: C["z"].set(m))["x", "z", "y"] == 2
        assert chm.at["x", "y"].update(lambda v: v * v)["x", "y"] == 4
        assert chm.at["q"].update(lambda m: C["z"].set(m))(("q", "z")).static_is_empty()
        assert chm.at["q"].update(lambda m: C["z"].set(2))["q", "z"] == 2

    def test_empty(self):
        """:337-341"""
        assert C.n() == ChoiceMap.empty() and C["x", "y"].n() == ChoiceMap.empty()

    def test_v_matches_set(self):
        """:343-349"""
        assert C["a", "b"].set(1) == C["a", "b"].v(1)
        inner = C["y"].v(2)
        assert C["x"].v(inner)("x").get_value() == inner

    def test_from_mapping(self):
        """:351-362"""
        chm = C["base"].from_mapping([("a", 1.0), (("b", "c"), 2.0), (("b", "d", "e"), {"f": 3.0})])
        assert chm["base", "a"] == 1 and chm["base", "b", "c"] == 2 and chm["base", "b", "d", "e", "f"] == 3
        assert ("base", "a") in chm and ("base", "b", "c") in chm and ("b", "c") in chm("base")

    def test_d(self):
        """:364-373: dict values become nested maps"""
        chm = C["top"].d({"x": 3, "y": {"z": 4, "w": C["bottom"].d({"v": 5})}})
        assert chm["top", "x"] == 3 and chm["top", "y", "z"] == 4 and chm["top", "y", "w", "bottom", "v"] == 5

    def test_kw(self):
        """:375-381"""
        chm = C["root"].kw(a=1, b=C["nested"].kw(c=2, d={"deep": 3}))
        assert chm["root", "a"] == 1 and chm["root", "b", "nested", "c"] == 2
        assert chm["root", "b", "nested", "d", "deep"] == 3


class TestChoiceMap:
    def test_empty(self):
        """:408-410"""
        assert ChoiceMap.empty().static_is_empty()

    def test_choice(self):
        """:412-436: a value-only map; concrete masks resolve at once, an empty array is an empty map"""
        c = ChoiceMap.choice(42.0)
        assert c.get_value() == 42.0 and c.has_value() and () in c
        assert ChoiceMap.choice(Mask(42.0, False)).static_is_empty()
        assert ChoiceMap.choice(Mask(42.0, True)) == ChoiceMap.choice(42.0)
        mv = Mask(42.0, torch.tensor(False))
        assert ChoiceMap.choice(mv).get_value() is mv
        assert ChoiceMap.choice(torch.ones((0,))).static_is_empty()

    def test_kw_d_from_mapping(self):
        """:438-468"""
        chm = ChoiceMap.kw(x=1, y=2)
        assert chm["x"] == 1 and chm["y"] == 2 and "x" in chm and "other_value" not in chm
        chm = ChoiceMap.d({"a": 1, "b": {"c": 2, "d": {"e": 3}}})
        assert chm["a"] == 1 and chm["b", "c"] == 2 and chm["b", "d", "e"] == 3 and ("b", "d", "e") in chm
        chm = ChoiceMap.from_mapping([("x", 1), (("y", "z"), 2), (("w", "v", "u"), 3)])
        assert chm["x"] == 1 and chm["y", "z"] == 2 and chm["w", "v", "u"] == 3 and ("w", "v", "u") in chm

    def test_extend_through_at(self):
        """:470-507: at[...].set keeps everything else, overwrites in place, chains"""
        base = ChoiceMap.kw(x=1, y={"z": 2})
        ext = base.at["y", "w"].set(3)
        assert ext["x"] == 1 and ext["y", "z"] == 2 and ext["y", "w"] == 3
        multi = base.at["y", "w"].set(3).at["a", "b", "c"].set(4)
        assert multi["y", "w"] == 3 and multi["a", "b", "c"] == 4 and multi["y", "z"] == 2
        assert base.at["y", "z"].set(5)["y", "z"] == 5 and base["y", "z"] == 2
        nested = base.at["nested"].set(ChoiceMap.kw(a=6, b=7))
        assert nested["nested", "a"] == 6 and nested["nested", "b"] == 7 and nested["x"] == 1

    def test_filter_mask_extend(self):
        """:509-537"""
        chm = ChoiceMap.kw(x=1, y=2, z=3)
        got = (S["x"] | S["y"]).filter(chm)
        assert got["x"] == 1 and got["y"] == 2 and "z" not in got
        two = ChoiceMap.kw(x=1, y=2)
        assert two.mask(True) == two and two.mask(False).static_is_empty()
        ext = ChoiceMap.choice(1).extend("a", "b")
        assert ext["a", "b"] == 1 and ext.get_value() is None and ext.get_submap("a", "b").get_value() == 1
        assert ChoiceMap.empty().extend("a", "b").static_is_empty()

    def test_or_xor_access(self):
        """:598-626"""
        left, right = ChoiceMap.kw(x=1, y=2), ChoiceMap.kw(z=3, w=4)
        for m in (left | right, left ^ right):
            assert m["x"] == 1 and m["y"] == 2 and m["z"] == 3 and m["w"] == 4
            with pytest.raises(ChoiceMapNoValueAtAddress):
                m["does_not_exist"]

    def test_lookup_dynamic(self):
        """:708-716: integer addresses index a value-only array"""
        chm = ChoiceMap.choice(torch.tensor([2.3, 4.4, 3.3]))
        assert chm.get_submap("x").static_is_empty()
        assert float(chm[0]) == pytest.approx(2.3) and float(chm[1]) == pytest.approx(4.4) and float(chm[2]) == pytest.approx(3.3)
        assert ChoiceMap.empty().extend(slice(None, None, None)).static_is_empty()

    def test_access_dynamic(self):
        """:718-727: an array of indices in an address; missing indices are flagged False"""
        chm = C[jnp.array([4, 8, 2]), "x"].set(jnp.array([4.0, 8.0, 2.0]))
        for j in (2, 4, 8):
            assert chm[j, "x"] == Mask(float(j), True) or float(chm[j, "x"]) == float(j)
        assert (2, "x") in chm and (0, "x") not in chm and (11, "x") not in chm

    def test_merge_and_selection(self):
        """:729-749"""
        a, b = ChoiceMap.kw(x=1), ChoiceMap.kw(y=2)
        merged = a.merge(b)
        assert merged["x"] == 1 and merged["y"] == 2 and merged == a | b
        sel = ChoiceMap.kw(x=1, y=2).get_selection()
        assert sel["x"] and sel["y"] and not sel["z"]
        assert ChoiceMap.empty().static_is_empty() and not ChoiceMap.kw(x=1).static_is_empty()

    def test_xor(self):
        """:751-764"""
        a, b = ChoiceMap.kw(x=1), ChoiceMap.kw(y=2)
        x = a ^ b
        assert x["x"] == 1 and x["y"] == 2
        assert (ChoiceMap.empty() ^ ChoiceMap.empty()).static_is_empty()
        assert (a ^ ChoiceMap.empty()) == a and (ChoiceMap.empty() ^ a) == a

    def test_or(self):
        """:766-785: first operand wins; a value and a sub-map at one address do not merge"""
        a, b = ChoiceMap.kw(x=1), ChoiceMap.kw(y=2)
        o = a | b
        assert o.get_value() is None and o["x"] == 1 and o["y"] == 2
        assert (a | ChoiceMap.empty()) == a and (ChoiceMap.empty() | a) == a
        xm = ChoiceMap.choice(2.0).mask(torch.tensor(True))
        ym = ChoiceMap.choice(3.0).mask(torch.tensor(True))
        assert (xm | ym).get_value().unmask() == 2.0
        with pytest.raises(Exception, match="Choice and non-Choice in Or"):
            C["x"].set(1.0) | C["x", "y"].set(2.0)

    def test_and(self):
        """:787-816: common addresses, values from the right"""
        a, b = ChoiceMap.kw(x=1, y=2, z=3), ChoiceMap.kw(y=20, z=30, w=40)
        both = a & b
        assert "x" not in both and "w" not in both and both["y"] == 20 and both["z"] == 30
        assert (a & ChoiceMap.empty()).static_is_empty() and (ChoiceMap.empty() & a).static_is_empty()
        n1, n2 = ChoiceMap.kw(a={"b": 1, "c": 2}, d=3), ChoiceMap.kw(a={"b": 10, "d": 20}, d=30)
        n = n1 & n2
        assert n["a", "b"] == 10 and "c" not in n("a") and "d" not in n("a") and n["d"] == 30

    def test_call_getitem_contains(self):
        """:818-836"""
        chm = ChoiceMap.kw(x={"y": 1})
        assert chm("x")("y") == ChoiceMap.choice(1)
        assert "x" not in chm and "y" in chm("x") and ("x", "y") in chm and "z" not in chm
        one = ChoiceMap.kw(x=1)
        assert one["x"] == 1
        with pytest.raises(ChoiceMapNoValueAtAddress, match="y"):
            one["y"]

    def test_filter_with_wildcard(self):
        """:838-862: C[:].set({...}) is a plate of values; filtering keeps the plate structure"""
        xs, ys = torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0, 5.0, 6.0])
        got = C[:].set({"x": xs, "y": ys}).filter(S["x"])
        assert torch.equal(got[:, "x"], xs)
        with pytest.raises(ChoiceMapNoValueAtAddress):
            got[:, "y"]
        assert [float(got[j, "x"]) for j in range(3)] == [1.0, 2.0, 3.0]

    def test_static_idx(self):
        """:864-870: a concrete integer index gives plain values"""
        chm = C[0].set({"x": 1.0, "y": 2.0})
        assert chm[0, "x"] == 1.0 and chm[0, "y"] == 2.0

    def test_slices(self):
        """:1058-1088: no partial slices when setting; full slices, integer and 0-d array indices when reading"""
        for sl in (slice(None, 3), slice(0, 3), slice(0, 3, 1)):
            with pytest.raises(ValueError):
                C[sl, "x"].set(jnp.array([1, 2]))
        vals = torch.arange(10)
        chm = C[:, "x"].set(vals)
        assert torch.equal(chm[:, "x"], vals)
        assert int(chm[1, "x"]) == 1 and int(chm[torch.tensor(5), "x"]) == 5
        assert torch.equal(chm[0:4, "x"], vals[0:4])

    def test_address_validation(self):
        """:1094-1158 (the static parts): scalar / string keys, 0-d index arrays, one array of indices followed by
        full slices; partial slices and two index arrays are refused when setting"""
        chm = C[0, "x", 1].set(10)
        assert chm[0, "x", 1] == 10
        chm = C[torch.tensor(2, dtype=torch.int32), "y"].set(20)
        assert chm[2, "y"] == 20
        idx = torch.tensor([0, 1, 2])
        chm = C[0, "w", idx, :, :].set(torch.ones((3, 2, 2)))
        assert torch.equal(chm[0, "w", 1, :, :], torch.ones((2, 2)))
        assert float(chm[0, "w", 1, 0, 0]) == 1.0
        complex_chm = C[0, "a", idx, :, "b"].set(torch.ones((3, 2)))
        assert torch.equal(complex_chm[0, "a", 1, :, "b"], torch.ones(2))
        with pytest.raises(ValueError):
            C[0, "x", 1:3].set(jnp.array([1, 2]))
        with pytest.raises(ValueError):
            C[idx, idx].set(torch.ones((3, 3)))


@pytest.mark.usefixtures("hostsim")
class TestChoiceMapsAgainstModels:
    def test_filtered_constraint_updates_a_plate(self):
        """:864-894: a plate constraint filtered to one address updates only that address"""
        @genjax.gen
        def m():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = genjax.normal(10.0, 1.0) @ "y"
            return x, y
        key = genjax.key(0)
        tr = m.repeat(n=4).simulate(key, ())
        xs, ys = jnp.ones(4), 5 * jnp.ones(4)
        constraint = C[:].set({"x": xs, "y": ys})
        key, sub = genjax.split(key)
        ch = tr.update(sub, constraint.filter(S["x"]))[0].get_choices()
        assert np.array_equal(ch[:, "x"].numpy(), np.ones(4)) and not np.array_equal(ch[:, "y"].numpy(), 5 * np.ones(4))
        key, sub = genjax.split(key)
        ch = tr.update(sub, constraint.filter(S["y"]))[0].get_choices()
        assert not np.array_equal(ch[:, "x"].numpy(), np.ones(4)) and np.array_equal(ch[:, "y"].numpy(), 5 * np.ones(4))

    def test_invalid_subset(self):
        """:876-896: addresses of a choice map the model never visits"""
        @genjax.gen
        def model(x):
            y = genjax.normal(x, 1.0) @ "y"
            z = genjax.bernoulli(probs=0.5) @ "z"
            return y + z
        assert ChoiceMap.kw(y=1.0, z=1).invalid_subset(model, (0.0,)) is None
        bad = ChoiceMap.kw(x=1.0)
        assert bad.invalid_subset(model, (0.0,)) == bad
        assert ChoiceMap.kw(y=1.0, z=1, extra=0.5).invalid_subset(model, (0.0,)) == ChoiceMap.kw(extra=0.5)

    def test_invalid_subset_nested(self):
        """:898-934: a missing address is fine, an extra one is reported under its call's address"""
        @genjax.gen
        def inner():
            a = genjax.normal(0.0, 1.0) @ "a"
            b = genjax.bernoulli(probs=0.5) @ "b"
            return a + b

        @genjax.gen
        def outer():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = inner() @ "y"
            return x + y
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1)).invalid_subset(outer, ()) is None
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5)).invalid_subset(outer, ()) is None
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1, c=2.0)).invalid_subset(outer, ()) == ChoiceMap.kw(y=ChoiceMap.kw(c=2.0))
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1), z=3.0).invalid_subset(outer, ()) == ChoiceMap.kw(z=3.0)

    def test_invalid_subset_under_a_plate_and_a_scan(self):
        """:936-976, 1022-1056: the index layer is optional; an extra address under a plate is reported with it"""
        @genjax.gen
        def inner(x):
            a = genjax.normal(x, 1.0) @ "a"
            b = genjax.bernoulli(probs=0.5) @ "b"
            return a + b

        @genjax.gen
        def outer():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = inner.vmap(in_axes=(0,))(jnp.array([1.0, 2.0, 3.0])) @ "y"
            return x + jnp.sum(y)
        a_, b_ = jnp.array([0.5, 1.5, 2.5]), jnp.array([1, 0, 1])
        assert ChoiceMap.kw(x=1.0, y=C[:].set(ChoiceMap.kw(a=a_, b=b_))).invalid_subset(outer, ()) is None
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=a_, b=b_)).invalid_subset(outer, ()) is None
        extra = ChoiceMap.kw(x=1.0, y=C[:].set(ChoiceMap.kw(a=a_, b=b_, c=jnp.array([0.1, 0.2, 0.3]))))
        got = extra.invalid_subset(outer, ())
        assert got.addresses() == [("y", "c")] and np.allclose(np.asarray(got["y", :, "c"]), [0.1, 0.2, 0.3])

        @genjax.gen
        def step(mean):
            return genjax.normal(mean, 1.0) @ "x"
        it = step.iterate(n=4)
        xs = jnp.array([0.5, 1.2, 0.8, 0.9])
        assert C[:, "x"].set(xs).invalid_subset(it, (1.0,)) is None
        assert C["x"].set(xs).invalid_subset(it, (1.0,)) is None
        got = C[:].set({"x": xs, "z": xs}).invalid_subset(it, (1.0,))
        assert got.addresses() == [("z",)] and np.allclose(np.asarray(got[:, "z"]), np.asarray(xs))


# :1161-1202: property tests over random nested dictionaries
nested_dicts = st.deferred(lambda: st.dictionaries(
    st.text(), st.floats(allow_nan=False) | st.lists(st.floats(allow_nan=False)) | nested_dicts, min_size=1))


def all_paths(mapping):
    out, stack = [], [((), mapping)]
    while stack:
        prefix, m = stack.pop()
        if isinstance(m, dict) and m:
            for k, v in m.items():
                stack.append(((*prefix, k), v))
        else:
            out.append((prefix, m))
    return out


class TestSubmap:
    @settings(max_examples=60, deadline=None)
    @given(nested_dicts, st.data())
    def test_get_submap_split_path(self, mapping, data):
        chm = ChoiceMap.d(mapping)
        path, value = data.draw(st.sampled_from(all_paths(mapping)))
        assume(path)
        i = data.draw(st.integers(0, len(path)))
        assert chm.get_submap(path[:i])[path[i:]] == value
        assert chm.get_submap(path[:i], path[i:]) == chm.get_submap(path)

    @settings(max_examples=60, deadline=None)
    @given(nested_dicts, st.data())
    def test_path_can_be_splat(self, mapping, data):
        chm = ChoiceMap.d(mapping)
        path, _ = data.draw(st.sampled_from(all_paths(mapping)))
        assume(path)
        assert chm.get_submap(path) == chm.get_submap(*path)
