"""The reference's behavioural tests of the static language, restated against this package
(/root/reference/tests/generative_functions/test_static_gen_fn.py; the line ranges each test follows are given in
its docstring).  They are the reference-held fixtures for rows A5 / A6 / A17 of SURVEY.md §8: score == assess,
importance / update weight algebra, address checks, closures and kwargs, `StaticRequest` round trips, `inline`,
`partial_apply`, `@gen` methods of a Pytree.  Run here on the CPU mirror of the C-ABI; tests/test_reference_gpu.py
runs the same classes through libgenmi_hip.so.

Not mirrored (SURVEY §2 out of scope): `ChoiceMap.switch` (test_switch_chm_and_static :95-109,
test_assess_vmap_masked :111-137: the Switch machinery)."""
import numpy as np
import pytest
import torch

import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import Diff, Pytree, Regenerate, StaticRequest, Update
from genjax_amd import SelectionBuilder as S
from genjax_amd import numpy as jnp

pytestmark = pytest.mark.usefixtures("hostsim")


def f(x):
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


def lp(v, loc=0.0, scale=1.0):
    """log N(v; loc, scale) through the package's own assess"""
    return f(genjax.normal.assess(C.v(v), (loc, scale))[0])


def two_normals():
    @genjax.gen
    def simple_normal():
        y1 = genjax.normal(0.0, 1.0) @ "y1"
        y2 = genjax.normal(0.0, 1.0) @ "y2"
        return y1 + y2
    return simple_normal


class TestMetadata:
    def test_gen_keeps_the_function_metadata(self):
        """:40-83: __doc__, __name__, __module__, __qualname__, __wrapped__, __annotations__ of the wrapped function"""
        def original(x: float, y: float) -> float:
            """adds two numbers"""
            return x + y
        g = genjax.gen(original)
        assert g.__doc__ == original.__doc__ and g.__name__ == original.__name__
        assert g.__module__ == original.__module__ and g.__qualname__ == original.__qualname__
        assert getattr(g, "__wrapped__") == original
        assert g.__annotations__ == {"x": float, "y": float, "return": float}


class TestMisc:
    def test_static_sample_shape(self):
        """:87-93: sample_shape as a Const"""
        @genjax.gen
        def m():
            return genjax.normal(0.0, 1.0, sample_shape=genjax.Const((2, 2))) @ "normal"
        assert tuple(m.simulate(genjax.key(0), ()).get_retval().shape) == (2, 2)

    def test_literal_return_value_survives_update(self):
        """:139-151"""
        @genjax.gen
        def m():
            return 1
        k = genjax.key(0)
        tr = m.simulate(k, ())
        tr.update(k, C.n(), ())
        assert tr.get_retval() == 1

    def test_get_zero_trace(self):
        """:153-171: a trace of the right structure holding zeros"""
        @genjax.gen
        def model(x):
            y = genjax.normal(x, 1.0) @ "y"
            z = genjax.bernoulli(probs=0.7) @ "z"
            return y + z
        zt = model.get_zero_trace(0.0)
        assert isinstance(zt, genjax.Trace)
        assert zt.get_args() == (0.0,) and f(zt.get_retval()) == 0.0 and f(zt.get_score()) == 0.0
        ch = zt.get_choices()
        assert "y" in ch and "z" in ch and f(ch["y"]) == 0.0 and f(ch["z"]) == 0.0

    def test_get_zero_trace_nested(self):
        """:173-193"""
        @genjax.gen
        def nested_model():
            @genjax.gen
            def inner_model():
                return genjax.normal(0.0, 1.0) @ "inner"
            outer = genjax.normal(0.0, 1.0) @ "outer"
            return outer + (inner_model() @ "nested")
        zt = nested_model.get_zero_trace()
        assert zt.get_args() == () and f(zt.get_retval()) == 0.0 and f(zt.get_score()) == 0.0
        assert f(zt.get_choices()["outer"]) == 0.0 and f(zt.get_choices()["nested", "inner"]) == 0.0


class TestSimulate:
    def test_no_choices(self):
        """:197-206: a model without sites scores 0"""
        @genjax.gen
        def empty(x):
            return jnp.square(x - 3.0)
        _, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(empty.simulate)(sub, (jnp.ones(4),))
        assert f(tr.get_score()) == 0.0

    def test_score_is_the_sum_of_the_site_scores(self):
        """:208-223"""
        m = two_normals()
        key, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        ch = tr.get_choices()
        _, s1 = genjax.normal.importance(key, ch.get_submap("y1"), (0.0, 1.0))
        _, s2 = genjax.normal.importance(key, ch.get_submap("y2"), (0.0, 1.0))
        assert f(tr.get_score()) == pytest.approx(f(s1) + f(s2), rel=0.01)

    def test_multiple_returns(self):
        """:225-244"""
        @genjax.gen
        def m():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1, y2
        _, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        y1, y2 = tr.get_retval()
        assert f(y1) == f(tr.get_choices()["y1"]) and f(y2) == f(tr.get_choices()["y2"])
        assert f(tr.get_score()) == pytest.approx(lp(y1) + lp(y2), rel=0.01)

    def test_hierarchical_multiple_returns(self):
        """:246-270: the sub-model's sites live under its address"""
        @genjax.gen
        def sub():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1, y2

        @genjax.gen
        def m():
            y1, y2 = sub() @ "y1"
            return y1, y2
        _, k = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(k, ())
        y1, y2 = tr.get_retval()
        assert f(y1) == f(tr.get_choices()["y1", "y1"]) and f(y2) == f(tr.get_choices()["y1", "y2"])
        assert f(tr.get_score()) == pytest.approx(lp(y1) + lp(y2), rel=0.01)


class TestAssess:
    def test_no_choices(self):
        """:274-285"""
        @genjax.gen
        def empty(x):
            return jnp.square(x - 3.0)
        _, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(empty.simulate)(sub, (jnp.ones(4),))
        score, _ = genjax.jit(empty.assess)(tr.get_choices(), (jnp.ones(4),))
        assert f(score) == f(tr.get_score())

    def test_assess_of_simulated_choices_is_the_score(self):
        """:287-300 (and :402-414)"""
        m = two_normals()
        _, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        score, _ = genjax.jit(m.assess)(tr.get_choices(), ())
        assert f(score) == f(tr.get_score())

    def test_missing_address_names_the_address_and_the_literal(self):
        """:302-318: MissingAddress carries the address; assess({y1: 1, y2: -1}) == (-2.837877, 0.0)"""
        m = two_normals()
        with pytest.raises(genjax.MissingAddress) as exc:
            m.assess(C["y1"].set(1.0), ())
        assert exc.value.args == ("y2",)
        with pytest.raises(genjax.MissingAddress) as exc:
            m.assess(C["y2"].set(1.0), ())
        assert exc.value.args == ("y1",)
        score, ret = m.assess(C["y1"].set(1.0).at["y2"].set(-1.0), ())
        assert f(score) == pytest.approx(-2.837877, abs=5e-7) and f(ret) == 0.0


@Pytree.dataclass
class CustomTree(genjax.Pytree):
    x: object
    y: object


@genjax.gen
def tree_normal(custom_tree):
    y1 = genjax.normal(custom_tree.x, 1.0) @ "y1"
    y2 = genjax.normal(custom_tree.y, 1.0) @ "y2"
    return CustomTree(y1, y2)


class TestCustomPytree:
    def test_simulate_with_a_pytree_argument_and_return(self):
        """:356-369"""
        key = genjax.key(314159)
        tree = CustomTree(3.0, 5.0)
        tr = genjax.jit(tree_normal.simulate)(key, (tree,))
        ch = tr.get_choices()
        assert f(tr.get_score()) == pytest.approx(lp(ch["y1"], 3.0) + lp(ch["y2"], 5.0), rel=0.01)
        ret = tr.get_retval()
        assert isinstance(ret, CustomTree) and f(ret.x) == f(ch["y1"]) and f(ret.y) == f(ch["y2"])

    def test_importance_with_a_pytree_argument(self):
        """:383-398"""
        key = genjax.key(314159)
        tree = CustomTree(3.0, 5.0)
        tr, w = genjax.jit(tree_normal.importance)(key, C["y1"].set(5.0), (tree,))
        ch = tr.get_choices()
        assert f(tr.get_score()) == pytest.approx(lp(ch["y1"], 3.0) + lp(ch["y2"], 5.0), rel=0.01)
        assert f(w) == pytest.approx(lp(5.0, 3.0), rel=0.01)

    def test_a_user_distribution_with_a_pytree_argument(self):
        """:334-352, 371-381: a Distribution subclass whose sampler / density take a Pytree"""
        class _CustomNormal(genjax.Distribution):
            def estimate_logpdf(self, key, v, *args):
                (tree,) = args
                return genjax.normal.assess(C.v(v), (tree.x, tree.y))[0]

            def random_weighted(self, key, *args):
                (tree,) = args
                return genjax.normal.random_weighted(key, tree.x, tree.y)
        custom = _CustomNormal()

        @genjax.gen
        def m(tree):
            y = custom(tree) @ "y"
            return CustomTree(y, y)
        key = genjax.key(314159)
        tree = CustomTree(3.0, 5.0)
        tr = m.simulate(key, (tree,))
        assert f(tr.get_score()) == pytest.approx(lp(tr.get_choices()["y"], 3.0, 5.0), rel=0.01)


class TestImportance:
    def test_constrained_values_are_kept(self):
        """:418-439"""
        m = two_normals()
        _, sub = genjax.split(genjax.key(314159))
        choice = C["y1"].set(0.5).at["y2"].set(0.5)
        tr, _ = m.importance(sub, choice, ())
        assert f(tr.get_choices()["y1"]) == 0.5 and f(tr.get_choices()["y2"]) == 0.5
        assert f(tr.get_score()) == pytest.approx(2 * lp(0.5), rel=0.01)

    def test_weight_full_partial_and_no_constraints(self):
        """:441-489: w = score of the constrained sites"""
        m = two_normals()
        key = genjax.key(314159)
        tr, w = m.importance(key, C["y1"].set(0.5).at["y2"].set(0.5), ())
        assert f(tr.get_score()) == pytest.approx(2 * lp(0.5), rel=1e-4) and f(w) == pytest.approx(2 * lp(0.5), rel=1e-4)
        tr, w = m.importance(key, C["y2"].set(0.5), ())
        ch = tr.get_choices()
        assert f(ch["y2"]) == 0.5
        assert f(tr.get_score()) == pytest.approx(lp(ch["y1"]) + lp(0.5), rel=1e-4)
        assert f(w) == pytest.approx(lp(0.5), rel=1e-4)
        tr, w = m.importance(key, C.n(), ())
        ch = tr.get_choices()
        assert f(tr.get_score()) == pytest.approx(lp(ch["y1"]) + lp(ch["y2"]), rel=1e-4)
        assert f(w) == 0.0


def linked_models():
    """the six spellings of one model the reference holds to the same update algebra (:669-731)"""
    @genjax.gen
    def linked():
        y1 = genjax.normal(0.0, 1.0) @ "y1"
        y2 = genjax.normal(y1, 1.0) @ "y2"
        y3 = genjax.normal(y1 + y2, 1.0) @ "y3"
        return y1 + y2 + y3

    @genjax.gen
    def curried(v1, v2, v3):
        y1 = genjax.normal(0.0, v1) @ "y1"
        y2 = genjax.normal(y1, v2) @ "y2"
        y3 = genjax.normal(y1 + y2, v3) @ "y3"
        return y1 + y2 + y3

    @Pytree.dataclass
    class Model(Pytree):
        v1: object
        v2: object

        @genjax.gen
        def run(self, v3):
            y1 = genjax.normal(0.0, self.v1) @ "y1"
            y2 = genjax.normal(y1, self.v2) @ "y2"
            y3 = genjax.normal(y1 + y2, v3) @ "y3"
            return y1 + y2 + y3
    m = Model(jnp.asarray(1.0), jnp.asarray(1.0))

    @genjax.gen
    def m_linked(mm, v2, v3):
        y1 = genjax.normal(0.0, mm.v1) @ "y1"
        y2 = genjax.normal(y1, v2) @ "y2"
        y3 = genjax.normal(y1 + y2, v3) @ "y3"
        return y1 + y2 + y3

    @genjax.gen
    def m_created_internally(scale):
        return Model(scale, scale).run.inline(scale)
    return {"plain": linked,
            "curried": curried.partial_apply(1.0, 1.0, 1.0),
            "double-curried": curried.partial_apply(1.0).partial_apply(1.0, 1.0),
            "model method": m.run.partial_apply(1.0),
            "pytree argument curried": m_linked.partial_apply(m).partial_apply(1.0, 1.0),
            "model made inside, inlined": m_created_internally.partial_apply(jnp.asarray(1.0))}


class TestUpdate:
    def test_two_normals(self):
        """:502-550: score' = score + w; the discard holds the old value"""
        m = two_normals()
        key, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        upd = genjax.jit(m.update)
        key, sub = genjax.split(key)
        new_tr, w, _, discard = upd(sub, tr, C["y1"].set(2.0), ())
        ch = new_tr.get_choices()
        assert f(tr.get_choices()["y1"]) == f(discard["y1"])
        assert f(new_tr.get_score()) == pytest.approx(f(tr.get_score()) + f(w), abs=1e-6)
        assert f(new_tr.get_score()) == pytest.approx(lp(ch["y1"]) + lp(ch["y2"]), rel=0.01)
        key, sub = genjax.split(key)
        new_tr, w, _, _ = upd(sub, tr, C["y1"].set(2.0).at["y2"].set(3.0), ())
        assert f(new_tr.get_score()) == pytest.approx(f(tr.get_score()) + f(w), abs=1e-6)
        assert f(new_tr.get_score()) == pytest.approx(lp(2.0) + lp(3.0), rel=0.01)

    def test_linked_normals(self):
        """:552-581"""
        m = linked_models()["plain"]
        key, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        key, sub = genjax.split(key)
        new_tr, w, _, discard = genjax.jit(m.update)(sub, tr, C["y1"].set(2.0), ())
        ch = new_tr.get_choices()
        y1, y2, y3 = f(ch["y1"]), f(ch["y2"]), f(ch["y3"])
        assert f(tr.get_choices()["y1"]) == f(discard["y1"])
        assert f(new_tr.get_score()) == pytest.approx(f(tr.get_score()) + f(w), rel=0.01)
        assert f(new_tr.get_score()) == pytest.approx(lp(y1) + lp(y2, y1) + lp(y3, y1 + y2), rel=0.01)

    def test_hierarchical(self):
        """:583-621: untouched sub-model sites keep their values"""
        @genjax.gen
        def inner(x):
            return genjax.normal(x, 1.0) @ "y1"

        @genjax.gen
        def m():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = inner(y1) @ "y2"
            y3 = inner(y1 + y2) @ "y3"
            return y1 + y2 + y3
        key, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        old = tr.get_choices()
        key, sub = genjax.split(key)
        new_tr, w, _, discard = genjax.jit(m.update)(sub, tr, C["y1"].set(2.0), ())
        ch = new_tr.get_choices()
        y1, y2, y3 = f(ch["y1"]), f(ch["y2", "y1"]), f(ch["y3", "y1"])
        assert y1 == 2.0 and y2 == f(old["y2", "y1"]) and y3 == f(old["y3", "y1"])
        assert f(old["y1"]) == f(discard["y1"])
        assert f(new_tr.get_score()) == pytest.approx(f(tr.get_score()) + f(w), abs=2e-6)
        assert f(new_tr.get_score()) == pytest.approx(lp(y1) + lp(y2, y1) + lp(y3, y1 + y2), rel=0.01)

    @pytest.mark.parametrize("spelling", ["plain", "curried", "double-curried", "model method", "pytree argument curried",
                                          "model made inside, inlined"])
    def test_update_weight_correctness(self, spelling):
        """:623-731: w is the sum of the density changes of the sites the new value reaches; `edit(Update)` gives the
        same weight as `update`; updates compose — for every spelling of the model (partial_apply once / twice, a @gen
        method of a Pytree, a Pytree argument, a Pytree built inside the model and inlined)"""
        m = linked_models()[spelling]
        key, sub = genjax.split(genjax.key(314159))
        tr = genjax.jit(m.simulate)(sub, ())
        upd = genjax.jit(m.update)
        old = {a: f(tr.get_choices()[a]) for a in ("y1", "y2", "y3")}
        key, sub = genjax.split(key)
        new_tr, w, _, _ = upd(sub, tr, C["y1"].set(2.0), ())
        _, w_edit, _, _ = tr.edit(sub, Update(C["y1"].set(2.0)))
        assert f(w_edit) == f(w)
        assert f(new_tr.get_choices()["y1"]) == 2.0
        d3 = lp(old["y3"], 2.0 + old["y2"]) - lp(old["y3"], old["y1"] + old["y2"])
        d2 = lp(old["y2"], 2.0) - lp(old["y2"], old["y1"])
        d1 = lp(2.0) - lp(old["y1"])
        assert f(w) == pytest.approx(d3 + d2 + d1, rel=1e-4, abs=2e-6)
        key, sub = genjax.split(key)
        newer, w2, _, _ = upd(sub, new_tr, C["y3"].set(2.0), ())
        assert f(newer.get_choices()["y3"]) == 2.0
        assert f(w2) == pytest.approx(lp(2.0, 2.0 + old["y2"]) - lp(old["y3"], 2.0 + old["y2"]), rel=1e-4, abs=2e-6)

    def test_pytree_argument_with_argdiffs(self):
        """:733-769: Diff.no_change / Diff.unknown_change of a Pytree argument"""
        @Pytree.dataclass
        class SomePytree(genjax.Pytree):
            x: object
            y: object

        @genjax.gen
        def m(tree):
            return genjax.normal(tree.x, tree.y) @ "y1"
        key, sub = genjax.split(genjax.key(314159))
        tree = SomePytree(0.0, 1.0)
        tr = genjax.jit(m.simulate)(sub, (tree,))
        upd = genjax.jit(m.update)
        key, sub = genjax.split(key)
        new_tr, _, _, _ = upd(sub, tr, C["y1"].set(2.0), (Diff.no_change(tree),))
        assert f(new_tr.get_choices()["y1"]) == 2.0
        key, sub = genjax.split(key)
        new_tr, _, _, _ = upd(sub, tr, C["y1"].set(2.0), (Diff.unknown_change(SomePytree(1.0, 2.0)),))
        assert f(new_tr.get_choices()["y1"]) == 2.0


class TestAddressChecks:
    def test_duplicate_address(self):
        """:778-788: AddressReuse names the address"""
        @genjax.gen
        def dup():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y1"
            return y1 + y2
        with pytest.raises(genjax.AddressReuse) as exc:
            dup.simulate(genjax.key(314159), ())
        assert exc.value.args[0] == "y1"

    def test_a_traced_value_is_not_an_address(self):
        """:790-799"""
        @genjax.gen
        def bad():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ y1
            return y1 + y2
        with pytest.raises(TypeError):
            bad.simulate(genjax.key(314159), ())


class TestForwardRefAndClosures:
    def test_forward_reference(self):
        """:803-821: a callee defined after its caller"""
        def make():
            @genjax.gen
            def proposal(x):
                return outlier(x) @ "x"

            @genjax.gen
            def outlier(prob):
                return genjax.bernoulli(probs=prob) @ "is_outlier"
            return proposal
        tr = make().simulate(genjax.key(314159), (0.3,))
        assert f(tr.get_score()) == f(genjax.bernoulli.logpdf(tr.get_retval(), probs=0.3))

    def test_closure_is_a_generative_function(self):
        """:825-836 (GEN-420): gf(*args) simulates / importances with () as its arguments"""
        @genjax.gen
        def model():
            return genjax.normal(1.0, 0.001) @ "x"
        gfc = model()
        tr = gfc.simulate(genjax.key(0), ())
        assert f(tr.get_score()) == f(genjax.normal.logpdf(tr.get_retval(), 1.0, 0.001))
        tr_u, w = gfc.importance(genjax.key(1), C.kw(x=1.1), ())
        assert f(tr_u.get_score()) == f(genjax.normal.logpdf(tr_u.get_retval(), 1.0, 0.001))
        assert f(w) == f(tr_u.get_score())

    def test_closure_with_kwargs(self):
        """:838-886"""
        @genjax.gen
        def model(x, y, z=None):
            if z is None:
                raise ValueError("z must be provided")
            _ = genjax.normal(x + y, z) @ "sampled"
            return z
        key = genjax.key(0)
        with pytest.raises(ValueError, match="z must be provided"):
            model(1.0, 2.0)(key)
        gfc = model(1.0, 2.0, z=3.0)
        assert gfc(key) == 3.0                       # keyword arguments are passed through
        assert gfc(key, z=10.0) == 10.0              # and can be overridden at the call
        assert gfc.handle_kwargs()(key, z=5.0) == gfc(key, z=5.0)
        args = (1.0, 2.0, 3.0)
        assert f(gfc.simulate(key, ()).get_choices()["sampled"]) == f(model.simulate(key, args).get_choices()["sampled"])
        chm = C.kw(sampled=3.5)
        a, b = gfc.assess(chm, ()), model.assess(chm, args)
        assert f(a[0]) == f(b[0]) and a[1] == b[1]
        con = C.kw(sampled=3.0)
        assert f(gfc.importance(key, con, ())[1]) == f(model.generate(key, con, args)[1])
        assert gfc.handle_kwargs() == gfc.handle_kwargs().handle_kwargs()      # idempotent on a closure

    def test_handle_kwargs(self):
        """:963-984: a model taking ((args), {kwargs})"""
        @genjax.gen
        def model(x, y, z=None):
            if z is None:
                raise ValueError("z must be provided")
            _ = genjax.normal(x + y, z) @ "sampled"
            return z
        kwm = model.handle_kwargs()
        key = genjax.key(0)
        a = kwm.simulate(key, ((1.0,), {"y": 2.0, "z": 3.0}))
        b = model.simulate(key, (1.0, 2.0, 3.0))
        assert f(a.get_choices()["sampled"]) == f(b.get_choices()["sampled"])
        assert f(a.get_score()) == f(b.get_score()) and a.get_retval() == b.get_retval()
        assert a.get_args() == ((1.0,), {"y": 2.0, "z": 3.0}) and b.get_args() == (1.0, 2.0, 3.0)


class TestStaticEditRequest:
    def _round_trip(self, model, request, addr):
        key = genjax.key(0)
        tr = model.simulate(key, ())
        key, sub = genjax.split(key)
        new_tr, w, _, bwd = request.edit(key, tr, ())
        assert f(new_tr.get_choices()[addr]) == 3.0 and f(w) != 0.0
        old_tr, w_, _, _ = bwd.edit(sub, new_tr, ())
        assert f(old_tr.get_choices()[addr]) == f(tr.get_choices()[addr])
        assert f(w_) != 0.0 and f(w) + f(w_) == pytest.approx(0.0, abs=1e-6)

    def test_composition(self):
        """:890-910: Regenerate at one address, Update at another; the backward request undoes it, w + w' = 0"""
        self._round_trip(two_normals(), StaticRequest({"y1": Regenerate(S.all()), "y2": Update(C.v(3.0))}), "y2")

    def test_tuple_address(self):
        """:912-932"""
        @genjax.gen
        def m():
            y1 = genjax.normal(0.0, 1.0) @ ("y1", "y3")
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2
        self._round_trip(m, StaticRequest({("y1", "y3"): Regenerate(S.all()), "y2": Update(C.v(3.0))}), "y2")

    def test_hierarchical(self):
        """:934-959: a StaticRequest inside a StaticRequest"""
        @genjax.gen
        def sub():
            return genjax.normal(0.0, 1.0) @ "y2"

        @genjax.gen
        def m():
            y1 = genjax.normal(0.0, 1.0) @ ("y1", "y3")
            y2 = sub() @ "y2"
            return y1 + y2
        req = StaticRequest({("y1", "y3"): Regenerate(S.all()), "y2": StaticRequest({"y2": Update(C.v(3.0))})})
        self._round_trip(m, req, ("y2", "y2"))


def inline_models():
    base = two_normals()

    @genjax.gen
    def higher():
        return base.inline()

    @genjax.gen
    def higher_higher():
        return higher.inline()
    return higher, higher_higher


class TestInline:
    def test_simulate(self):
        """:988-1014: an inlined callee's addresses are the caller's"""
        for m in inline_models():
            ch = genjax.jit(m.simulate)(genjax.key(314159), ()).get_choices()
            assert "y1" in ch and "y2" in ch

    def test_importance(self):
        """:1016-1041"""
        for m in inline_models():
            tr, w = genjax.jit(m.importance)(genjax.key(314159), C["y1"].set(3.0), ())
            assert f(w) == lp(f(tr.get_choices()["y1"]))

    def test_update(self):
        """:1043-1082"""
        for m in inline_models():
            key, sub = genjax.split(genjax.key(314159))
            tr = genjax.jit(m.simulate)(sub, ())
            old = f(tr.get_choices()["y1"])
            new_tr, w, _, _ = genjax.jit(m.update)(key, tr, C["y1"].set(3.0), ())
            assert f(w) == pytest.approx(lp(3.0) - lp(old), rel=1e-4, abs=1e-6)

    def test_assess(self):
        """:1084-1114"""
        for m in inline_models():
            score, _ = genjax.jit(m.assess)(C["y1"].set(3.0).at["y2"].set(3.0), ())
            assert f(score) == pytest.approx(2 * lp(3.0), abs=1e-6)


class TestMethodsAndPartialApply:
    def test_gen_method_of_a_pytree(self):
        """:1116-1145: `self` is curried: absent from get_args(), present as partial_args"""
        @Pytree.dataclass
        class Model(Pytree):
            foo: object
            bar: object

            @genjax.gen
            def run(self, x):
                y = genjax.normal(self.foo, self.bar) @ "y"
                z = genjax.normal(x, 1.0) @ "z"
                return y + z
        m = Model(jnp.asarray(4.0), jnp.asarray(6.0))
        tr = m.run.simulate(genjax.key(0), (1.0,))
        ch = tr.get_choices()
        assert tr.get_args() == (1.0,)
        assert tr.get_gen_fn().partial_args[0] == m
        assert "y" in ch and "z" in ch and "q" not in ch

    def test_partial_apply(self):
        """:1147-1163"""
        @genjax.gen
        def model(x, y, z):
            return genjax.normal(x, y + z) @ "x"
        dc = model.partial_apply(1.0).partial_apply(1.0)
        tr = dc.simulate(genjax.key(0), (2.0,))
        assert tr.get_args() == (2.0,)
        assert tr.get_gen_fn().partial_args == (1.0, 1.0)
