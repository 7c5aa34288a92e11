"""The reference's behavioural tests of its combinators and trace accessors, restated against this package:

  /root/reference/tests/generative_functions/test_scan_combinator.py   (27 tests; 19 of them the scan SUGAR with exact
                                                                        integer expectations: free golden values)
  /root/reference/tests/generative_functions/test_vmap_combinator.py   (13)
  /root/reference/tests/generative_functions/test_repeat_combinator.py (3)
  /root/reference/tests/core/generative/test_core.py                   (10)

Run here on the CPU mirror of the C-ABI; tests/test_reference_gpu.py runs the same classes through libgenmi_hip.so.

Not mirrored (SURVEY §2 out of scope, or a JAX implementation detail):
  test_core.py: test_get_subtrace_switch, test_or_else (the Switch / or_else combinators); test_tupled_address_conflict
    (skipped in the reference itself);
  test_vmap_combinator.py: the dtype-promotion message of test_vmap_validation (a jax.vmap error text); the `mask()`
    half of test_vmap_combinator_vmap_pytree is in tests/test_ref_mask_combinator.py;
  test_scan_combinator.py: the last assertion block of TestScanIndexRequest (an out-of-range dynamic index is CLAMPED by
    XLA's dynamic_slice and the test only passes because the assert inside `pytest.raises(AssertionError)` trips)."""
import numpy as np
import pytest
import torch

import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import Diff, IndexRequest, Regenerate, Selection, StaticRequest, Update
from genjax_amd import SelectionBuilder as S
from genjax_amd import numpy as jnp

pytestmark = pytest.mark.usefixtures("hostsim")


def f(x):
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


def arr(x):
    """a result as a numpy array (tuples of tensors stack, as jnp.asarray does)"""
    if isinstance(x, (tuple, list)):
        return np.stack([arr(v) for v in x])
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def lp(v, loc=0.0, scale=1.0):
    return f(genjax.normal.assess(C.v(v), (loc, scale))[0])


KEY = 314159


# ---------------------------------------------------------------------------
# test_scan_combinator.py
# ---------------------------------------------------------------------------
def make_scanner():
    @genjax.iterate(n=10)
    @genjax.gen
    def scanner(x):
        return genjax.normal(x, 1.0) @ "z"
    return scanner


def make_scanner_n(n):
    @genjax.iterate(n=n)
    @genjax.gen
    def scanner(x):
        return genjax.normal(x, 1.0) @ "z"
    return scanner


class TestIterateSimpleNormal:
    def test_project_all_is_the_score(self):
        """:42-53"""
        scanner = make_scanner()
        key, sub = genjax.split(genjax.key(KEY))
        tr = genjax.jit(scanner.simulate)(sub, (0.01,))
        assert f(tr.project(key, Selection.all())) == f(tr.get_score())

    def test_importance_at_one_step(self):
        """:55-62: C[i, "z"] constrains step i; the weight is that step's density given the step before"""
        scanner = make_scanner()
        _, sub = genjax.split(genjax.key(KEY))
        for i in range(1, 5):
            tr, w = genjax.jit(scanner.importance)(sub, C[i, "z"].set(0.5), (0.01,))
            ch = tr.get_choices()
            assert f(ch[i, "z"]) == 0.5
            assert f(w) == lp(0.5, f(ch[i - 1, "z"]))

    def test_update_at_one_step(self):
        """:64-81"""
        scanner = make_scanner()
        _, sub = genjax.split(genjax.key(KEY))
        for i in range(1, 5):
            tr, _ = genjax.jit(scanner.importance)(sub, C[i, "z"].set(0.5), (0.01,))
            new_tr, _, _, _ = genjax.jit(scanner.update)(sub, tr, C[i, "z"].set(1.0), Diff.no_change((0.01,)))
            assert f(new_tr.get_choices()[i, "z"]) == 1.0


    def test_assess_round_trip_of_the_scan_sugar(self):
        """simulate, then assess(choices) is the trace's score and return value, for all four sugar combinators
        (the adapters' assess read the step axis of the choices as a particle batch)"""
        @genjax.gen
        def walk(x):
            return genjax.normal(x, 1.0) @ "z"

        @genjax.gen
        def drift(c, a):
            return genjax.normal(c + a, 1.0) @ "z"
        xs = jnp.array([0.5, -0.25, 1.0])
        cases = [(walk.iterate(n=3), (0.0,)), (walk.iterate_final(n=3), (0.0,)),
                 (drift.accumulate(), (0.0, xs)), (drift.reduce(), (0.0, xs))]
        for gf, args in cases:
            tr = gf.simulate(genjax.key(KEY), args)
            score, ret = gf.assess(tr.get_choices(), args)
            assert tuple(getattr(score, "shape", ())) == (), repr(gf)
            assert f(score) == pytest.approx(f(tr.get_score()), rel=1e-6), repr(gf)
            assert np.array_equal(arr(ret), arr(tr.get_retval())), repr(gf)
            # and it is the chain of step densities, from scipy-checked normal log-densities
            z = arr(tr.get_choices()[..., "z"]).reshape(-1)
            prev = 0.0
            want = 0.0
            for t in range(3):
                want += lp(z[t], prev + (f(xs[t]) if len(args) == 2 else 0.0))
                prev = float(z[t])
            assert f(score) == pytest.approx(want, rel=1e-5), repr(gf)


@genjax.gen
def inc(prev):
    return prev + 1


@genjax.gen
def inc_tupled(arg):
    prev, offset = arg
    return (prev + offset, offset)


class TestIterate:
    """:97-198: exact integer expectations"""
    key = genjax.key(KEY)

    def test_inc(self):
        assert int(inc.simulate(self.key, (0,)).get_retval()) == 1

    def test_iterate_includes_the_initial_value(self):
        r = inc.iterate(n=4).simulate(self.key, (0,)).get_retval()
        assert np.array_equal(arr(r), [0, 1, 2, 3, 4])
        wrapped = inc.iterate(n=4).simulate(self.key, (jnp.array(0),)).get_retval()
        assert np.array_equal(arr(r), arr(wrapped))

    def test_iterate_final(self):
        assert int(inc.iterate_final(n=10).simulate(self.key, (0,)).get_retval()) == 10

    def test_iterate_and_accumulate_past_the_unroll_limit(self):
        """scan.py:916-977, :1050-1103 for any n: past Scan.unroll_max steps the scan is a counted loop and the stack
        [init, f(init), ...] is assembled from the loop's stored rows; same values as the unrolled form on the shared
        prefix (the step keys chain: fold_in(key, t)), one particle and a batch of keys"""
        from genjax_amd import combinators as cmb
        n_long, n_short = 200, cmb.SCAN_UNROLL_MAX
        assert np.array_equal(arr(inc.iterate(n=n_long).simulate(self.key, (0,)).get_retval()), np.arange(n_long + 1))
        r = add.accumulate().simulate(self.key, (0, jnp.ones(n_long))).get_retval()
        assert np.array_equal(arr(r), np.arange(n_long + 1))
        scanner_long, scanner_short = make_scanner_n(n_long), make_scanner_n(n_short)
        for key in (self.key, genjax.split(self.key, 7)):
            tl, ts = scanner_long.simulate(key, (0.25,)), scanner_short.simulate(key, (0.25,))
            rl, rs = arr(tl.get_retval()), arr(ts.get_retval())
            assert rl.shape[-1] == n_long + 1 and np.all(rl[..., 0] == np.float32(0.25))
            assert np.array_equal(rl[..., : n_short + 1], rs)
            assert np.array_equal(rl[..., 1:], arr(tl.get_choices()["z"]))

    def test_inc_tupled(self):
        assert np.array_equal(arr(inc_tupled.simulate(self.key, ((0, 2),)).get_retval()), [2, 2])

    def test_iterate_tupled(self):
        r = inc_tupled.iterate(n=4).simulate(self.key, ((0, 2),)).get_retval()
        assert np.array_equal(arr(r), [[0, 2, 4, 6, 8], [2, 2, 2, 2, 2]])

    def test_iterate_final_tupled(self):
        r = inc_tupled.iterate_final(n=10).simulate(self.key, ((0, 2),)).get_retval()
        assert np.array_equal(arr(r), [20, 2])

    def test_iterate_array(self):
        @genjax.gen
        def double(prev):
            return prev + prev
        r = double.iterate(n=4).simulate(self.key, (jnp.ones(4),)).get_retval()
        assert np.array_equal(arr(r), np.outer([1, 2, 4, 8, 16], np.ones(4)))

    def test_iterate_matrix(self):
        fib = jnp.array([[1, 1], [1, 0]])

        @genjax.gen
        def step(prev):
            return fib @ prev
        r = step.iterate(n=5).simulate(self.key, (fib,)).get_retval()
        want = [[[1, 1], [1, 0]], [[2, 1], [1, 1]], [[3, 2], [2, 1]], [[5, 3], [3, 2]], [[8, 5], [5, 3]], [[13, 8], [8, 5]]]
        assert np.array_equal(arr(r), want)


@genjax.gen
def add(carry, x):
    return carry + x


@genjax.gen
def add_tupled(acc, x):
    carry, offset = acc
    return (carry + x + offset, offset)


class TestAccumulateReduce:
    """:214-318"""
    key = genjax.key(KEY)

    def test_add(self):
        assert f(add.simulate(self.key, (0, 2)).get_retval()) == 2

    def test_accumulate(self):
        r = add.accumulate().simulate(self.key, (0, jnp.ones(4))).get_retval()
        assert np.array_equal(arr(r), [0, 1, 2, 3, 4])
        wrapped = add.accumulate().simulate(self.key, (jnp.array(0), jnp.ones(4))).get_retval()
        assert np.array_equal(arr(r), arr(wrapped))

    def test_reduce(self):
        assert f(add.reduce().simulate(self.key, (0, jnp.ones(10))).get_retval()) == 10

    def test_add_tupled(self):
        assert np.array_equal(arr(add_tupled.simulate(self.key, ((0, 2), 10)).get_retval()), [12, 2])

    def test_accumulate_tupled(self):
        r = add_tupled.accumulate().simulate(self.key, ((0, 2), jnp.ones(4))).get_retval()
        assert np.array_equal(arr(r), [[0, 3, 6, 9, 12], [2, 2, 2, 2, 2]])

    def test_reduce_tupled(self):
        r = add_tupled.reduce().simulate(self.key, ((0, 2), jnp.ones(10))).get_retval()
        assert np.array_equal(arr(r), [30, 2])

    def test_accumulate_array(self):
        r = add.accumulate().simulate(self.key, (jnp.ones(4), jnp.eye(4))).get_retval()
        want = [[1, 1, 1, 1], [2, 1, 1, 1], [2, 2, 1, 1], [2, 2, 2, 1], [2, 2, 2, 2]]
        assert np.array_equal(arr(r), want)

    def test_accumulate_matrix(self):
        fib = jnp.array([[1, 1], [1, 0]])

        @genjax.gen
        def matmul(prev, nxt):
            return prev @ nxt
        r = matmul.accumulate().simulate(self.key, (fib, jnp.broadcast_to(fib, (5, 2, 2)))).get_retval()
        want = [[[1, 1], [1, 0]], [[2, 1], [1, 1]], [[3, 2], [2, 1]], [[5, 3], [3, 2]], [[8, 5], [5, 3]], [[13, 8], [8, 5]]]
        assert np.array_equal(arr(r), want)


class TestScanBehaviour:
    key = genjax.key(KEY)

    def test_update_of_one_step_inside_a_model(self):
        """:326-346: a Pytree of scanned inputs; updating step 1 re-scores step 2 against the new carry"""
        @genjax.Pytree.dataclass
        class A(genjax.Pytree):
            x: object

        @genjax.gen
        def step(b, a):
            return genjax.normal(b + a.x, 1e-6) @ "b", None

        @genjax.gen
        def model(k):
            return step.scan(n=3)(k, A(jnp.array([1.0, 2.0, 3.0]))) @ "steps"
        k1, k2 = genjax.split(self.key)
        tr = model.simulate(k1, (jnp.array(1.0),))
        u, w, _, _ = tr.update(k2, C["steps", 1, "b"].set(99.0))
        assert np.allclose(arr(u.get_choices()["steps", :, "b"]), [2.0, 99.0, 7.0], atol=0.1)
        assert f(w) < -100.0

    def test_scan_with_parameters(self):
        """:354-384: a partially applied kernel closing over a dict of parameters"""
        @genjax.gen
        def step(data, state, update):
            new_state = state + genjax.normal(update, data["noise"]) @ "state"
            return new_state, new_state

        @genjax.gen
        def model(data):
            return step.partial_apply(data).scan(n=3)(data["initial"], data["updates"]) @ "s"
        tr = model.simulate(self.key, ({"initial": jnp.array(3.0), "updates": jnp.array([5.0, 6.0, 7.0]), "noise": 1e-6},))
        end, steps = tr.get_retval()
        assert np.allclose(arr(steps), [8.0, 14.0, 21.0], atol=0.1) and np.allclose(arr(end), 21.0, atol=0.1)

    def test_length_inferred_from_the_scanned_input(self):
        """:386-406"""
        @genjax.gen
        def walk_step(x, std):
            new_x = genjax.normal(x, std) @ "x"
            return new_x, new_x
        args = (0.0, jnp.array([2.0, 4.0, 3.0, 5.0, 1.0]))
        tr = walk_step.scan(n=5).simulate(self.key, args)
        _, expected = tr.get_retval()
        assert np.allclose(arr(tr.get_choices()[:, "x"]), arr(expected))
        for sim in (walk_step.scan().simulate, genjax.jit(walk_step.scan().simulate)):
            assert np.allclose(arr(sim(self.key, args).get_choices()[:, "x"]), arr(expected))

    def test_zero_length_scan(self):
        """:408-427 (GEN-333)"""
        @genjax.gen
        def step(state, sigma):
            new_x = genjax.normal(state, sigma) @ "x"
            return (new_x, new_x + 1)
        tr = step.scan(n=0).simulate(self.key, (2.0, jnp.arange(0, dtype=float)))
        assert tr.get_choices().static_is_empty()
        _, sub = genjax.split(self.key)
        step.scan().importance(sub, tr.get_choices(), (2.0, 2.0 + jnp.arange(0, dtype=float)))

    def test_validation_of_scanned_lengths(self):
        """:429-447"""
        @genjax.gen
        def foo(shift, d):
            x = genjax.normal(d["loc"], d["scale"]) @ "x"
            return x + shift, None
        d = {"loc": jnp.array([10.0, 12.0]), "scale": jnp.array([1.0])}
        with pytest.raises(ValueError, match="different leading axis sizes: 2, 1"):
            genjax.jit(foo.scan().simulate)(self.key, (jnp.array([1.0]), d))

    def test_vmap_over_keys_of_a_scan(self):
        """:449-469: the scan aggregates a score per key, its choices gain the step axis"""
        @genjax.gen
        def model(x, _):
            y = genjax.normal(x, 1.0) @ "y"
            return y, None
        sc = model.scan()
        keys = genjax.split(self.key, 10)
        args = (jnp.array(1.0), jnp.arange(5, dtype=float))
        res = genjax.vmap(lambda k: sc.simulate(k, args))(keys)
        assert tuple(res.get_score().shape) == (10,)
        assert tuple(res.get_choices()[:, "y"].shape) == (10, 5)


def scanned_normal_model():
    @genjax.gen
    def scanned_normal():
        @genjax.gen
        def kernel(carry, _):
            z = genjax.normal(0.0, 1.0) @ "z"
            return z, None
        y1 = genjax.normal(0.0, 1.0) @ "y1"
        _ = genjax.normal(0.0, 1.0) @ "y2"
        return kernel.scan(n=10)(y1, None) @ "kernel"
    return scanned_normal


class TestScanRegenerateAndIndexRequest:
    def test_regenerate_a_site_outside_the_scan(self):
        """:477-498: w = new density - old density of the regenerated site"""
        m = scanned_normal_model()
        key, sub = genjax.split(genjax.key(KEY))
        tr = m.simulate(sub, ())
        old = f(tr.get_choices()["y1"])
        new_tr, w, _, _ = Regenerate(S["y1"]).edit(key, tr, ())
        assert f(w) == pytest.approx(lp(f(new_tr.get_choices()["y1"])) - lp(old), abs=1e-6)

    def test_index_request_regenerates_one_step(self):
        """:506-531: StaticRequest({"kernel": IndexRequest(idx, Regenerate(z))}) for every step"""
        m = scanned_normal_model()
        key, sub = genjax.split(genjax.key(KEY))
        tr = m.simulate(sub, ())
        for idx in range(10):
            old = f(tr.get_choices()["kernel", idx, "z"])
            req = StaticRequest({"kernel": IndexRequest(jnp.array(idx), Regenerate(S["z"]))})
            new_tr, w, _, _ = req.edit(key, tr, ())
            new = f(new_tr.get_choices()["kernel", idx, "z"])
            assert new != old
            assert f(w) == pytest.approx(lp(new) - lp(old), abs=1e-6)


# ---------------------------------------------------------------------------
# test_vmap_combinator.py
# ---------------------------------------------------------------------------
def vmapped_normal():
    @genjax.vmap(in_axes=(0,))
    @genjax.gen
    def model(x):
        return genjax.normal(x, 1.0) @ "z"
    return model


class TestVmap:
    key = genjax.key(KEY)

    def test_plate_score_is_the_sum_of_the_inner_scores(self):
        """:28-39: a 50-element plate (a counted loop here)"""
        tr = genjax.jit(vmapped_normal().simulate)(self.key, (jnp.arange(0, 50, dtype=float),))
        assert f(tr.get_score()) == pytest.approx(float(arr(tr.inner.get_score()).astype(np.float64).sum()), rel=1e-5)

    def test_project(self):
        """:41-56"""
        tr = genjax.jit(vmapped_normal().simulate)(self.key, (jnp.arange(0, 10, dtype=float),))
        assert f(tr.project(self.key, Selection.all())) == f(tr.get_score())
        assert f(tr.project(self.key, Selection.none())) == 0.0

    def test_vector_choice_map_importance(self):
        """:58-76: a choice map holding one value per element"""
        chm = genjax.vmap(lambda idx, v: C[idx, "z"].set(v))(jnp.arange(3), jnp.array([3.0, 2.0, 3.0]))
        _, w = genjax.jit(vmapped_normal().importance)(self.key, chm, (jnp.arange(0, 3, dtype=float),))
        assert f(w) == pytest.approx(lp(3.0, 0.0) + lp(2.0, 1.0) + lp(3.0, 2.0), abs=2e-6)

    def test_indexed_choice_map_importance(self):
        """:78-98"""
        kernel = vmapped_normal()
        map_over = jnp.arange(0, 3, dtype=float)
        key, sub = genjax.split(self.key)
        _, w = genjax.jit(kernel.importance)(sub, C[0, "z"].set(3.0), (map_over,))
        assert f(w) == lp(3.0, 0.0)
        key, sub = genjax.split(key)
        zv = np.array([3.0, -1.0, 2.0], np.float32)
        chm = genjax.vmap(lambda idx, v: C[idx, "z"].set(v))(jnp.arange(3), jnp.array(zv))
        tr, _ = kernel.importance(sub, chm, (map_over,))
        for i in range(3):
            assert f(tr.get_choices()[i, "z"]) == zv[i]

    def test_nested_indexed_choice_map_importance(self):
        """:100-117: C[0, "outer", 1, "z"] reaches one element of a plate inside a plate"""
        model = vmapped_normal()

        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def higher_model(x):
            return model(x) @ "outer"
        _, w = genjax.jit(higher_model.importance)(self.key, C[0, "outer", 1, "z"].set(1.0), (jnp.ones((3, 3), dtype=float),))
        assert f(w) == lp(1.0, 1.0)

    def test_in_axes_as_a_tree_prefix(self):
        """:139-146: in_axes=(None, (0, None)) maps one leaf of a nested argument"""
        @genjax.vmap(in_axes=(None, (0, None)))
        @genjax.gen
        def foo(y, args):
            loc, (scale, _) = args
            x = genjax.normal(loc, scale) @ "x"
            return x + y
        tr = genjax.jit(foo.simulate)(self.key, (10.0, (jnp.arange(3.0), (1.0, jnp.arange(3)))))
        assert tuple(tr.get_retval().shape) == (3,)

    def test_assess_of_simulated_choices(self):
        """:148-161"""
        model = vmapped_normal()
        map_over = jnp.arange(0, 50, dtype=float)
        tr = genjax.jit(model.simulate)(self.key, (map_over,))
        assert f(model.assess(tr.get_choices(), (map_over,))[0]) == f(tr.get_score())

    def test_validation(self):
        """:163-199: a scalar cannot be mapped; in_axes must be a prefix of the arguments; mapped lengths must agree"""
        @genjax.gen
        def foo(loc, scale):
            return genjax.normal(loc, scale) @ "x"
        with pytest.raises(ValueError, match="rank should be at least 1, but is only 0"):
            genjax.jit(foo.vmap(in_axes=(0, None)).simulate)(self.key, (10.0, jnp.arange(3.0)))
        with pytest.raises(ValueError, match="in_axes specification must be a tree prefix"):
            genjax.jit(foo.vmap(in_axes=(0, (0, None))).simulate)(self.key, (10.0, jnp.arange(3.0)))
        with pytest.raises((IndexError, ValueError)):
            genjax.jit(foo.vmap(in_axes=0).simulate)(self.key, (jnp.arange(2), jnp.arange(3)))

    def test_vmap_over_keys_of_a_plate(self):
        """:201-221"""
        @genjax.gen
        def model(x):
            return genjax.normal(x, 1.0) @ "y"
        vm = model.vmap(in_axes=(0,))
        keys = genjax.split(self.key, 10)
        xs = jnp.arange(5, dtype=float)
        res = genjax.vmap(lambda k: vm.simulate(k, (xs,)))(keys)
        assert tuple(res.get_score().shape) == (10,)
        assert tuple(res.get_choices()[:, "y"].shape) == (10, 5)

    def test_zero_length_plate(self):
        """:223-235"""
        @genjax.gen
        def step(state, sigma):
            new_x = genjax.normal(state, sigma) @ "x"
            return (new_x, new_x + 1)
        tr = step.vmap(in_axes=(None, 0)).simulate(genjax.key(20), (2.0, jnp.arange(0, dtype=float)))
        assert tr.get_choices().static_is_empty()

    def test_plate_over_a_batched_pytree(self):
        """:243-265: a generative function mapped over the leading axis of a Pytree's leaves"""
        @genjax.Pytree.dataclass
        class MyClass(genjax.PythonicPytree):
            x: object

        @genjax.gen
        def gf(mc):
            return mc.x + 5
        batched = MyClass(jnp.arange(5))
        assert np.array_equal(arr(gf.vmap(in_axes=0)(batched)(genjax.key(0))), np.arange(5) + 5)


def bare_plate_model():
    @genjax.gen
    def model():
        x = genjax.normal(0.0, 1.0) @ "x"
        _ = genjax.normal.vmap()(jnp.zeros(1000), jnp.ones(1000)) @ "a"
        return x
    return model


class TestVmapIndexRequest:
    """:268-325: a 1000-element plate of a bare distribution inside a model, edited one element at a time"""

    def test_regenerate_one_element(self):
        m = bare_plate_model()
        key, sub = genjax.split(genjax.key(KEY))
        tr = m.simulate(sub, ())
        for idx in range(10):
            old = f(tr.get_choices()["a", idx])
            req = StaticRequest({"a": IndexRequest(jnp.array(idx), Regenerate(S.all()))})
            new_tr, w, _, _ = req.edit(key, tr, ())
            new = f(new_tr.get_choices()["a", idx])
            assert f(w) == pytest.approx(lp(new) - lp(old), abs=1e-6)

    def test_update_one_element(self):
        m = bare_plate_model()
        key, sub = genjax.split(genjax.key(KEY))
        tr = m.simulate(sub, ())
        for idx in range(10):
            old = f(tr.get_choices()["a", idx])
            req = StaticRequest({"a": IndexRequest(jnp.array(idx), Update(C.v(idx + 7.0)))})
            new_tr, w, _, _ = req.edit(key, tr, ())
            new = f(new_tr.get_choices()["a", idx])
            assert new == idx + 7.0
            assert f(w) == pytest.approx(lp(new) - lp(old), abs=1e-5)


# ---------------------------------------------------------------------------
# test_repeat_combinator.py
# ---------------------------------------------------------------------------
class TestRepeat:
    def test_importance_at_one_element(self):
        """:22-30"""
        @genjax.gen
        def model():
            return genjax.normal(0.0, 1.0) @ "x"
        tr, w = model.repeat(n=10).importance(genjax.key(314), C[1, "x"].set(3.0), ())
        assert lp(f(tr.get_choices()[1, "x"])) == f(w)

    def test_repeat_matches_vmap_of_equal_inputs(self):
        """:32-45"""
        @genjax.gen
        def square(x):
            return x * x
        key = genjax.key(314)
        r = square.repeat(n=10)(2)(key)
        assert tuple(r.shape) == (10,)
        assert np.array_equal(arr(square.vmap()(jnp.repeat(2, 10))(key)), arr(r))

    def test_repeat_combinator_is_the_decorator(self):
        """repeat.py:28-42 / :45-81: RepeatCombinator(gen_fn, n=) is what `@repeat(n=)` builds"""
        @genjax.gen
        def model(mu):
            return genjax.normal(mu, 1.0) @ "x"
        key = genjax.key(7)
        a = genjax.RepeatCombinator(model, n=6).simulate(key, (2.0,))
        b = genjax.repeat(n=6)(model).simulate(key, (2.0,))
        assert np.array_equal(arr(a.get_retval()), arr(b.get_retval())) and f(a.get_score()) == f(b.get_score())

    def test_nested_lookup(self):
        """:47-58: C[0, :, "x"] on a repeat of a repeat"""
        @genjax.gen
        def model():
            return genjax.normal(0.0, 1.0) @ "x"
        big = model.repeat(n=10).repeat(n=10)
        tr, _ = big.importance(genjax.key(0), C[jnp.array(0), :, "x"].set(jnp.ones(10)), ())
        assert np.array_equal(arr(tr.get_choices()[0, :, "x"]), np.ones(10))


# ---------------------------------------------------------------------------
# core/generative/test_core.py
# ---------------------------------------------------------------------------
class TestCore:
    def test_tupled_address_and_project(self):
        """:26-36"""
        @genjax.gen
        def m():
            x = genjax.normal(0.0, 1.0) @ ("x", "x0")
            return genjax.normal(x, 1.0) @ "y"
        tr = m.simulate(genjax.key(0), ())
        assert lp(f(tr.get_choices()["x", "x0"])) == f(tr.project(genjax.key(1), Selection.at["x", "x0"]))

    def test_project_and_get_subtrace(self):
        """:54-77: project(S[a]) = the sub-trace's score; the tuple spelling of get_subtrace is deprecated"""
        @genjax.gen
        def m():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = genjax.normal(0.0, 1.0) @ "y"
            return x, y
        tr = m.simulate(genjax.key(0), ())
        xs, ys = tr.project(genjax.key(1), S["x"]), tr.project(genjax.key(1), S["y"])
        with pytest.deprecated_call():
            assert f(xs) == f(tr.get_subtrace(("x",)).get_score())
        assert f(xs) == f(tr.get_subtrace("x").get_score()) and f(ys) == f(tr.get_subtrace("y").get_score())
        assert f(tr.get_score()) == pytest.approx(f(xs) + f(ys), abs=1e-6)

    def test_get_subtrace_nested(self):
        """:80-120: address components may be split across calls any way"""
        @genjax.gen
        def ff():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = genjax.normal(0.0, 1.0) @ "y"
            return x, y

        @genjax.gen
        def g():
            x, y = ff() @ "f"
            return x + y

        @genjax.gen
        def h():
            return g() @ "g"
        tr = g.simulate(genjax.key(1), ())
        f_tr = tr.get_subtrace("f")
        assert isinstance(f_tr, genjax.StaticTrace)
        for a in ("x", "y"):
            assert f(tr.get_subtrace("f", a).get_score()) == f(f_tr.get_subtrace(a).get_score())
        tr = h.simulate(genjax.key(2), ())
        want = f(tr.get_subtrace("g", "f", "x").get_score())
        assert f(tr.get_subtrace("g").get_subtrace("f").get_subtrace("x").get_score()) == want
        assert f(tr.get_subtrace("g").get_subtrace("f", "x").get_score()) == want
        assert f(tr.get_subtrace("g", "f").get_subtrace("x").get_score()) == want

    def test_get_subtrace_of_a_plate(self):
        """:147-156: per-element scores under a plate"""
        @genjax.vmap()
        @genjax.gen
        def m(x):
            return genjax.normal(x, 0.01) @ "y"
        tr = m.simulate(genjax.key(0), (jnp.arange(5.0),))
        s = tr.get_subtrace("y").get_score()
        assert tuple(s.shape) == (5,) and f(tr.get_score()) == pytest.approx(float(arr(s).sum()), rel=1e-5)

    def test_get_subtrace_of_a_scan(self):
        """:158-166"""
        @genjax.gen
        def m(state, step):
            return state + genjax.normal(step, 0.01) @ "y", None
        tr = m.scan().simulate(genjax.key(0), (5.0, jnp.arange(3.0)))
        s = tr.get_subtrace("y").get_score()
        assert tuple(s.shape) == (3,) and f(tr.get_score()) == pytest.approx(float(arr(s).sum()), rel=1e-5)

    def test_vmap_method_and_slice_addresses(self):
        """:172-192: chm[:, "v"] groups a plate's values"""
        @genjax.gen
        def model(x):
            v = genjax.normal(x, 1.0) @ "v"
            return (v, genjax.normal(v, 0.01) @ "q")
        tr = genjax.jit(model.vmap().simulate)(genjax.key(KEY), (jnp.array([10.0, 20.0, 30.0]),))
        varr, qarr = tr.get_retval()
        assert np.array_equal(arr(tr.get_choices()[:, "v"]), arr(varr))
        assert np.array_equal(arr(tr.get_choices()[:, "q"]), arr(qarr))

    def test_repeat_method(self):
        """:194-217"""
        @genjax.gen
        def model(x):
            return genjax.normal(x, 1.0) @ "x"
        key = genjax.key(KEY)
        vt = genjax.jit(model.vmap().simulate)(key, (jnp.zeros(3),))
        rt = genjax.jit(model.repeat(n=3).simulate)(key, (0.0,))
        assert np.array_equal(arr(rt.get_choices()[:, "x"]), arr(rt.get_retval()))
        assert np.array_equal(arr(vt.get_choices()[:, "x"]), arr(vt.get_retval()))
