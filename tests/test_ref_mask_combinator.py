"""The reference's behavioural tests of its MaskCombinator and of the masked scan sugar, restated against this package:

  /root/reference/tests/generative_functions/test_mask_combinator.py   (10 tests, all mirrored)
  /root/reference/tests/generative_functions/test_vmap_combinator.py   (the `mask()` half of
                                                                        test_vmap_combinator_vmap_pytree)

Run here on the CPU mirror of the C-ABI; tests/test_reference_gpu.py runs the same class through libgenmi_hip.so.
Where the reference jits a call its flag is an array; here a Python flag stays a host value and the trace's choices
keep it as `np.bool_` (static._host_flag) — the assertions are the reference's."""
import numpy as np
import pytest
import torch

import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import Diff
from genjax_amd import numpy as jnp
from genjax_amd.static import VmapTrace

pytestmark = pytest.mark.usefixtures("hostsim")

KEY = 314159


def f(x):
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


def arr(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def make_model():
    @genjax.mask
    @genjax.gen
    def model(x):
        z = genjax.normal(x, 1.0) @ "z"
        return z
    return model


def make_step(masks):
    @genjax.gen
    def step(x):
        _ = genjax.normal.mask().vmap(in_axes=(0, None, None))(masks, x, 1.0) @ "rats"
        return x
    return step


class TestMaskCombinator:
    def test_mask_simple_normal_true(self):
        model, key = make_model(), genjax.key(KEY)
        tr = model.simulate(key, (True, -4.0))
        assert f(tr.get_score()) == f(tr.inner.get_score())
        assert tr.get_retval() == genjax.Mask(tr.inner.get_retval(), True)
        tr = model.simulate(key, (False, -4.0))
        assert f(tr.get_score()) == 0.0
        assert tr.get_retval() == genjax.Mask(tr.inner.get_retval(), False)

    def test_mask_simple_normal_false(self):
        model, key = make_model(), genjax.key(KEY)
        tr = model.simulate(key, (False, 2.0))
        assert f(tr.get_score()) == 0.0
        assert not tr.get_retval().flag
        score, retval = model.assess(tr.get_choices(), tr.get_args())
        assert f(score) == 0.0
        assert not retval.flag
        _, w = model.importance(key, C["z"].set(-2.0), tr.get_args())
        assert f(w) == 0.0

    def test_mask_update_weight_to_argdiffs_from_true(self):
        model, key = make_model(), genjax.key(KEY)
        tr = model.simulate(key, (True, 2.0))                                  # pre-update, the mask is True
        argdiffs = (Diff.unknown_change(True), Diff.no_change(tr.get_args()[1]))      # True --> True
        w = tr.update(key, C.n(), argdiffs)[1]
        assert f(w) == f(tr.inner.update(key, C.n())[1])
        assert f(w) == 0.0
        argdiffs = (Diff.unknown_change(False), Diff.no_change(tr.get_args()[1]))     # True --> False
        w = tr.update(key, C.n(), argdiffs)[1]
        assert f(w) == -f(tr.get_score())

    @pytest.mark.parametrize("_again", [0, 1])       # (the reference holds this test twice)
    def test_mask_update_weight_to_argdiffs_from_false(self, _again):
        model, key = make_model(), genjax.key(KEY)
        tr = model.simulate(key, (False, 2.0))                                 # pre-update mask arg is False
        w = tr.update(key, C.n(), (Diff.unknown_change(True), Diff.no_change(tr.get_args()[1])))[1]     # False --> True
        assert f(w) == f(tr.inner.update(key, C.n())[1]) + f(tr.inner.get_score())
        assert f(w) == f(tr.inner.update(key, C.n())[0].get_score())
        w = tr.update(key, C.n(), (Diff.unknown_change(False), Diff.no_change(tr.get_args()[1])))[1]    # False --> False
        assert f(w) == 0.0
        assert f(w) == f(tr.get_score())

    def test_mask_vmap(self):
        key = genjax.key(KEY)

        @genjax.gen
        def init():
            x = genjax.normal(0.0, 1.0) @ "x"
            return x
        masks = jnp.array([True, False, True])

        @genjax.gen
        def model_2():
            vmask_init = init.mask().vmap(in_axes=(0))(masks) @ "init"
            return vmask_init
        tr = model_2.simulate(key, ())
        retval = tr.get_retval()
        flag, val = arr(retval.flag), arr(retval.value)
        lps = np.array([f(genjax.normal.assess(C.v(float(v)), (0.0, 1.0))[0]) for v in val], np.float32)
        assert f(tr.get_score()) == pytest.approx(float(np.sum(flag * lps)), abs=1e-6)
        vmap_tr = tr.get_subtrace("init")
        assert isinstance(vmap_tr, VmapTrace)
        inner_scores = arr(vmap_tr.inner.get_score())
        assert f(tr.get_score()) == np.float32(inner_scores[0] + inner_scores[2])   # the sub-scores masked True

    def test_masked_iterate_final_update(self):
        step = make_step(jnp.array([True, True]))
        key = genjax.key(0)
        mask_steps = jnp.arange(10) < 5
        model = step.masked_iterate_final()
        init_particle = model.simulate(key, (0.0, mask_steps))
        assert f(init_particle.get_retval()) == 0.0
        step_particle, step_weight, _, _ = model.update(key, init_particle, C.n(), Diff.no_change((0.0, mask_steps)))
        assert f(step_weight) == 0.0
        assert f(step_particle.get_retval()) == 0.0
        # inference keeps working when the model is extended by unmasking a value
        argdiffs_ = (Diff.no_change(0.0), Diff.unknown_change(jnp.arange(10) < 6))
        step_particle, step_weight, _, _ = model.update(key, init_particle, C.n(), argdiffs_)
        assert f(step_weight) != 0.0
        assert f(step_particle.get_score()) == np.float32(np.float32(f(step_weight)) + np.float32(f(init_particle.get_score())))

    def test_masked_iterate(self):
        step = make_step(jnp.array([True, True]))
        key = genjax.key(0)
        mask_steps = jnp.arange(10) < 5
        model = step.masked_iterate()
        init_particle = model.simulate(key, (0.0, mask_steps))
        assert np.array_equal(arr(init_particle.get_retval()), np.zeros(11)), \
            "0.0 is threaded through 10 times in addition to the initial value"

    def test_mask_scan_update_type_error(self):
        key = genjax.key(KEY)

        @genjax.gen
        def model_inside():
            masks = jnp.array([True, False, True])
            return genjax.normal(0.0, 1.0).mask().vmap()(masks) @ "init"
        outside_mask = jnp.array([True, False, True])

        @genjax.gen
        def model_outside():
            return genjax.normal(0.0, 1.0).mask().vmap()(outside_mask) @ "init"
        inside_tr = model_inside.simulate(key, ())
        outside_tr = model_outside.simulate(key, ())
        assert f(outside_tr.get_score()) == f(inside_tr.get_score())
        assert inside_tr.get_retval() == outside_tr.get_retval()
        assert inside_tr.get_choices() == outside_tr.get_choices()
        retval = outside_tr.get_retval()
        flag, val = arr(retval.flag), arr(retval.value)
        lps = np.array([f(genjax.normal.assess(C.v(float(v)), (0.0, 1.0))[0]) for v in val], np.float32)
        assert f(outside_tr.get_score()) == pytest.approx(float(np.sum(flag * lps)), abs=1e-6)

    def test_mask_fails_with_vector_mask(self):
        key = genjax.key(KEY)

        @genjax.gen
        def model():
            return genjax.normal(0.0, 1.0) @ "x"
        masks = jnp.array([True, True, False])
        with pytest.raises(TypeError):
            model.mask().simulate(key, (masks,))
        tr = model.mask().vmap().simulate(key, (masks,))          # it is still possible to vmap
        assert np.all(arr(tr.get_retval().flag) == np.asarray(masks))

    def test_vmap_combinator_vmap_pytree_masked(self):
        """test_vmap_combinator.py:121-146: the `mask()` half"""
        @genjax.gen
        def model2(x):
            _ = genjax.normal(x, 1.0) @ "y"
            return x
        model_mv2 = model2.mask().vmap()
        masks = jnp.array([True, False] * 5)
        xs = jnp.arange(0.0, 10.0, 1.0)
        tr = model_mv2.simulate(genjax.key(KEY), (masks, xs))     # mapping along several arguments by the default
        assert np.array_equal(arr(tr.get_retval().value), np.asarray(xs))
        assert np.array_equal(arr(tr.get_retval().flag), np.asarray(masks))
