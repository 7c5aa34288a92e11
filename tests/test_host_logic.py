"""Host-logic tests (no GPU): the Python layer — tracer, site-program encoder,
binding plans, GFI semantics, SMC combinators — runs against tests/hostsim (the
product's own interpreter template compiled for the host) and is checked
against the reference's behavioural tests and the CPU oracle.

Models and tolerances follow the reference's tests:
  tests/generative_functions/test_distributions.py, test_static_gen_fn.py,
  tests/inference/test_smc.py, test_requests.py, README.md:88-123.
"""
import math

import numpy as np
import pytest
import torch

import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import SelectionBuilder as S
from genjax_amd import numpy as jnp
from oracle import genjax_oracle as O

pytestmark = pytest.mark.usefixtures("hostsim")


def f(x):
    return float(x) if not isinstance(x, torch.Tensor) else float(x.item())


# ---------------------------------------------------------------------------
# choice maps / selections (core/generative/choice_map.py)
# ---------------------------------------------------------------------------
class TestChoiceMap:
    def test_builders_and_lookup(self):
        chm = C["a", "b"].set(1.0) | C["c"].set(2.0)
        assert chm["a", "b"] == 1.0 and chm["c"] == 2.0
        assert ("a", "b") in chm and "zz" not in chm
        assert chm("a")["b"] == 1.0
        assert genjax.ChoiceMap.kw(x=3)["x"] == 3
        assert genjax.ChoiceMap.d({"x": 1, ("y", "z"): 2})["y", "z"] == 2
        assert genjax.ChoiceMap.empty().static_is_empty()
        assert C.v(5.0).get_value() == 5.0
        assert chm.at["d"].set(4.0)["d"] == 4.0
        with pytest.raises(genjax.ChoiceMapNoValueAtAddress):
            chm["nope"]

    def test_merge_first_wins(self):
        a, b = C["x"].set(1.0), C["x"].set(2.0) | C["y"].set(3.0)
        m = a | b
        assert m["x"] == 1.0 and m["y"] == 3.0          # Or.build: first operand wins

    def test_filter_and_selection(self):
        chm = C.kw(x=1.0, y=2.0) | C["z", "w"].set(3.0)
        assert chm.filter(S["x"]).to_dict() == {"x": 1.0}
        assert set(chm.filter(~S["x"]).to_dict()) == {"y", ("z", "w")}
        assert set(chm.filter(S["x"] | S["z"]).to_dict()) == {"x", ("z", "w")}
        assert chm.filter(S["x"] & S["y"]).static_is_empty()
        sel = chm.get_selection()
        assert sel["x"] and sel["z", "w"] and not sel["q"]
        assert genjax.Selection.all()["anything"] and not genjax.Selection.none()["anything"]
        assert () in genjax.Selection.all()("x")

    def test_target_filter_to_unconstrained(self):
        @genjax.gen
        def m():
            x = genjax.normal(0.0, 1.0) @ "x"
            _ = genjax.normal(x, 1.0) @ "y"
        t = genjax.Target(m, (), C["y"].set(3.0))
        assert set(t.filter_to_unconstrained(C.kw(x=1.0, y=3.0)).to_dict()) == {"x"}
        assert t["y"] == 3.0


# ---------------------------------------------------------------------------
# distributions as generative functions (test_distributions.py:25-193)
# ---------------------------------------------------------------------------
class TestDistributions:
    def test_simulate_score_is_assess(self):
        key = genjax.key(314159)
        for dist, args in [(genjax.normal, (0.0, 1.0)), (genjax.beta, (2.0, 3.0)), (genjax.flip, (0.3,)),
                           (genjax.uniform, (-1.0, 2.0))]:
            tr = dist.simulate(key, args)
            score, v = dist.assess(tr.get_choices(), args)
            assert f(tr.get_score()) == f(score)

    def test_importance_rules(self):
        key = genjax.key(314159)
        tr, w = genjax.normal.importance(key, genjax.ChoiceMap.empty(), (0.0, 1.0))
        assert f(w) == 0.0                                  # unconstrained: w = 0
        tr, w = genjax.normal.importance(key, C.v(1.0), (0.0, 1.0))
        assert f(w) == pytest.approx(-0.5 - 0.9189385, abs=1e-6) and f(tr.get_score()) == f(w)
        assert f(tr.get_choices().get_value()) == 1.0

    def test_update_weights(self):
        key = genjax.key(314159)
        tr = genjax.normal.simulate(key, (0.0, 1.0))
        old_v, old_s = f(tr.get_retval()), f(tr.get_score())
        # no constraint, changed args: w = logpdf(old; new args) - old score
        new_tr, w, _, bwd = genjax.Update(genjax.ChoiceMap.empty()).edit(key, tr, genjax.Diff.unknown_change((1.0, 1.0)))
        assert f(w) == pytest.approx(f(genjax.normal.logpdf(old_v, 1.0, 1.0)) - old_s, abs=1e-6)
        assert f(new_tr.get_retval()) == old_v
        # constraint: w = logpdf(new) - old score, discard = old value
        new_tr, w, _, bwd = genjax.Update(C.v(1.0)).edit(key, tr, genjax.Diff.no_change((0.0, 1.0)))
        assert f(w) == pytest.approx(f(genjax.normal.logpdf(1.0, 0.0, 1.0)) - old_s, abs=1e-6)
        assert f(bwd.constraint.get_value()) == old_v

    def test_kwargs_and_logit_warning(self):
        key = genjax.key(0)
        s1 = genjax.normal.assess(C.v(0.5), (0.0, 0.1))[0]

        @genjax.gen
        def m():
            return genjax.normal(loc=0.0, scale=0.1) @ "x"
        s2, _ = m.assess(C.kw(x=0.5), ())
        assert f(s1) == f(s2)
        with pytest.warns(DeprecationWarning):
            @genjax.gen
            def m2():
                return genjax.categorical([0.0, 0.0]) @ "k"
            m2.simulate(key, ())

    def test_logpdfs_match_oracle_bitwise(self):
        xs = np.linspace(-3, 3, 101).astype(np.float32)
        got = genjax.normal.logpdf(torch.from_numpy(xs), 0.5, 1.5).numpy()
        assert np.array_equal(got, O.normal.logpdf(xs, np.float32(0.5), np.float32(1.5)))
        ps = np.linspace(0.01, 0.99, 99).astype(np.float32)
        got = genjax.beta.logpdf(torch.from_numpy(ps), 2.0, 3.0).numpy()
        assert np.array_equal(got, O.beta.logpdf(ps, np.float32(2.0), np.float32(3.0)))


# ---------------------------------------------------------------------------
# static language (test_static_gen_fn.py)
# ---------------------------------------------------------------------------
class TestStatic:
    def test_reference_literal(self):
        """test_static_gen_fn.py:317-318"""
        @genjax.gen
        def model():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2
        score, retval = model.assess(C.kw(y1=1.0, y2=-1.0), ())
        assert f(score) == pytest.approx(-2.837877, abs=5e-7)
        assert f(retval) == 0.0

    def test_score_is_sum_of_site_logpdfs(self):
        @genjax.gen
        def model():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(y1, 1.0) @ "y2"
            return y1 + y2
        key = genjax.key(314159)
        tr = model.simulate(key, ())
        ch = tr.get_choices()
        s = f(genjax.normal.logpdf(ch["y1"], 0.0, 1.0)) + f(genjax.normal.logpdf(ch["y2"], ch["y1"], 1.0))
        assert f(tr.get_score()) == pytest.approx(s, abs=1e-6)
        score, _ = model.assess(ch, ())
        assert f(score) == f(tr.get_score())
        assert f(tr.get_retval()) == pytest.approx(f(ch["y1"]) + f(ch["y2"]))

    def test_missing_address_and_reuse(self):
        @genjax.gen
        def model():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2
        with pytest.raises(genjax.MissingAddress):
            model.assess(C.kw(y1=1.0), ())

        @genjax.gen
        def bad():
            _ = genjax.normal(0.0, 1.0) @ "y"
            _ = genjax.normal(0.0, 1.0) @ "y"
        with pytest.raises(genjax.AddressReuse):
            bad.simulate(genjax.key(0), ())

    def test_importance_weight_is_constrained_logpdf(self):
        @genjax.gen
        def model():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(y1, 2.0) @ "y2"
            return y2
        key = genjax.key(1)
        tr, w = model.importance(key, genjax.ChoiceMap.empty(), ())
        assert f(w) == 0.0
        tr, w = model.importance(key, C.kw(y2=0.5), ())
        y1 = tr.get_choices()["y1"]
        assert f(w) == pytest.approx(f(genjax.normal.logpdf(0.5, y1, 2.0)), abs=1e-6)
        tr, w = model.importance(key, C.kw(y1=0.1, y2=0.5), ())
        assert f(w) == pytest.approx(f(tr.get_score()), abs=1e-6)

    def test_nested_generative_functions(self):
        @genjax.gen
        def inner(m):
            a = genjax.normal(m, 1.0) @ "a"
            b = genjax.normal(a, 1.0) @ "b"
            return a + b

        @genjax.gen
        def outer():
            x = genjax.normal(0.0, 1.0) @ "x"
            r = inner(x) @ "sub"
            _ = genjax.normal(r, 1.0) @ "y"
            return r

        @O.gen
        def o_inner(m):
            a = O.normal(m, 1.0) @ "a"
            b = O.normal(a, 1.0) @ "b"
            return a + b

        @O.gen
        def o_outer():
            x = O.normal(0.0, 1.0) @ "x"
            r = o_inner(x) @ "sub"
            _ = O.normal(r, 1.0) @ "y"
            return r
        n = 257
        tr = outer.simulate(genjax.split(genjax.key(5), n), ())
        tro = o_outer.simulate(O.split(O.key(5), n), ())
        ch, cho = tr.get_choices(), tro.get_choices()
        for a in ("x", ("sub", "a"), ("sub", "b"), "y"):
            assert np.array_equal(ch[a].numpy(), cho[a]), a
        assert np.array_equal(tr.get_score().numpy(), tro.get_score())
        assert np.array_equal(tr.get_retval().numpy(), tro.get_retval())
        # constrain a nested address
        tr, w = outer.importance(genjax.split(genjax.key(6), n), C["sub", "b"].set(0.25) | C["y"].set(1.0), ())
        tro, wo = o_outer.importance(O.split(O.key(6), n),
                                     O.C.d({("sub", "b"): np.float32(0.25), "y": np.float32(1.0)}), ())
        assert np.array_equal(w.numpy(), wo)
        assert np.array_equal(tr.get_subtrace("sub", "a").get_retval().numpy(), tro.subtraces["sub"].subtraces["a"].value)

    def test_control_flow_and_tables(self):
        """lax.cond on a traced flip + table lookup by a traced categorical (tests/inference/test_smc.py:59-96)"""
        @genjax.gen
        def flip_flip():
            v1 = genjax.flip(0.5) @ "x"
            p = jnp.lax.cond(v1, lambda: 0.9, lambda: 0.3)
            _ = genjax.flip(p) @ "y"
            return p

        @O.gen
        def o_flip_flip():
            v1 = O.flip(0.5) @ "x"
            p = np.where(v1, np.float32(0.9), np.float32(0.3))
            _ = O.flip(p) @ "y"
            return p
        n = 300
        tr = flip_flip.simulate(genjax.split(genjax.key(2), n), ())
        tro = o_flip_flip.simulate(O.split(O.key(2), n), ())
        assert np.array_equal(tr.get_choices()["x"].numpy(), tro.get_choices()["x"])
        assert np.array_equal(tr.get_choices()["y"].numpy(), tro.get_choices()["y"])
        assert np.array_equal(tr.get_score().numpy(), tro.get_score())

        @genjax.gen
        def mixture():
            idx = genjax.categorical(probs=[0.5, 0.25, 0.25]) @ "idx"
            means = jnp.array([0.0, 10.0, 11.0])
            x = genjax.normal(means[idx], 1.0) @ "x"
            return x

        @O.gen
        def o_mixture():
            idx = O.categorical(probs=[0.5, 0.25, 0.25]) @ "idx"
            means = np.array([0.0, 10.0, 11.0], np.float32)
            x = O.normal(means[idx], 1.0) @ "x"
            return x
        tr = mixture.simulate(genjax.split(genjax.key(3), n), ())
        tro = o_mixture.simulate(O.split(O.key(3), n), ())
        assert np.array_equal(tr.get_choices()["idx"].numpy(), tro.get_choices()["idx"])
        assert np.array_equal(tr.get_choices()["x"].numpy(), tro.get_choices()["x"])
        np.testing.assert_allclose(tr.get_score().numpy(), tro.get_score(), rtol=0, atol=0)

    def test_vector_valued_site(self):
        """one site key, element j takes counter j; log_prob summed (distribution.py:383-396)"""
        sig = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]

        @genjax.gen
        def schools(ys):
            mu = genjax.normal(0.0, 5.0) @ "mu"
            log_tau = genjax.normal(0.0, 1.0) @ "log_tau"
            theta = genjax.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
            _ = genjax.normal(theta, jnp.array(sig)) @ "y"
            return theta

        @O.gen
        def o_schools(ys):
            mu = O.normal(0.0, 5.0) @ "mu"
            log_tau = O.normal(0.0, 1.0) @ "log_tau"
            theta = O.normal(mu[..., None] * np.ones(8, np.float32), O.exp(log_tau)[..., None] * np.ones(8, np.float32)) @ "theta"
            _ = O.normal(theta, np.array(sig, np.float32)) @ "y"
            return theta
        ys = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)
        n = 130
        tr, w = schools.importance(genjax.split(genjax.key(9), n), C["y"].set(ys), (ys,))
        tro, wo = o_schools.importance(O.split(O.key(9), n), O.C.d({"y": ys}), (ys,))
        assert tuple(tr.get_choices()["theta"].shape) == (n, 8)
        assert np.array_equal(tr.get_choices()["theta"].numpy(), tro.get_choices()["theta"])
        assert np.array_equal(w.numpy(), wo)
        assert np.array_equal(tr.get_score().numpy(), tro.get_score())


# ---------------------------------------------------------------------------
# SMC (tests/inference/test_smc.py, README.md:88-123)
# ---------------------------------------------------------------------------
class TestSMC:
    def test_exact_flip_flip_trivial(self):
        @genjax.gen
        def flip_flip_trivial():
            _ = genjax.flip(0.5) @ "x"
            _ = genjax.flip(0.7) @ "y"
        key = genjax.key(314159)
        problem = genjax.Target(flip_flip_trivial, (), C["y"].set(True))
        z_exact = f(genjax.flip.assess(problem.constraint.get_submap("y"), (0.7,))[0])
        assert z_exact == pytest.approx(math.log(0.7), abs=1e-6)
        z = genjax.inference.smc.Importance(problem).log_marginal_likelihood_estimate(key)
        assert f(z) == pytest.approx(z_exact, rel=1e-1)
        z = genjax.inference.smc.ImportanceK(problem, k_particles=1000).log_marginal_likelihood_estimate(key)
        assert f(z) == pytest.approx(z_exact, rel=1e-3)

    def test_exact_flip_flip(self):
        @genjax.gen
        def flip_flip():
            v1 = genjax.flip(0.5) @ "x"
            p = jnp.lax.cond(v1, lambda: 0.9, lambda: 0.3)
            _ = genjax.flip(p) @ "y"
        key = genjax.key(314159)
        problem = genjax.Target(flip_flip, (), C["y"].set(True))
        z = genjax.inference.smc.ImportanceK(problem, k_particles=2000).log_marginal_likelihood_estimate(key)
        assert f(z) == pytest.approx(math.log(0.5 * 0.9 + 0.5 * 0.3), rel=1e-1)

    def test_importancek_matches_oracle(self):
        @genjax.gen
        def m():
            x = genjax.flip(0.5) @ "x"
            _ = genjax.flip(0.7) @ "y"

        @O.gen
        def om():
            x = O.flip(0.5) @ "x"
            _ = O.flip(0.7) @ "y"
        coll = genjax.inference.smc.ImportanceK(genjax.Target(m, (), C["y"].set(True)), k_particles=500).run_smc(genjax.key(7))
        ocoll = O.ImportanceK(O.Target(om, (), O.C.kw(y=True)), 500).run_smc(O.key(7))
        assert np.array_equal(coll.get_log_weights().numpy(), ocoll.get_log_weights())
        assert np.array_equal(coll.get_particles().get_choices()["x"].numpy(), ocoll.get_particles().get_choices()["x"])
        assert f(coll.get_log_marginal_likelihood_estimate()) == pytest.approx(
            float(ocoll.get_log_marginal_likelihood_estimate()), rel=1e-6)
        # sample_particle: same Gumbel-max index
        k2 = genjax.key(8)
        assert int(coll.sample_index(k2)) == int(ocoll.sample_index(O.key(8)))

    def test_non_marginal_target(self):
        @genjax.gen
        def model():
            idx = genjax.categorical(probs=[0.5, 0.25, 0.25]) @ "idx"
            means = jnp.array([0.0, 10.0, 11.0])
            x = genjax.normal(means[idx], 1.0) @ "x"
            y = genjax.normal(means[idx], 1.0) @ "y"
            return x, y
        marginal_model = model.marginal(selection=S["x"] | S["y"])
        with pytest.raises(TypeError):
            genjax.Target(marginal_model, (), C["x"].set(1.0))

    def test_readme_quickstart(self):
        """README.md:88-123: 50 trials of SIR with K = 50; mean p -> 0.6 / 0.4."""
        @genjax.gen
        def beta_bernoulli(a, b):
            p = genjax.beta(a, b) @ "p"
            v = genjax.flip(p) @ "v"
            return v

        def run_inference(obs: bool):
            target = genjax.Target(beta_bernoulli, (2.0, 2.0), genjax.ChoiceMap.d({"v": obs}))
            alg = genjax.inference.smc.ImportanceK(target, k_particles=50)
            key = genjax.key(314159)
            sub_keys = genjax.split(key, 50)
            _, p_chm = genjax.vmap(alg.random_weighted, in_axes=(0, None))(sub_keys, target)
            assert tuple(p_chm["p"].shape) == (50,)
            return f(jnp.mean(p_chm["p"]))
        t, fl = run_inference(True), run_inference(False)
        assert t == pytest.approx(0.6, abs=0.09) and fl == pytest.approx(0.4, abs=0.09)   # 3 sigma_MC


# ---------------------------------------------------------------------------
# edit requests (tests/inference/test_requests.py)
# ---------------------------------------------------------------------------
class TestRequests:
    def test_simple_normal_regenerate(self):
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2
        key = genjax.key(314159)
        key, sub_key = genjax.split(key)
        tr = simple_normal.simulate(sub_key, ())
        for addr in ("y1", "y2"):
            old_v = tr.get_choices()[addr]
            new_tr, fwd_w, _, bwd_request = genjax.Regenerate(S[addr]).edit(key, tr, ())
            new_v = new_tr.get_choices()[addr]
            old_d, new_d = genjax.normal.logpdf(old_v, 0.0, 1.0), genjax.normal.logpdf(new_v, 0.0, 1.0)
            assert f(fwd_w) != 0.0 and f(fwd_w) == f(new_d - old_d)
            assert f(old_v) != f(new_v)
            old_tr, bwd_w, _, _ = bwd_request.edit(sub_key, new_tr, ())
            assert f(bwd_w) != 0.0 and f(fwd_w) + f(bwd_w) == 0.0
            assert f(old_tr.get_choices()[addr]) == f(old_v)
        new_tr, fwd_w, _, bwd_request = genjax.Regenerate(S["y1"] | S["y2"]).edit(key, tr, ())
        old_tr, bwd_w, _, _ = bwd_request.edit(key, new_tr, ())
        assert f(fwd_w) + f(bwd_w) == 0.0
        assert f(old_tr.get_choices()["y2"]) == f(tr.get_choices()["y2"])

    def test_linked_normal_regenerate(self):
        @genjax.gen
        def linked_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            _ = genjax.normal(y1, 1.0) @ "y2"
        key = genjax.key(314159)
        key, sub_key = genjax.split(key)
        tr = linked_normal.simulate(sub_key, ())
        ch = tr.get_choices()
        old = f(genjax.normal.logpdf(ch["y1"], 0.0, 1.0)) + f(genjax.normal.logpdf(ch["y2"], ch["y1"], 1.0))
        new_tr, fwd_w, _, _ = genjax.Regenerate(S["y1"]).edit(key, tr, ())
        ch = new_tr.get_choices()
        new = f(genjax.normal.logpdf(ch["y1"], 0.0, 1.0)) + f(genjax.normal.logpdf(ch["y2"], ch["y1"], 1.0))
        assert f(fwd_w) != 0.0 and f(fwd_w) == pytest.approx(new - old, rel=1e-5)

    def _mh(self, model, request, steps, key, obs):
        key, sub_key = genjax.split(key)
        tr, _ = model.importance(sub_key, obs, ())
        for _ in range(steps):
            key, sub_key = genjax.split(key)
            new_tr, w, _, _ = request.edit(sub_key, tr, ())
            key, sub_key = genjax.split(key)
            check = f(jnp.log(genjax.uniform.sample(sub_key, 0.0, 1.0))) < f(w)
            tr = new_tr if check else tr
        return tr

    def test_linked_normal_convergence(self):
        @genjax.gen
        def linked_normal():
            y1 = genjax.normal(0.0, 3.0) @ "y1"
            _ = genjax.normal(y1, 0.01) @ "y2"
        tr = self._mh(linked_normal, genjax.Regenerate(S["y1"]), 200, genjax.key(314159), C.kw(y2=3.0))
        assert f(tr.get_choices()["y1"]) == pytest.approx(3.0, rel=1e-2)

    def test_rejuvenate_prior_proposal_weight_zero(self):
        @genjax.gen
        def simple_normal():
            _ = genjax.normal(0.0, 1.0) @ "y1"
        key = genjax.key(314159)
        key, sub_key = genjax.split(key)
        tr = simple_normal.simulate(sub_key, ())
        old_v = tr.get_choices()["y1"]
        request = genjax.StaticRequest({"y1": genjax.Rejuvenate(genjax.normal, lambda chm: (0.0, 1.0))})
        new_tr, w, _, _ = request.edit(sub_key, tr, ())
        assert f(old_v) != f(new_tr.get_choices()["y1"])
        assert f(w) == 0.0

    def test_linked_normal_rejuvenate_convergence(self):
        @genjax.gen
        def linked_normal():
            y1 = genjax.normal(0.0, 3.0) @ "y1"
            _ = genjax.normal(y1, 0.001) @ "y2"
        request = genjax.StaticRequest({"y1": genjax.Rejuvenate(genjax.normal, lambda chm: (chm.get_value(), 0.3))})
        tr = self._mh(linked_normal, request, 100, genjax.key(314159), C.kw(y2=3.0))
        assert f(tr.get_choices()["y1"]) == pytest.approx(3.0, rel=5e-3)


# ---------------------------------------------------------------------------
# build-defined SMC moves vs the oracle
# ---------------------------------------------------------------------------
class TestSMCMoves:
    def test_sweep_matches_oracle(self):
        from tests import parity
        res = parity.check_lgssm_sweep(n=3000, T=6)
        assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"]
        assert res["lw_max_abs_diff"] == 0.0 and res["log_ml"] == res["log_ml_oracle"]

    @pytest.mark.parametrize("kind", ["systematic", "stratified", "multinomial", "multinomial_tiled", "multinomial_sorted"])
    def test_resample_extend_api(self, kind):
        from genjax_amd import workloads
        from genjax_amd.inference import smc
        init, step = workloads.make_lgssm(genjax)
        oi, ost = workloads.make_lgssm(O)
        n = 1500
        ys = workloads.lgssm_data(3)
        k0, k1, k2, k3 = genjax.split(genjax.key(11), 4)
        ok = O.split(O.key(11), 4)
        coll = smc.ImportanceK(genjax.Target(init, (), C["y"].set(float(ys[0]))), k_particles=n).run_smc(k0)
        ocoll = O.ImportanceK(O.Target(oi, (), O.C.kw(y=np.float32(ys[0]))), n).run_smc(ok[0])
        assert np.array_equal(coll.get_log_weights().numpy(), ocoll.get_log_weights())
        res = smc.resample(k1, coll, kind)
        cdf, total, M, shift = O.weight_cdf(ocoll.get_log_weights())
        oanc = O.ancestors_of_kind(smc._KINDS[kind], ok[1], cdf)
        assert np.array_equal(res.ancestors.numpy(), oanc)
        ext = smc.extend(k2, res, step, lambda tr: (tr.get_retval(),), C["y"].set(float(ys[1])))
        ox = ocoll.get_particles().get_retval()[oanc]
        otr, ow = ost.importance(O.split(ok[2], n), O.C.kw(y=np.float32(ys[1])), (ox,))
        assert np.array_equal(ext.get_particles().get_retval().numpy(), otr.get_retval())
        assert np.array_equal(ext.get_log_weights().numpy(), ow)
        lml = f(ext.get_log_marginal_likelihood_estimate())
        ref = O.log_ml_increment(M, total, shift, n) + float(O.logsumexp(ow) - np.log(np.float32(n)))
        assert lml == pytest.approx(ref, abs=1e-5)


def test_nonlinear_ssm_with_mh_rejuvenation_matches_oracle():
    """BASELINE config 3 in miniature (resample -> fused MH Rejuvenate -> extend)."""
    from tests import parity
    res = parity.check_nlssm_mh(n=1200, T=4)
    assert res["ok"], res
    assert all(0.0 < s["acc_rate"] <= 1.0 for s in res["steps"])


def test_plates_match_oracle():
    from tests import parity
    parity.check_plates(n=257)


def test_global_resampling_routes_match_oracle():
    """gmx_shard_plan / gmx_shard_route, all ranks emulated in one process."""
    from tests import parity
    # shards start on a 1024-particle CDF tile (the two-level CDF is defined on GLOBAL tiles)
    assert not parity.check_shard_route(1024, 4)["overflow"]
    assert not parity.check_shard_route(1024, 4, kind=O.STRATIFIED, seed=3)["overflow"]
    assert not parity.check_shard_route(1024, 3, skew=2.0, seed=1)["overflow"]
    assert parity.check_shard_route(1024, 3, skew=2.0, seed=1, capacity=5)["overflow"]
    assert not parity.check_shard_route(1024, 8, skew=-3.0, seed=2)["overflow"]
    assert not parity.check_shard_route(64, 2, dead=True)["overflow"]          # no mass anywhere
    assert not parity.check_shard_route(1000, 1)["overflow"]
    assert not parity.check_shard_route(2048, 4, fused=True, seed=4)["overflow"]
    assert parity.check_shard_route(1024, 3, skew=2.0, seed=1, capacity=5, fused=True)["overflow"]
    # the two-collective form (tile statistics instead of a CDF array and a max all-reduce)
    assert not parity.check_shard_route(2048, 4, fused="tiles", seed=4)["overflow"]
    assert not parity.check_shard_route(1024, 8, fused="tiles", skew=-3.0, seed=2, kind=O.STRATIFIED)["overflow"]
    assert parity.check_shard_route(1024, 3, skew=2.0, seed=1, capacity=5, fused="tiles")["overflow"]
    assert not parity.check_shard_route(64, 2, dead=True, fused="tiles")["overflow"]
    assert not parity.check_shard_route(1000, 1, fused="tiles")["overflow"]
    # gmx_shard_step_fused: straight from the gathered statistics table
    assert not parity.check_shard_route(2048, 4, fused="stats", seed=4)["overflow"]
    assert not parity.check_shard_route(1024, 8, fused="stats", skew=-3.0, seed=2, kind=O.STRATIFIED)["overflow"]
    assert parity.check_shard_route(1024, 3, skew=2.0, seed=1, capacity=5, fused="stats")["overflow"]
    assert not parity.check_shard_route(64, 2, dead=True, fused="stats")["overflow"]
    assert not parity.check_shard_route(1000, 1, fused="stats")["overflow"]
    assert not parity.check_shard_route(2048, 4, seed=12, spike=14.0, fused="stats")["overflow"]


def test_conditional_smc_and_proposals():
    from tests import parity
    parity.check_csmc(k=257)


def test_runtime_indexed_addresses():
    """ref choice_map.py:1453-1531 with a traced index: one plate index per particle"""
    from tests import parity
    parity.check_runtime_indexed()


def test_conditional_smc_under_a_batch_of_keys():
    """ref smc.py:317-351, 398-465, sp.py:217-240 under vmap: one launch set over [keys, K]"""
    from tests import parity
    parity.check_batched_csmc(k=17, B=40)


def test_nested_marginal_and_change_target():
    from tests import parity
    parity.check_nested_marginal()


def test_change_target_away_from_subset_constraints():
    """ref smc.py:370-396 + sp.py:89-91 + choice_map.py:658-663, 1494-1496, 1714-1743: a first target that constrains a
    SUBSET of a plate's elements keeps the whole site among the latents; unrolled (n = 5) and loop (n = 40) plates"""
    from tests import parity
    parity.check_change_target_from_subset_constraints()
    parity.check_change_target_from_subset_constraints(n=40, k=17, seed=9)


def test_mixture_assignments_match_oracle():
    """BASELINE config 5 (integer gate) at a CPU-sized N."""
    from tests import parity
    parity.check_mixture_assignments(n=3000, K=64)
    parity.check_mixture_assignments(n=501, K=5, seed=2)


def test_jax_docs_values_through_the_product():
    k = genjax.key(42)
    assert float(genjax.normal.sample(k, 0.0, 1.0)) == float(np.float32(-0.028304616))
    new_key, subkey = genjax.split(k)
    assert float(genjax.normal.sample(subkey, 0.0, 1.0)) == float(np.float32(0.60576403))
    assert [int(v) for v in subkey.host()] == [64467757, 2916123636]
    ind = genjax.normal.sample(genjax.split(k, 3), 0.0, 1.0).numpy()
    assert np.all(np.abs(ind.astype(np.float64) - [0.07592554, 0.60576403, 0.4323065]) < 5.1e-9)    # digits as printed
    allatonce = genjax.normal.sample(k, np.zeros(3, np.float32), 1.0).numpy().reshape(-1)
    assert np.all(np.abs(allatonce.astype(np.float64) - [-0.02830462, 0.46713185, 0.29570296]) < 5.1e-9)


def test_scan_matches_oracle():
    from tests import parity
    parity.check_scan(n=257, T=6)


def test_long_scan_runs_as_a_loop_and_matches_oracle():
    """Scan of T > 16 steps = a counted loop in the site program (ref scan.py:200-294, 638-664)"""
    from tests import parity
    parity.check_scan_long(n=257, T=100)
    parity.check_scan_long(n=64, T=17)
    # the program is a loop, not T copies of the kernel
    from genjax_amd import static
    sizes = [int(ent[0].blob[2]) for ent in static._CACHE.values() if hasattr(ent[0], "blob")]
    assert min(sizes) < 64


def test_scan_carries_that_forward_each_other(hostsim):
    """shift-register and swap carries: the counted loop's carry update is a parallel copy"""
    from tests import parity
    parity.check_scan_carry_forms()


def test_plate_on_the_launch_axis_matches_oracle():
    """a plate of >= 4096 elements called directly under ONE key: its elements on the launch axis (combinators.Vmap.
    _launch_axis); simulate / importance / assess / Update and the fixed-tree plate sums, bit-exact vs the oracle"""
    from tests import parity
    parity.check_plate_on_the_launch_axis(n=4096)
    parity.check_plate_on_the_launch_axis(n=4096 * 3 + 17, seed=5)


def test_config5_gibbs_sweep_through_the_plate():
    """BASELINE config 5 through the GFI: generate_datapoint.repeat(n=N) + gibbs.enumerative_gibbs on its trace"""
    from tests import parity
    parity.check_mixture_gibbs_through_the_plate(n=5000)


def test_large_plates_run_as_a_loop_and_match_oracle():
    """Vmap of more than 16 elements = a counted loop in the site program (ref vmap.py:180-218); incl. edits"""
    from tests import parity
    parity.check_plates_long(n=130, P=40)
    parity.check_plates_long(n=33, P=17, seed=2)


def test_index_request_edits_one_element_of_a_long_plate():
    """IndexRequest on a long plate held per particle: slice / edit / lazy write-back (combinators._vmap_edit_index_o1),
    chains of edits past PATCH_DEPTH_MAX, int and per-particle index, a plate of plates; bit-exact vs the oracle"""
    from tests import parity
    parity.check_index_request_o1(edits=6)
    parity.check_index_request_o1(n=64, P=24, seed=3, edits=70, nested=False)


def test_program_cache_is_bounded(hostsim, monkeypatch):
    """GENMI_PROGRAM_CACHE: the LRU of compiled site programs holds at most that many entries (the oldest go), and a
    model evicted from it is simply compiled again"""
    import genjax_amd as G
    from genjax_amd import engine
    monkeypatch.setenv("GENMI_PROGRAM_CACHE", "3")
    cache = engine.new_cache()
    try:
        assert cache.limit == 3
        for k in range(6):
            cache[("k", k)] = object()
        assert len(cache) == 3 and cache.get(("k", 5)) is not None and cache.get(("k", 0)) is None
        assert cache.get(("k", 3)) is not None            # a hit moves the entry to the young end
        cache[("k", 6)] = object()
        assert cache.get(("k", 3)) is not None and cache.get(("k", 4)) is None
    finally:
        engine._ALL_CACHES.remove(cache)


def test_one_trace_of_a_model_with_large_plates_runs_site_by_site():
    """ref static.py:254-673 for ONE trace whose plates hold thousands of elements (4_index_request.ipynb c3-c9): the
    model's source runs on the host site by site, large plates on the launch axis (genjax_amd/sitewise.py); simulate /
    importance / assess / Update / StaticRequest + IndexRequest bit-exact vs the oracle, untouched sub-traces shared"""
    from tests import parity
    parity.check_one_trace_with_large_plates(n=5000)
    parity.check_one_trace_with_large_plates(n=4096 * 2 + 5, seed=4)
    parity.check_one_trace_with_large_vector_sites(n=5000)
    parity.check_one_trace_with_large_vector_sites(n=4096 + 7, K=5, seed=8)
    parity.check_mixture_notebook_model(n=5000, k=12)
    parity.check_mixture_notebook_model(n=5000, k=40, seed=2)          # the notebook's own sizes
    parity.check_mixture_notebook_model(n=700, k=12, seed=5)           # 65 .. 4095 elements: launch axis, element-order sums
    parity.check_one_trace_with_large_vector_sites(n=300, K=5, seed=3)


def test_mixture_notebook_gibbs_inference_as_written():
    """7_application_dirichlet_mixture_model.ipynb c6-c12 with this package where the notebook has genjax: ONE trace of
    `generate_data`, importance under the data, Gibbs sweeps whose moves draw with the library (`generate_cluster.vmap()
    .simulate`, `categorical.simulate(key, (local_densities,))`, `generate_cluster_weight.simulate`) and write back with
    `trace.update` — the populated clusters end up on the data (tools/experiments/notebook7_gibbs.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nb7", os.path.join(root, "tools", "experiments", "notebook7_gibbs.py"))
    nb7 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nb7)
    import tests.hostsim as hs
    hs.install()
    try:
        import genjax_amd as G
        n, k = 1600, 8
        tr, true_means, _, score0 = nb7.infer(n, k, 25, G._lib.get().device)
        ch = tr.get_choices()
        means = ch["clusters", "mean"].numpy()
        counts = np.bincount(ch["datapoints", "idx"].numpy(), minlength=k)
        big = means[counts > n // (4 * k)]
        # a Gibbs sampler may leave two neighbouring blocks of data in one cluster: every populated cluster sits ON the data
        # (within half the block spacing of a true mean), most of them on one block, and the joint density went up a lot
        assert len(big) >= k // 2 and all(np.min(np.abs(true_means - m)) < 5.5 for m in big), (np.sort(big), true_means)
        assert sum(np.min(np.abs(true_means - m)) < 1.0 for m in big) >= len(big) // 2
        assert float(tr.get_score()) > score0 + 1000.0, (float(tr.get_score()), score0)
    finally:
        hs.uninstall()


def test_large_plate_of_a_small_particle_batch_is_deferred():
    """ref vmap.py:180-218 under `jax.vmap` over particles (smc.py:308-310): a few dozen particles over a model whose last
    site is a plate of thousands of elements — the plate runs over particles x elements after the program
    (combinators.Vmap._defer); equal to the loop form and to the oracle bit for bit, incl. ImportanceK.run_smc"""
    from tests import parity
    parity.check_deferred_plate(B=64, n=4096)
    parity.check_deferred_plate(B=37, n=4096 * 2 + 3, seed=5)


def test_random_models_match_the_oracle():
    """tests/fuzz_models.py: 130 random `@gen` models (leaf sites, plates, scans, masked calls and plates, plates of
    scans, scans of plates; unrolled and loop sizes mixed) — simulate / importance / assess / update under new
    constraints and changed arguments / IndexRequest / regenerate; ImportanceK over such models; a plate of thousands of
    elements as ONE trace and under K particles — bit for bit against the oracle"""
    from tests import fuzz_models as F
    ran = 0
    for seed in range(130):          # (200 until round 6; the on-device fuzz and tools/experiments/fuzz_on_device.py draw hundreds more)
        try:
            F.run_one(seed)
            ran += 1
        except F.OverTheLimits:
            pass
    assert ran == 130, ran      # (a model that does not fit ONE launch runs as loops or as a chain of launches: none is left over)
    # ImportanceK over a random model and random constraints under ONE key
    ran = 0
    for seed in range(60):
        try:
            F.run_smc_one(seed)
            ran += 1
        except F.OverTheLimits:
            pass
    assert ran == 60, ran
    # a plate of thousands of elements as the last statement: ONE trace (site by site) and K particles (deferred)
    for seed in range(12):
        F.run_big_one(seed)


def test_update_under_a_changed_table_argument_rescores_every_element():
    """ref vmap.py:236-275 / scan.py:417-503 with an UnknownChange argument that is a launch-uniform table of more than
    16 elements (read at a run-time index inside the loop): found stale in round 4 (weight 0) — one plate, a plate of
    plates, a scan over a table, `means[idx]`"""
    from tests import parity
    parity.check_update_under_changed_table_arguments()
    parity.check_update_under_changed_table_arguments(B=4, n=40, seed=7)


def test_mask_combinator_and_masked_scans_match_oracle():
    """ref combinators/mask.py:96-262, scan.py:1050-1150: `gen_fn.mask()` under per-particle flags (all four update
    transitions), plates of masked elements (unrolled and as a loop), masked_iterate / masked_iterate_final incl. the
    update that unmasks a step — bit for bit against the oracle's restatement"""
    from tests import parity
    parity.check_mask_combinator()
    parity.check_mask_combinator(B=5, T=7, n_plate=17, seed=3)
    parity.check_masked_image_model()              # the masking notebook's image model and its chain of updates


def test_indexed_and_masked_constraints_match_oracle():
    from tests import parity
    parity.check_masked_constraints()


def test_plate_edits_match_oracle():
    from tests import parity
    parity.check_plate_edits(n=257)


def test_hmc_reference_behaviour_and_oracle():
    from tests import parity
    parity.check_hmc(n=257)


def test_nonlinear_ssm_mh_sweep_matches_oracle():
    """config 3 as one sweep (BootstrapSweep(rejuvenate=...)) == the oracle, bit for bit."""
    from tests import parity
    res = parity.check_nlssm_mh_sweep(n=1500, T=4)
    assert 0.5 < res["accept_rate"] <= 1.0


def test_autodiff_rules_against_finite_differences():
    """genjax_amd/autodiff.py (reverse mode over the IR, what HMC differentiates with) against
    central differences in float64, rule by rule — including the density ops."""
    from genjax_amd.autodiff import value_and_grad
    rng = np.random.default_rng(0)
    x = rng.uniform(0.5, 2.0, 200).astype(np.float32)
    y = rng.uniform(0.5, 2.0, 200).astype(np.float32)
    from math import lgamma
    nlp = lambda v, m, s_: -0.5 * ((v - m) / s_) ** 2 - np.log(s_) - 0.5 * np.log(2 * np.pi)
    cases = {
        "arith": (lambda a, b: a * b + a / b - b * 0.5 + (-a), lambda a, b: a * b + a / b - b * 0.5 - a),
        "exp_log": (lambda a, b: jnp.exp(a * 0.3) * jnp.log(b + 1.0) + jnp.log1p(a),
                    lambda a, b: np.exp(a * 0.3) * np.log(b + 1.0) + np.log1p(a)),
        "sqrt_sq": (lambda a, b: jnp.sqrt(a + b) + jnp.square(a - b), lambda a, b: np.sqrt(a + b) + (a - b) ** 2),
        "trig": (lambda a, b: jnp.sin(a) * jnp.cos(b) + jnp.tanh(a - b), lambda a, b: np.sin(a) * np.cos(b) + np.tanh(a - b)),
        "pow": (lambda a, b: jnp.power(a, b), lambda a, b: a ** b),
        "minmax": (lambda a, b: jnp.minimum(a, b) * 2.0 + jnp.maximum(a, b * 1.1),
                   lambda a, b: np.minimum(a, b) * 2.0 + np.maximum(a, b * 1.1)),
        "where": (lambda a, b: jnp.where(a > b, a * a, b * 3.0), lambda a, b: np.where(a > b, a * a, b * 3.0)),
        "sigmoid": (lambda a, b: jnp.sigmoid(a - b) + jnp.softplus(b),
                    lambda a, b: 1 / (1 + np.exp(-(a - b))) + np.log1p(np.exp(b))),
        "normal_x_loc": (lambda a, b: genjax.normal.sym_logpdf(a, (b, 0.7)), lambda a, b: nlp(a, b, 0.7)),
        "normal_scale": (lambda a, b: genjax.normal.sym_logpdf(0.3, (a, b)), lambda a, b: nlp(0.3, a, b)),
        "bernoulli_logits": (lambda a, b: genjax.bernoulli.sym_logpdf(1, (a - b,)),
                             lambda a, b: (a - b) - np.log1p(np.exp(a - b))),
        "flip": (lambda a, b: genjax.flip.sym_logpdf(True, (a / (a + b),)), lambda a, b: np.log(a / (a + b))),
    }
    for name, (fn, ref) in cases.items():
        v, (ga, gb) = value_and_grad(fn)(torch.from_numpy(x), torch.from_numpy(y))
        a64, b64, h = x.astype(np.float64), y.astype(np.float64), 1e-6
        fa = (ref(a64 + h, b64) - ref(a64 - h, b64)) / (2 * h)
        fb = (ref(a64, b64 + h) - ref(a64, b64 - h)) / (2 * h)
        assert np.allclose(v.numpy(), ref(a64, b64), rtol=3e-5, atol=2e-5), name
        assert np.max(np.abs(ga.numpy() - fa) / (1 + np.abs(fa))) < 2e-5, name
        assert np.max(np.abs(gb.numpy() - fb) / (1 + np.abs(fb))) < 2e-5, name


def test_dirichlet_matches_oracle_and_scipy():
    from tests import parity
    parity.check_dirichlet(n=2000)


def test_api_surface_and_derived_distributions():
    """SURVEY App. C names exist; log_normal / half_normal (compositions of the Normal sampler) equal the
    oracle's Normal stream through exp / abs bit for bit, and scipy's densities; Const rides in args."""
    from scipy import stats
    for name in ("Address AddressComponent Argdiffs Arguments ChoiceMap ChoiceMapBuilder EditRequest GenerativeFunction "
                 "GenerativeFunctionClosure Mask R Retdiff Score Selection SelectionBuilder Trace Update Weight DiffAnnotate "
                 "EmptyRequest Regenerate Closure Const PythonicPytree Pytree nth Diff NoChange UnknownChange gen trace trace_p "
                 "StaticGenerativeFunction StaticRequest AddressReuse MissingAddress Distribution ExactDensity exact_density "
                 "tfp_distribution normal beta bernoulli flip categorical uniform dirichlet half_cauchy half_normal log_normal "
                 "vmap Vmap repeat scan Scan IndexRequest VectorRequest Target Algorithm SampleDistribution Marginal marginal").split():
        assert hasattr(genjax, name), name

    @genjax.gen
    def m(k, s):
        genjax.normal.vmap(in_axes=(0, None))(jnp.zeros(k.unwrap()), s) @ "xs"
        a = genjax.log_normal(0.25, 0.5) @ "a"
        h = genjax.half_normal(2.0) @ "h"
        return a + h
    n = 4000
    tr = m.simulate(genjax.split(genjax.key(0), n), (genjax.Const(3), 1.0))
    ch = tr.get_choices()
    assert tuple(ch["xs"].shape) == (n, 3)
    keys = O.split(O.key(0), n)
    ka, kh = O.fold_in(keys, 2), O.fold_in(keys, 3)                   # sites 2 and 3 of the model
    assert np.array_equal(ch["a"].numpy(), O.exp(O.normal.sample(ka, np.float32(0.25), np.float32(0.5))))
    assert np.array_equal(ch["h"].numpy(), np.abs(O.normal.sample(kh, np.float32(0.0), np.float32(1.0)) * np.float32(2.0)))
    a64, h64 = ch["a"].numpy().astype(np.float64), ch["h"].numpy().astype(np.float64)
    assert np.abs(tr.get_subtrace("a").get_score().numpy() - stats.lognorm.logpdf(a64, 0.5, scale=np.exp(0.25))).max() < 2e-5
    assert np.abs(tr.get_subtrace("h").get_score().numpy() - stats.halfnorm.logpdf(h64, scale=2.0)).max() < 2e-5
    # exact_density: user-defined sampler / density traced into the program
    mine = genjax.exact_density(lambda key, loc: genjax.normal.sym_sample(key, (loc, 1.0)),
                                lambda v, loc: -0.5 * jnp.square(v - loc) - 0.9189385, "mynormal")

    @genjax.gen
    def m2():
        return mine(1.0) @ "z"
    t2 = m2.simulate(genjax.split(genjax.key(1), 16), ())
    z = t2.get_choices()["z"].numpy().astype(np.float64)
    assert np.abs(t2.get_score().numpy() - stats.norm.logpdf(z, 1.0, 1.0)).max() < 1e-5
    with pytest.raises(NotImplementedError):
        genjax.tfp_distribution(None)


def test_categorical_sample_shape():
    """`categorical(probs=..., sample_shape=n)` (the mixture notebook's `generate_datapoints`): n draws at one
    site, draw j / category k on gumbel counter j*K + k; larger n is refused, not silently ignored."""
    probs = [0.2, 0.5, 0.3]

    @genjax.gen
    def m():
        return genjax.categorical(probs=probs, sample_shape=5) @ "idx"
    n = 2000
    tr = m.simulate(genjax.split(genjax.key(2), n), ())
    idx = tr.get_choices()["idx"].numpy()
    keys = np.ascontiguousarray(O.fold_in(O.split(O.key(2), n), 1))
    logits = O.log(np.asarray(probs, np.float32))
    lb = np.ascontiguousarray(np.broadcast_to(logits, (n, 3)))
    want = np.empty((n, 5), np.int32)
    for j in range(5):
        o, ctr = np.empty(n, np.int32), np.full(n, j * 3, np.uint64)
        O.lib().orc_categorical_sample(O.I64(n), O.I64(3), O._p(keys), O.I64(1), O._p(lb), O.I64(3), O._p(ctr), O.I64(1), O._p(o))
        want[:, j] = o
    assert np.array_equal(idx, want)
    assert np.allclose(tr.get_score().numpy(), (logits[idx] - O.logsumexp(logits)).sum(1), atol=1e-6)

    @genjax.gen
    def big():
        return genjax.categorical(probs=probs, sample_shape=5000) @ "idx"
    # ONE trace: the 5000 draws run on the launch axis (sitewise.vector_site); under a batch of keys they are ONE counted
    # loop per particle (distributions._Categorical.loop_site), draw j on the counters j * K .. j * K + K - 1 all the same
    assert tuple(big.simulate(genjax.key(0), ()).get_retval().shape) == (5000,)
    trb = big.simulate(genjax.split(genjax.key(0), 4), ())

    @O.gen
    def obig():
        return O.categorical(logits=O.log(np.asarray(probs, np.float32)), sample_shape=5000) @ "idx"
    otrb = obig.simulate(O.split(O.key(0), 4), ())
    assert np.array_equal(trb.get_choices()["idx"].numpy(), otrb.get_choices()["idx"])
    assert np.array_equal(trb.get_score().numpy(), np.asarray(otrb.get_score(), np.float32))


def test_mixture_model_gibbs_end_to_end():
    """The Dirichlet-mixture application (7_application_dirichlet_mixture_model.ipynb, cells 6-10) through this
    package: cluster means by a Vmap'd conjugate draw, assignments by gibbs_categorical, weights by a
    dirichlet draw.  Recovers well-separated clusters in a few sweeps."""
    from genjax_amd import workloads
    from genjax_amd.inference.gibbs import gibbs_categorical
    K, N, PRIOR_MEAN, PRIOR_VAR, OBS_VAR, ALPHA = 6, 3000, 0.0, 100.0, 1.0, 1.0
    rng = np.random.default_rng(3)
    true_means = (6.0 * (np.arange(K) - (K - 1) / 2.0)).astype(np.float32)
    z = rng.integers(0, K, N)
    x = torch.from_numpy((true_means[z] + rng.standard_normal(N)).astype(np.float32))

    @genjax.gen
    def generate_cluster(mean, var):
        return genjax.normal(mean, var) @ "mean"

    @genjax.gen
    def generate_cluster_weight(alphas):
        return genjax.dirichlet(alphas) @ "probs"
    generate_datapoint = workloads.make_mixture(genjax, obs_scale=OBS_VAR)
    key = genjax.key(32421)
    means = torch.from_numpy(rng.normal(0.0, 8.0, K).astype(np.float32))
    probs = torch.full((K,), 1.0 / K)
    idx = torch.from_numpy(rng.integers(0, K, N).astype(np.int32))
    for _ in range(12):
        # -- cluster means: conjugate Normal update, one Vmap'd draw (update_cluster_means) --
        counts = torch.bincount(idx.long(), minlength=K).float()
        sums = torch.zeros(K).index_add_(0, idx.long(), x)
        cm = sums / counts
        post_mean = PRIOR_VAR / (PRIOR_VAR + OBS_VAR / counts) * cm + (OBS_VAR / counts) / (PRIOR_VAR + OBS_VAR / counts) * PRIOR_MEAN
        post_var = 1.0 / (1.0 / PRIOR_VAR + counts / OBS_VAR)
        key, sub = genjax.split(key)
        ok = counts > 0
        drawn = generate_cluster.vmap().simulate(sub, (jnp.array(torch.where(ok, post_mean, means).numpy()),
                                                       jnp.array(torch.where(ok, post_var, torch.ones(K)).numpy()))
                                                 ).get_choices()["mean"]
        means = torch.where(ok, drawn.reshape(-1), means)
        # -- assignments: enumerative Gibbs in ONE launch (update_datapoint_assignment) --
        key, sub = genjax.split(key)
        idx = gibbs_categorical(sub, generate_datapoint, (probs, means), C["obs"].set(x), "idx", K, batch_shape=(N,))
        # -- weights: Dirichlet conjugate update (update_cluster_weights) --
        key, sub = genjax.split(key)
        new_alpha = ALPHA / K + torch.bincount(idx.long(), minlength=K).float()
        probs = generate_cluster_weight.simulate(sub, (jnp.array(new_alpha.numpy()),)).get_retval().reshape(-1)
    found = np.sort(means.numpy())
    big = np.sort(means.numpy()[torch.bincount(idx.long(), minlength=K).numpy() > N // (4 * K)])
    # every well-populated cluster sits on a true mean
    assert all(np.min(np.abs(true_means - m)) < 0.5 for m in big), (found, true_means)
    assert len(big) >= K - 2
    assert abs(float(probs.sum()) - 1.0) < 1e-5


def test_vector_state_sweep_matches_oracle():
    from tests import parity
    parity.check_vector_state_sweep(n=2000, T=5)


def test_numpy_namespace_compositions():
    """genjax_amd.numpy: the composed functions (log2, sinh, logaddexp, mod, max/min/prod/dot/cumsum/std over
    traced vectors, ...) evaluated in one launch (engine.elementwise) against numpy float64."""
    from genjax_amd.engine import elementwise
    rng = np.random.default_rng(0)
    x = rng.uniform(0.5, 2.0, 100).astype(np.float32)
    y = rng.uniform(0.5, 2.0, 100).astype(np.float32)
    cases = [
        (lambda a, b: jnp.log2(a) + jnp.log10(b) + jnp.exp2(a * 0.5), lambda a, b: np.log2(a) + np.log10(b) + np.exp2(a * 0.5)),
        (lambda a, b: jnp.sinh(a) - jnp.cosh(b) + jnp.tan(a * 0.3), lambda a, b: np.sinh(a) - np.cosh(b) + np.tan(a * 0.3)),
        (lambda a, b: jnp.logaddexp(a, b) + jnp.sign(a - b) + jnp.mod(a * 3.0, b),
         lambda a, b: np.logaddexp(a, b) + np.sign(a - b) + np.mod(a * 3.0, b)),
        (lambda a, b: jnp.max(jnp.stack([a, b, a * b])) + jnp.min(jnp.stack([a, b])) + jnp.prod(jnp.stack([a, b]))
         + jnp.dot(jnp.stack([a, b]), jnp.stack([b, a])) + jnp.cumsum(jnp.stack([a, b, a]))[2] + jnp.std(jnp.stack([a, b, a + b])),
         lambda a, b: np.maximum(np.maximum(a, b), a * b) + np.minimum(a, b) + a * b + 2 * a * b + (2 * a + b)
         + np.std(np.stack([a, b, a + b]), axis=0)),
        (lambda a, b: jnp.expm1(a) + jnp.round(b * 3.0) + jnp.nan_to_num((b - b) / (b - b), nan=0.5) * 0.0 + jnp.clip(a, 0.8, 1.2),
         lambda a, b: np.expm1(a) + np.rint(b * 3.0) + np.clip(a, 0.8, 1.2)),
    ]
    a64, b64 = x.astype(np.float64), y.astype(np.float64)
    for fn, ref in cases:
        got = elementwise(fn, torch.from_numpy(x), torch.from_numpy(y)).numpy()
        want = ref(a64, b64)
        assert np.max(np.abs(got - want) / (1 + np.abs(want))) < 2e-6


_CAPTURED_SCALE = 1.0


def test_trace_cache_follows_captured_values():
    """a model reads `scale` from its enclosing scope / a module global; the reference re-traces on every
    call, so changing the captured value changes the result — the program cache must not replay the old trace."""
    global _CAPTURED_SCALE
    scale = 1.0

    @genjax.gen
    def model():
        return genjax.normal(0.0, scale) @ "x"
    a = f(model.assess(C.kw(x=0.5), ())[0])
    scale = 3.0
    b = f(model.assess(C.kw(x=0.5), ())[0])
    from scipy import stats
    assert a == pytest.approx(stats.norm.logpdf(0.5, 0.0, 1.0), abs=1e-6)
    assert b == pytest.approx(stats.norm.logpdf(0.5, 0.0, 3.0), abs=1e-6)

    @genjax.gen
    def inner():
        return genjax.normal(0.0, _CAPTURED_SCALE) @ "z"

    @genjax.gen
    def outer():
        return inner() @ "sub"
    _CAPTURED_SCALE = 1.0
    c = f(outer.assess(C["sub", "z"].set(0.5), ())[0])
    _CAPTURED_SCALE = 2.0
    d = f(outer.assess(C["sub", "z"].set(0.5), ())[0])          # through a called generative function
    assert c == pytest.approx(stats.norm.logpdf(0.5, 0.0, 1.0), abs=1e-6)
    assert d == pytest.approx(stats.norm.logpdf(0.5, 0.0, 2.0), abs=1e-6)
    # ... and an unchanged capture still hits the cache
    from genjax_amd import static
    n0 = len(static._CACHE)
    model.assess(C.kw(x=0.25), ())
    assert len(static._CACHE) == n0
    # the fingerprint is memoised per function and re-validated slot by slot (identity / version /
    # small contents), not re-walked: it must still notice a small array edited in place, a large array edited in place
    # (a sampled hash: elements on the sampling grid) and a rebound helper function
    small = np.array([1.0, 2.0], np.float32)
    big = np.ones(100_000, np.float32)

    def helper(x):
        return x * 2.0

    @genjax.gen
    def m2():
        return genjax.normal(0.0, float(small[1]) * float(big[0])) @ "x", helper

    def s2():
        return f(m2.assess(C.kw(x=0.5), ())[0])
    assert s2() == pytest.approx(stats.norm.logpdf(0.5, 0.0, 2.0), abs=1e-6)
    k0 = static._gfkey(m2)
    assert static._gfkey(m2) == k0 and m2.__dict__["_gmx_fp_memo"][3] == k0[2]
    small[1] = 4.0
    assert s2() == pytest.approx(stats.norm.logpdf(0.5, 0.0, 4.0), abs=1e-6)
    big[0] = 0.5
    assert s2() == pytest.approx(stats.norm.logpdf(0.5, 0.0, 2.0), abs=1e-6)
    k1 = static._gfkey(m2)

    def helper(x):          # noqa: F811  (the cell now holds another function object)
        return x * 3.0
    assert static._gfkey(m2) != k1


def test_eager_numpy_namespace_uses_the_device_math():
    """`jnp.exp(t)` on a tensor OUTSIDE a traced model runs the same fixed-sequence functions a model does (one
    launch through engine.elementwise), not torch's: bit-identical to the oracle's restatement."""
    x = torch.from_numpy(np.linspace(-20.0, 20.0, 257, dtype=np.float32))
    pos = torch.from_numpy(np.geomspace(1e-30, 1e30, 257).astype(np.float32))
    assert np.array_equal(jnp.exp(x).numpy(), O.exp(x.numpy()))
    assert np.array_equal(jnp.log(pos).numpy(), O.log(pos.numpy()))
    assert np.array_equal(jnp.log1p(pos).numpy(), O.log1p(pos.numpy()))
    assert tuple(jnp.exp(x.reshape(257, 1)).shape) == (257, 1)
    assert f(jnp.exp(torch.tensor(1.0))) == pytest.approx(math.e, rel=1e-6)


def test_vmap_in_axes_and_broadcast_marker():
    """transforms.vmap maps: in_axes None marks tensors launch-uniform (a vector as long as the batch is NOT taken for
    per-particle data), in_axes k != 0 moves the mapped axis to the front, out_axes moves the result axis."""
    n = 20
    w = torch.linspace(0.1, 2.0, n)                      # length == batch size: ambiguous without the marker
    keys = genjax.split(genjax.key(1), n)

    @genjax.gen
    def m(scales):
        return genjax.normal(0.0, scales[3]) @ "x"
    x = genjax.vmap(lambda k, s: m.simulate(k, (s,)).get_retval(), in_axes=(0, None))(keys, w)
    okeys = O.fold_in(O.split(O.key(1), n), 1)
    assert np.array_equal(x.numpy(), O.normal.sample(okeys, np.float32(0.0), np.float32(w[3].item())))
    # the shared argument is a tensor like any other: the mapped function computes with it before the generative
    # function sees it (`w * 2.0` used to raise on the marker type), and the results stay launch-uniform
    x2 = genjax.vmap(lambda k, s: m.simulate(k, (s * 2.0 + 1.0,)).get_retval(), in_axes=(0, None))(keys, w)
    assert np.array_equal(x2.numpy(), O.normal.sample(okeys, np.float32(0.0), np.float32((w * 2.0 + 1.0)[3].item())))
    from genjax_amd.engine import Broadcast
    b = Broadcast(w)
    assert isinstance(b * 2.0, Broadcast) and isinstance(torch.exp(b)[2:], Broadcast) and isinstance(b.reshape(4, 5).t(), Broadcast)
    mixed = b * torch.ones(n)                         # with per-instance data: an ordinary tensor
    assert not isinstance(mixed, Broadcast) and torch.equal(mixed, w)
    assert not isinstance(b.plain, Broadcast) and float(b.sum()) == pytest.approx(float(w.sum()))
    locs = torch.arange(3 * n, dtype=torch.float32).reshape(3, n)

    @genjax.gen
    def m2(loc):
        return genjax.normal(loc[1], 1.0) @ "x"
    y = genjax.vmap(lambda k, l: m2.simulate(k, (l,)).get_retval(), in_axes=(0, 1))(keys, locs)
    assert np.array_equal(y.numpy(), O.normal.sample(okeys, locs[1].numpy(), np.float32(1.0)))
    z = genjax.vmap(lambda k, l: l * 2.0, in_axes=(0, 1), out_axes=1)(keys, locs)
    assert tuple(z.shape) == (3, n)
    with pytest.raises(ValueError):
        genjax.vmap(lambda k, l: l, in_axes=(0, 0))(keys, locs)          # 20 keys vs a leading axis of 3


def test_program_limits():
    """<= 64 live 32-bit values per particle in ONE launch (<= 32 for the interpreter, up to 64 for specialised
    kernels); a model that needs more is cut into a chain of launches with the values in flight spilled to scratch
    leaves (program.split_graph) — same nodes, same order: bit-identical to the oracle, which has no such limit."""
    @genjax.gen
    def chain():
        acc = genjax.normal(0.0, 1.0) @ "x0"
        for i in range(1, 30):
            acc = acc + genjax.normal(acc, 1.0) @ f"x{i}"      # 30 sites (60 stored leaves), few live values
        return acc
    tr = chain.simulate(genjax.key(0), ())
    assert len(tr.get_choices().addresses()) == 30

    def mk(g, lit, exp, sin):
        @g.gen
        def too_wide(v):
            xs = [exp(v * lit(float(i) / 64.0)) * (sin(v + lit(float(i))) + lit(1.5)) for i in range(80)]   # 80 values, all live at once
            acc = g.normal(xs[0], lit(1.0)) @ "x"
            for x in xs[1:]:
                acc = acc * x
            acc2 = acc
            for x in xs:
                acc2 = acc2 + x * acc
            return acc2
        return too_wide
    v = np.linspace(-0.05, 0.05, 4).astype(np.float32)
    tr = mk(genjax, float, jnp.exp, jnp.sin).simulate(genjax.split(genjax.key(0), 4), (torch.from_numpy(v),))
    otr = mk(O, np.float32, O.exp, O.sin).simulate(O.split(O.key(0), 4), (v,))
    assert np.array_equal(tr.get_retval().numpy(), otr.get_retval()) and np.all(np.isfinite(otr.get_retval()))
    assert np.array_equal(tr.get_choices()["x"].numpy(), otr.get_choices()["x"])


def test_models_of_more_sites_than_one_launch_holds():
    """ref static.py:254-380 (the handlers walk any number of sites): 32, 40 and 200 sites — 2 stored leaves each, 64 per
    launch — as chains of launches, bit for bit against the oracle (simulate / importance / assess / update /
    regenerate / StaticRequest of Rejuvenate moves)"""
    from tests import parity
    for ns in (32, 40, 200):
        parity.check_many_sites(ns=ns)
    parity.check_many_sites(ns=67, B=130, seed=8, kinds=("normal", "flip", "normal", "uniform"))


def test_three_combinator_levels_at_loop_sizes():
    """ref vmap.py:180-218 nests freely: vmap(vmap(vmap(elem))) over 20 x 20 x 20 (three counted loops), and mixed
    sizes where a small plate sits between two loops — against the oracle"""
    from tests import parity
    parity.check_three_nested_plates()
    parity.check_three_nested_plates(dims=(3, 20, 17), B=9, seed=4)
    parity.check_three_nested_plates(dims=(18, 2, 33), B=3, seed=5)


def test_long_vector_valued_sites_under_a_particle_batch():
    """ref tensorflow_probability/__init__.py:52-62 + distribution.py:383-396: `normal(a * xs + b, sigma) @ "y"` over 40,
    500 and 5 000 observations under a batch of particles — one counted loop per particle, bit for bit against the
    oracle (simulate / ImportanceK / importance / assess / update; normal, flip and uniform sites)"""
    from tests import parity
    parity.check_long_vector_sites(n=500, K=64)
    parity.check_long_vector_sites(n=40, K=9, seed=3)
    parity.check_long_vector_sites(n=5000, K=17, seed=5)


def test_index_request_on_a_long_scan_is_o1():
    """ref scan.py:325-416 `edit_index`: slice t edited, slice t + 1 re-visited, nothing else — chains of edits on a
    40-step scan against the oracle and against the counted-loop form; a running-sum kernel is refused statically"""
    from tests import parity
    parity.check_scan_index_request_o1()
    parity.check_scan_index_request_o1(n=9, T=70, seed=3, edits=40)      # (longer than PATCH_DEPTH_MAX: the lazy leaves are folded)


def test_nested_index_request_on_a_plate_of_long_scans_is_o1():
    """VERDICT r5 item 8 (ref vmap.py:277-332 around scan.py:325-416): `IndexRequest(j, IndexRequest(t, sub))` on
    `kernel.scan(n=T).vmap()` — Python-int and per-particle indices at either level, chains of edits past
    PATCH_DEPTH_MAX — against the oracle and against the counted-loop form; a directly nested scan.vmap() under Update"""
    from tests import parity
    parity.check_plate_of_scans_index_request_o1()
    parity.check_plate_of_scans_index_request_o1(n=9, J=17, T=70, seed=5, edits=40)
    for seed in range(24):          # the nest written directly, random sizes around the unroll limits, every GFI method
        parity.check_direct_plate_of_scans_random(seed)
    # ... and a PLATE inside the step of a long scan: O(1) steps against the counted-loop form, chains of 9 / 40 edits
    parity.check_scan_of_plates_index_request_o1()
    parity.check_scan_of_plates_index_request_o1(n=5, T=70, P=20, seed=8, edits=40)


def test_gather_of_a_latent_vector_at_a_table_of_group_indices():
    """`normal(theta[group], s) @ "y"`: the gather of a latent vector (values in memory) at a long table of indices is a
    recipe evaluated in the consuming site's loop (engine.StepInput.__getitem__) — against the oracle and against the
    unrolled form, J = 40 x N = 300 and J = 200 x N = 5 000"""
    from tests import parity
    parity.check_gather_at_group_indices()
    parity.check_gather_at_group_indices(J=200, N=5000, K=4, seed=8)


def test_exponential_is_tfd_exponential():
    """ref tensorflow_probability/__init__.py:150 `exponential = tfp_distribution(tfd.Exponential)` (choice_maps.ipynb c18):
    -log(U) / rate with U on [tiny, 1) and log rate - rate x — the law against scipy (KS), the density to 2e-6, a rate that is
    itself a latent"""
    from scipy import stats
    import genjax_amd as G

    @G.gen
    def m(r):
        x = G.exponential(r) @ "x"
        G.exponential(x + 0.5) @ "y"
        return x
    n = 100_000
    tr = m.simulate(G.split(G.key(1), n), (2.0,))
    x = tr.get_choices()["x"].cpu().numpy().astype(np.float64)
    y = tr.get_choices()["y"].cpu().numpy().astype(np.float64)
    assert stats.kstest(x, "expon", args=(0, 0.5)).pvalue > 1e-3 and (x > 0).all()
    s, _ = m.assess(tr.get_choices(), (2.0,))
    want = stats.expon.logpdf(x, scale=0.5) + stats.expon.logpdf(y, scale=1 / (x + 0.5))
    assert np.abs(s.cpu().numpy() - want).max() < 2e-5 * np.abs(want).max()
    w = G.exponential.assess(G.ChoiceMap.choice(-1.0), (2.0,))[0]
    assert float(w) == float("-inf")


def test_empty_and_single_particle_batches():
    """jax.vmap over zero keys gives empty arrays, not an error; one particle is just a batch of one"""
    @genjax.gen
    def m():
        x = genjax.normal(0.0, 1.0) @ "x"
        _ = genjax.normal(x, 1.0) @ "y"
        return x
    ks = genjax.split(genjax.key(0), 0)
    tr, w = m.importance(ks, C.kw(y=1.0), ())
    assert w.shape == (0,) and tr.get_score().shape == (0,) and tr.get_choices()["x"].shape == (0,)
    assert m.simulate(ks, ()).get_retval().shape == (0,)
    s, _ = m.assess(C.kw(x=torch.zeros(0), y=torch.zeros(0)), ())
    assert s.shape == (0,)
    k1 = genjax.split(genjax.key(0), 1)
    tr1, w1 = m.importance(k1, C.kw(y=1.0), ())
    otr, ow = _oracle_two_site().importance(O.split(O.key(0), 1), O.C.kw(y=np.float32(1.0)), ())
    assert np.array_equal(w1.numpy(), ow) and np.array_equal(tr1.get_choices()["x"].numpy(), otr.get_choices()["x"])


def _oracle_two_site():
    @O.gen
    def m():
        x = O.normal(0.0, 1.0) @ "x"
        _ = O.normal(x, 1.0) @ "y"
        return x
    return m


def test_sharded_requires_tile_aligned_shards():
    """shards must start on a 1024-particle tile of the global CDF, or the result would depend on the rank count"""
    from genjax_amd import workloads
    from genjax_amd.inference.sharded import ShardedBootstrapSweep

    class _Two:
        @staticmethod
        def get_rank(): return 0
        @staticmethod
        def get_world_size(): return 2
    init, step = workloads.make_lgssm(genjax)
    with pytest.raises(ValueError, match="multiple of 1024"):
        ShardedBootstrapSweep(init, step, 1000, 3, _Two)
    ShardedBootstrapSweep(init, step, 2048, 3, _Two)          # aligned: constructs (no collective until prepare)


def test_evidence_is_minus_inf_when_no_particle_has_mass():
    """an observation impossible under every particle: resampling falls back to the last particle and the
    log-ML estimate is -inf (not an exception)"""
    from genjax_amd.inference import smc

    @genjax.gen
    def m():
        x = genjax.uniform(0.0, 1.0) @ "x"
        _ = genjax.uniform(x, x + 1.0) @ "y"
        return x
    coll = smc.ImportanceK(genjax.Target(m, (), C.kw(y=5.0)), k_particles=64).run_smc(genjax.key(0))
    assert np.all(np.isneginf(coll.get_log_weights().numpy()))
    res = smc.resample(genjax.key(1), coll, "systematic")
    assert np.all(res.ancestors.numpy() == 63)
    assert res.log_ml_offset.value() == -math.inf


def test_sweep_without_program_written_tile_stats(monkeypatch):
    """GENMI_HOSTSIM_TILE_STATS=0: the C-ABI mirror behaves like the interpreter (256-particle groups, no tile
    statistics from the site program), so BootstrapSweep takes the gmx_resample path; same sweep bit for bit."""
    from tests import parity
    monkeypatch.setenv("GENMI_HOSTSIM_TILE_STATS", "0")
    res = parity.check_lgssm_sweep(n=3000, T=5)
    assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0
    assert res["log_ml"] == res["log_ml_oracle"]


def test_tuple_state_sweep_matches_oracle():
    from tests import parity
    parity.check_tuple_state_sweep()


def test_noise_ahead_sweep_matches_oracle():
    """BootstrapSweep(noise_ahead=True): the steps' normal draws come from background programs (static.NoiseProgram),
    the site programs read them (MinimalGenerate(hoist_noise=True)); same particles, weights, ancestors and evidence
    as the oracle's sweep — T not a multiple of the noise group, one / three latent sites per step; the draws of a
    group's steps come from ONE launch per key (rows of keys, GMX_KEY_ROWSPLIT)."""
    from tests import parity
    res = parity.check_lgssm_sweep(n=3000, T=23, noise_ahead=True)
    assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0
    # the same integer totals; 23 float64 terms summed pairwise (numpy) here and one after the other in the oracle
    assert abs(res["log_ml"] - res["log_ml_oracle"]) < 1e-11
    parity.check_tuple_state_sweep(noise_ahead=True)
    # with an MH move per step (config 3): the move's proposal + accept draws and the extension's draw, from two keys
    parity.check_nlssm_mh_sweep(n=1500, T=7, noise_ahead=True)
    # stratified: the resampler's per-slot uniforms come from the background stream too (gmx_slot_uniforms ->
    # gmx_resample_tiles_u; the CPU mirror refuses uniforms that are not the resampling key's own)
    for na in (True, False):       # (the one-stream form draws them inside the resampler)
        res = parity.check_lgssm_sweep(n=3000, T=23, noise_ahead=na, resample="stratified")
        assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0


def test_noise_hoist_takes_launch_keyed_draws():
    """engine.NoiseHoist: a `normal` / `uniform` site's draw is hoisted when its key is a fold_in chain from the
    particle key (nested calls included: the chain has one counter per level); constrained sites draw nothing.
    The hoisted program + NoiseProgram reproduce the plain program's outputs bit for bit."""
    import genjax_amd as G
    from genjax_amd.core.choice_map import ChoiceMap
    from genjax_amd.random import lazy_split
    from genjax_amd.static import MinimalGenerate, NoiseProgram

    @G.gen
    def inner(m):
        a = G.normal(m, 2.0) @ "a"
        u = G.uniform(-0.5, m) @ "u"
        return a + u

    @G.gen
    def model(x0):
        z = G.normal(x0, 1.0) @ "z"
        s = inner(z) @ "sub"
        G.normal(s, 0.5) @ "y"
        return s

    n = 1500
    x0 = torch.linspace(-1, 1, n)
    obs = ChoiceMap.empty().set("y", torch.tensor(0.25))
    key = G.key(11)
    outs = []
    for hoist in (False, True):
        p = MinimalGenerate(model, (x0,), obs, (n,), hoist_noise=hoist)
        noise = []
        if hoist:
            # "z": site 1; "sub" is site 2 -> "a" site 1, "u" site 2 inside it
            assert p.noise == (("LDKEY", (1,), 0, "normal"), ("LDKEY", (2, 1), 0, "normal"), ("LDKEY", (2, 2), 0, "uniform"))
            q = NoiseProgram(p.noise, (n,))
            zs = [torch.zeros((1, n)) for _ in p.noise]
            q.run((n,), lazy_split(key, n), zs)
            noise = [z.reshape(n) for z in zs]
        else:
            assert p.noise == ()
        r, w = torch.zeros((1, n)), torch.zeros((1, n))
        part = torch.zeros((2, (n + 255) // 256))
        bufs = [None] * len(p.comp.outputs)
        bufs[p.ro[1]], bufs[p.wo[1]] = r, w
        p.comp.run(p.leaves((x0,), obs, noise), (n,), lazy_split(key, n), red_out=part, out_buffers=bufs)
        outs.append((r, w))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_vector_state_mh_sweep_matches_oracle():
    """BootstrapSweep(rejuvenate=...) with a 2-vector state held in one vector-valued site: the fused MH move
    gathers, proposes, accepts and selects all components; bit-exact vs the oracle incl. the accept bits"""
    from tests import parity
    res = parity.check_vector_mh_sweep(n=1500, T=5)
    assert 0.3 < res["accept_rate"] < 1.0


def test_weight_fixed_matches_its_definition(tmp_path):
    """csrc/gmx_math.h gmx_exp_fixed (the integer form the kernels use for the CDF's fixed-point weights) equals
    floor(gmx_expf(d) * 2^shift) — the definition the oracle restates — on a strided sweep of ALL float bit patterns
    of d (NaNs, infinities, denormals included) for shifts 1..62; tools/check_exp_fixed.c with STRIDE=1 is the
    exhaustive form (run once per change of either function; last run: all 2^32 patterns x 8 shifts, no mismatch)."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "check_exp_fixed"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-DSTRIDE=1021",
                           "-I", os.path.join(ROOT, "genjax_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "check_exp_fixed.c"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "shift 62 done, mismatches so far 0" in out.stdout


def test_sorted_multinomial_past_two_million_particles():
    """multinomial_sorted for n > 2^21 (tile prefixes + chunked table offsets): bit-exact vs the oracle"""
    from tests import parity
    assert parity.check_multinomial_sorted_big() > 100_000


def test_sweep_with_the_resampler_in_the_next_step_launch():
    """BootstrapSweep(fuse_resample=True): ONE launch per step — the program that gathers the resampled state first
    resamples the previous step itself (gmx_run_args.rs: tagged ancestors; the C-ABI mirror resamples, tags and gathers
    through the masked indices) — the same sweep bit for bit as the two-launch form and the oracle's: ragged sizes, one
    tile, noise ahead, and with the chained MH move as the launch that resamples."""
    from tests import parity
    for kw in (dict(n=3000, T=5), dict(n=1024, T=3), dict(n=700, T=4), dict(n=3000, T=12, noise_ahead=True)):
        res = parity.check_lgssm_sweep(specialize=True, fuse_resample=True, **kw)
        assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0, kw
        assert abs(res["log_ml"] - res["log_ml_oracle"]) < 1e-11
    parity.check_nlssm_mh_sweep(n=1500, T=5, specialize=True, fuse_resample=True, want_chained=True)
    parity.check_nlssm_mh_sweep(n=1500, T=5, specialize=True, noise_ahead=True, fuse_resample=True)


@pytest.mark.parametrize("chained", [False, True])
def test_mh_sweep_with_the_move_chained_into_the_extension(chained):
    """BootstrapSweep(rejuvenate=..., chain_mh=): the MH move and the extension that follows it as one program
    (static.MinimalMHGenerate, the extension's key through OP_KSPLITU) or as two launches (what a chained program too
    large for the tile statistics falls back to) — the same sweep, equal to the oracle's step-by-step statement either
    way (scalar and vector state)."""
    from tests import parity
    parity.check_nlssm_mh_sweep(n=1500, T=5, want_chained=chained, chain_mh=chained)
    res = parity.check_vector_mh_sweep(n=1200, T=4, chain_mh=chained)
    assert 0.3 < res["accept_rate"] < 1.0


def test_long_scan_with_a_vector_valued_site():
    """a counted-loop scan whose kernel has a vector-valued site (a 2-D latent state): values come back [n, T, 2]"""
    from tests import parity
    parity.check_scan_long_vector_site()


def test_long_scan_update_and_regenerate():
    """Scan.edit (Update / Regenerate) of a 40-step scan as a counted loop == the oracle's step-by-step edits"""
    from tests import parity
    parity.check_scan_long_edits()


def test_long_scan_vector_sites_constraints_and_edits():
    """long scan with vector-valued latent AND observation sites: [T, 2] table constraints, [n, T, 2] choices, edits"""
    from tests import parity
    parity.check_scan_long_vector_constraints()


@pytest.mark.parametrize("no,T", [(3, 40), (12, 20), (17, 17), (40, 24)])
def test_plate_of_long_scans_matches_oracle(hostsim, no, T):
    """`series.vmap()` where every element runs a long scan (ref: combinators nest freely, vmap.py:180-218 over
    scan.py:200-294).  no <= 4: the plate is unrolled around the elements' loops (their [T, n] outputs are stacked
    after the launch); more: the plate runs as a counted loop AROUND the scan's loop — two nested loops in one site
    program (OP_LOOP two deep, GMX_F_FLAT leaves [n, no, T]).  no == T: the square case an axis picked by shape
    would get wrong."""
    from tests import parity
    parity.check_plate_of_scans(n=33, no=no, T=T)


def test_nested_combinators_match_oracle(hostsim):
    """plate of plates, scan of plate, scan of scan — both levels long: two nested counted loops"""
    from tests import parity
    parity.check_nested_combinators(n=21)
    parity.check_nested_constraint_forms()
    parity.check_nested_edge_cases()


def test_two_stage_multinomial_matches_oracle(hostsim):
    """gmx_multinomial_tiled == the oracle's definition (C-ABI mirror), and a whole sweep resampled with it"""
    from tests import parity
    for kw in (dict(n=5000), dict(n=1024, seed=6), dict(n=3333, seed=7, spike=30.0), dict(n=2500, seed=8, dead=True),
               dict(n=1, seed=9), dict(n=1025, seed=10, sigma=8.0)):
        parity.check_multinomial_tiled(**kw)
    for na in (False, True):
        res = parity.check_lgssm_sweep(n=3000, T=7, resample="multinomial_tiled", noise_ahead=na)
        assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0


def test_resampling_kinds_say_what_they_do_not_take(hostsim):
    """the sharded router takes the ordered schemes it was built for; the tile / sorted multinomials want n_out = n"""
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference import sharded, smc
    init, step = workloads.make_lgssm(G)
    for kind in ("multinomial", "multinomial_tiled"):
        with pytest.raises(NotImplementedError, match="ORDERED schemes"):
            sharded.ShardedBootstrapSweep(init, step, 1024, 3, dist=None, resample=kind)
    coll = smc.ImportanceK(G.Target(init, (), G.ChoiceMap.kw(y=0.3)), k_particles=64).run_smc(G.key(1))
    for kind in ("multinomial_tiled", "multinomial_sorted"):
        with pytest.raises(NotImplementedError, match="n_out = n"):
            smc.resample(G.key(2), coll, kind, n_out=32)
    assert smc.resample(G.key(2), coll, "multinomial", n_out=32).ancestors.numel() == 32


def test_fused_resampling_beyond_2048_tiles(hostsim):
    """n > 2^21: tile statistics -> tile prefixes -> the prefix-reading resampler == CDF array + search == the oracle"""
    from tests import parity
    parity.check_resample_beyond_2048_tiles()


def test_sorted_multinomial_matches_oracle(hostsim):
    """gmx_sorted_uniforms / gmx_resample_sorted == the oracle's definition (C-ABI mirror), and whole sweeps resampled with it"""
    from tests import parity
    for kw in (dict(n=5000, rows=3), dict(n=1024, seed=6), dict(n=3333, seed=7, spike=30.0), dict(n=2500, seed=8, dead=True),
               dict(n=1, seed=9), dict(n=2, seed=12), dict(n=1025, seed=10, sigma=8.0), dict(n=40_000, seed=11, sigma=3.0)):
        parity.check_multinomial_sorted(**kw)
    for na in (False, True):
        res = parity.check_lgssm_sweep(n=3000, T=7, resample="multinomial_sorted", noise_ahead=na)
        assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"] and res["lw_max_abs_diff"] == 0.0


@pytest.mark.parametrize("kind", ["systematic", "stratified", "multinomial", "multinomial_tiled", "multinomial_sorted"])
def test_evidence_estimate_is_unbiased(hostsim, kind):
    """closed-form evidence of the resampling definitions (Kalman), independent of the oracle"""
    from tests import parity
    parity.check_evidence_unbiased(kind, R=2500)


def test_sampler_laws_against_scipy(hostsim):
    """the product's samplers against scipy's distributions (independent of the oracle)"""
    from tests import parity
    parity.check_sampler_laws(n=100_000)


@pytest.mark.parametrize("kind", ["systematic", "stratified", "multinomial", "multinomial_tiled", "multinomial_sorted"])
def test_offspring_laws(hostsim, kind):
    """E[offspring_i] = n w_i for every scheme; the multinomial variance; |offspring - n w| < 1 (systematic) / 2 (stratified)"""
    from tests import parity
    parity.check_offspring_laws(kind)


@pytest.mark.parametrize("A,T", [(3, 20), (12, 20), (20, 24)])
def test_update_through_a_plate_of_long_scans(hostsim, A, T):
    from tests import parity
    parity.check_nested_edits(A, T)


@pytest.mark.parametrize("A,T", [(20, 24), (3, 24)])
def test_index_requests_through_nested_loops(hostsim, A, T):
    """IndexRequest into a plate of long scans (one series; one step of one series) and into a scan of plates"""
    from tests import parity
    parity.check_nested_index_edits(A, T)


def test_importancek_evidence_is_unbiased(hostsim):
    """ImportanceK against a conjugate closed form (independent of the oracle)"""
    from tests import parity
    parity.check_importance_unbiased()


def test_evidence_estimate_is_unbiased_with_mh_moves(hostsim):
    """resample-move SMC: a valid MH move after resampling leaves the evidence estimate unbiased (Kalman closed form)"""
    from tests import parity
    parity.check_evidence_unbiased("systematic", R=3000, T=6, mh=True, seed0=900000)


def test_marginal_density_estimates_are_unbiased(hostsim):
    from tests import parity
    parity.check_marginal_density_unbiased()


def test_long_scan_importance_weights_against_kalman(hostsim):
    from tests import parity
    parity.check_scan_importance_vs_kalman()


def test_hmc_move_leaves_the_posterior_invariant(hostsim):
    from tests import parity
    parity.check_hmc_invariance()
    out = parity.check_hmc_invariance(n=50_000, L=5)      # the reference's L > 1 kernel: recorded, not asserted invariant
    assert out["var"] > 0.25


def test_edit_request_weights_against_scipy(hostsim):
    from tests import parity
    parity.check_edit_weights_against_scipy()


def test_csmc_weights_against_scipy(hostsim):
    from tests import parity
    parity.check_csmc_weights_against_scipy()


def test_more_closed_forms(hostsim):
    """Gibbs assignments, Mask weights, IndexRequest on a plate, ChangeTarget evidence: against closed forms / scipy"""
    from tests import parity
    parity.check_more_closed_forms(n=100_000)


def test_sweep_verdict(hostsim, monkeypatch):
    """include/genmi.h gmx_sweep_verdict on the CPU mirror; finish() raises when a status word is set"""
    from tests import parity
    monkeypatch.setenv("GENMI_COMM", "peer")
    assert parity.check_sweep_verdict() >= 1


@pytest.mark.parametrize("J", [17, 40, 200])
def test_latent_vector_feeding_the_next_vector_site(hostsim, J):
    """8-schools at J schools: the model computes with the values of a long vector site (unrolled again; a chain of
    launches past one launch's slots) — every GFI method against the oracle, bit for bit"""
    from tests import parity
    parity.check_hierarchical_vector_latent(J=J)


@pytest.mark.parametrize("J", [12, 64])
def test_rows_of_logits_at_one_categorical_site(hostsim, J):
    """`categorical(logits [J, 3])` per particle: J draws at ONE site (softmax regression without a plate) — simulate,
    importance and update equal the oracle; more rows than are unrolled say so"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from tests import parity
    parity.check_rows_of_logits_at_one_site(B=33, J=J)
    if J == 64:
        @G.gen
        def wide():
            w = G.normal(0.0, 1.0) @ "w"
            return G.categorical(logits=jnp.stack([w * jnp.ones(65), jnp.zeros(65)], axis=-1)) @ "z"
        with pytest.raises(NotImplementedError, match="rows of logits"):
            G.vmap(lambda k: wide.simulate(k, ()))(G.split(G.key(0), 3))


@pytest.mark.parametrize("n_comp", [3, 20])
def test_mixture_with_latent_means(hostsim, n_comp):
    """`ys ~ normal(mus[zs], 1)` with `mus` a latent vector and `zs` categorical draws — values computed in the model read
    at traced indices (3 components: registers, a chain of selects; 20: a long vector site's stored values, a search
    loop); the means also as a plate's return values: equals the oracle"""
    from tests import parity
    parity.check_mixture_with_latent_means(B=33, n_comp=n_comp)


def test_slices_of_a_long_per_particle_vector(hostsim):
    """`ys[1:]`, `ys[10:40]`, `ys[1:] - ys[:-1]`, `ys[::-1]`, `ys[::2]` of a per-particle vector of 50 elements — and of a
    latent one — as a vector site's parameter: a sliced view used to read element t of the whole leaf in the site's
    loop, silently.  Equals the oracle"""
    from tests import parity
    assert parity.check_slices_of_a_long_per_particle_vector(B=33) == 7


def test_changed_per_particle_vector_argument(hostsim):
    """`update` under a changed [B, 30] argument that a large plate maps over, a long scan scans over and a vector site
    computes with: every element is re-scored (the loops' step reads were not seen as changed). Equals the oracle"""
    from tests import parity
    parity.check_changed_per_particle_vector_argument(B=33)


def test_gather_by_index_vector(hostsim):
    """`means[zs]` with `means` a table and `zs` a vector of indices given as an argument (a table, or one vector per
    particle; 8 / 30 elements; 3 / 20 components): importance and update under changed assignments equal the oracle"""
    from tests import parity
    assert parity.check_gather_by_index_vector() == 6


@pytest.mark.parametrize("T_", [5, 20])
def test_hmm_with_latent_transition_rows(hostsim, T_):
    """a discrete HMM whose transition rows are LATENT (`row.repeat(n=3)()`: a plate of Dirichlets) and ride in the scan's
    carry: `categorical(probs=trans[z])` selects a row of values held in registers at the traced state — unrolled scan
    (5 steps) and the loop form (20: the carry comes back from loop variables).  Weights and joint score against numpy"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp, ChoiceMap as C
    B = 9
    means = np.array([-2.0, 0.0, 2.0], np.float32)

    @G.gen
    def row():
        return G.dirichlet(jnp.ones(3)) @ "p"

    @G.gen
    def step(carry, y_):
        z, trans = carry
        zn = G.categorical(probs=trans[z]) @ "z"
        G.normal(jnp.array(means)[zn], 1.0) @ "y"
        return (zn, trans), zn

    @G.gen
    def hmm():
        trans = row.repeat(n=3)() @ "trans"
        z0 = G.categorical(logits=jnp.zeros(3)) @ "z0"
        (zT, _), _ = G.Scan(step, T_)((z0, trans), jnp.zeros(T_)) @ "chain"
        return zT
    ys = np.linspace(-2, 2, T_).astype(np.float32)
    tr, w = G.vmap(lambda k: hmm.importance(k, C["chain", :, "y"].set(jnp.array(ys)), ()))(G.split(G.key(1), B))
    ch = tr.get_choices()
    P = np.asarray(ch["trans", slice(None), "p"], np.float64)
    z0, zs = np.asarray(ch["z0"]), np.asarray(ch["chain", slice(None), "z"])
    lp = lambda y, m: -0.5 * (y - m) ** 2 - 0.5 * np.log(2 * np.pi)
    w_ref = lp(ys[None], means[zs].astype(np.float64)).sum(1)
    prev = np.concatenate([z0[:, None], zs[:, :-1]], 1)
    joint = 3 * np.log(2.0) - np.log(3.0) + np.log(P[np.arange(B)[:, None], prev, zs]).sum(1) + w_ref
    assert np.abs(np.asarray(w) - w_ref).max() < 1e-4
    assert np.abs(np.asarray(tr.get_score()) - joint).max() < 1e-4


@pytest.mark.parametrize("N", [8, 30])
def test_plate_over_a_per_particle_index_vector(hostsim, N):
    """a plate (unrolled / a loop) mapped over one INTEGER vector per particle whose elements index a table and a latent
    vector inside the element (`normal(means[z] + mus[z], 1)`), and `update` under a changed index vector: joint score
    and weights against numpy"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp, ChoiceMap as C, Diff
    B, K = 9, 3
    dev = G._lib.get().device
    rng = np.random.default_rng(0)
    lp = lambda y, m: -0.5 * (y - m) ** 2 - 0.5 * np.log(2 * np.pi)
    zs, zs2 = (rng.integers(0, K, size=(B, N)).astype(np.int32) for _ in range(2))
    means = np.linspace(-3, 3, K).astype(np.float32)

    @G.gen
    def elem(means, mus, z):
        return G.normal(means[z] + mus[z], 1.0) @ "v"

    @G.gen
    def model(means, zs):
        mus = G.normal(jnp.zeros(K), 1.0) @ "mus"
        elem.vmap(in_axes=(None, None, 0))(means, mus, zs) @ "p"
        return mus[0]
    tr = model.simulate(G.split(G.key(1), B), (jnp.array(means), torch.from_numpy(zs).to(dev)))
    ch = tr.get_choices()
    mus, v = np.asarray(ch["mus"], np.float64), np.asarray(ch["p", slice(None), "v"], np.float64)
    joint = lambda z: lp(mus, 0.0).sum(1) + lp(v, means[z].astype(np.float64) + np.take_along_axis(mus, z, 1)).sum(1)
    assert np.abs(np.asarray(tr.get_score()) - joint(zs)).max() < 1e-4
    _, w, _, _ = model.update(G.split(G.key(2), B), tr, C.empty(),
                              (Diff.no_change(jnp.array(means)), Diff(torch.from_numpy(zs2).to(dev), G.UnknownChange)))
    assert np.abs(np.asarray(w)).max() > 10.0
    assert np.abs(np.asarray(w) - (joint(zs2) - joint(zs))).max() < 1e-3


def test_traced_index_into_a_long_per_particle_vector(hostsim):
    """`xs[z]` with xs a per-particle vector of more than 16 elements (one [T, n] input slot, addressed by a loop's
    iteration number only) and z a traced index that is NOT a loop counter used to read element 0, silently: now a search
    loop at top level: assess and update of `y ~ normal(mus[z] + centres[z], 1)` equal the oracle"""
    import genjax_amd as G
    from genjax_amd import numpy as jnp, ChoiceMap as C
    from oracle import genjax_oracle as O
    n, B = 20, 9

    @G.gen
    def mix(centres):
        mus = G.normal(centres, 5.0) @ "mus"
        z = G.categorical(logits=jnp.zeros(n)) @ "z"
        G.normal(mus[z] + centres[z], 1.0) @ "y"
        return z

    @O.gen
    def omix(centres):
        mus = O.normal(centres, np.float32(5.0)) @ "mus"
        z = np.asarray(O.categorical(np.zeros(n, np.float32)) @ "z")
        O.normal((np.take_along_axis(mus, z[..., None], axis=-1)[..., 0] + centres[z]).astype(np.float32), np.float32(1.0)) @ "y"
        return z
    cen = np.linspace(-3, 3, n).astype(np.float32)
    keys, okeys = G.split(G.key(2), B), O.split(O.key(2), B)
    dev = G._lib.get().device
    tr, w = G.vmap(lambda k: mix.importance(k, C.kw(y=1.5), (cen,)))(keys)
    otr, ow = omix.importance(okeys, O.ChoiceMap.kw(y=np.full(B, 1.5, np.float32)), (cen,))
    assert np.array_equal(np.asarray(w), ow)
    sc, _ = mix.assess(tr.get_choices(), (cen,))
    assert np.array_equal(np.asarray(sc), otr.get_score())
    newz = (np.arange(B) % n).astype(np.int32)
    _, w3, _, _ = tr.update(G.key(5), C.kw(z=torch.from_numpy(newz).to(dev)))
    _, ow3, _ = omix.update(O.split(O.key(5), B), otr, O.C.d({"z": newz}), (cen,))
    assert np.array_equal(np.asarray(w3), ow3)
    newm = np.random.default_rng(0).normal(size=(B, n)).astype(np.float32)
    tr4, w4, _, _ = tr.update(G.key(6), C.kw(mus=torch.from_numpy(newm).to(dev)))
    otr4, ow4, _ = omix.update(O.split(O.key(6), B), otr, O.C.d({"mus": newm}), (cen,))
    assert np.array_equal(np.asarray(w4), ow4) and np.array_equal(np.asarray(tr4.get_score()), otr4.get_score())


@pytest.mark.parametrize("m", [4, 24])
def test_sweep_with_vector_observations(hostsim, m):
    """an HMM with m observations per step through BootstrapSweep: equals the oracle's sweep (m = 24: a counted loop in the
    step program)"""
    from tests import parity
    parity.check_sweep_with_vector_observations(m=m)


# ---------------------------------------------------------------------------
# the reference's cookbook as a parity corpus (tests/cookbook.py)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [10, 100, 1000])
def test_cookbook_speed_gains_sir(n):
    """3_speed_gains.ipynb c8 at the notebook's own `model_sizes` x N_sir = 100: vmap(model.importance), the categorical
    draw over the weights, the gather; and the slow route through default_proposal + two assess calls"""
    from tests import cookbook
    cookbook.check_speed_gains_sir(n=n, N=100)


@pytest.mark.parametrize("n,N", [(10, None), (17, None), (1000, None), (4096, None), (100_000, None), (10, 100), (100, 100), (1000, 100)])
def test_cookbook_speed_gains_mh_move(n, N):
    """3_speed_gains.ipynb c15: the MH move through `model.update` (fast) and through two `model.assess` (slow) — ONE trace
    (c17's sizes, 1e5 here on the CPU mirror; the GPU test goes to 1e6) and N chains at once"""
    from tests import cookbook
    cookbook.check_speed_gains_mh(n=n, N=N)


@pytest.mark.parametrize("N", [None, 50])
def test_cookbook_mcmc_notebook_proposal_takes_a_trace(N):
    """mcmc.ipynb c4-c16: `prop(tr, *_)` reads `tr.get_choices()["a"]` — a Trace as an argument of a @gen function
    (generative_function.py:72-230) — six MH moves with `lax.cond` over whole traces, every quantity bit-exact"""
    from tests import cookbook
    cookbook.check_mcmc_notebook(N=N)


def test_cookbook_importance_sampling_sir():
    from tests import cookbook
    cookbook.check_importance_sampling_sir()


@pytest.mark.parametrize("k,n", [(12, 40), (20, 100), (40, 500), (64, 1000)])
def test_cookbook_mixture_model_under_a_batch_of_keys(k, n):
    """7_application_dirichlet_mixture_model.ipynb c6 / c10 under 5 keys at once, up to BASELINE config 5's K = 64"""
    from tests import cookbook
    cookbook.check_mixture_notebook_under_a_batch(k=k, n=n)


@pytest.mark.parametrize("T_,N", [(8, None), (8, 6), (40, None), (40, 6), (300, 6)])
def test_cookbook_scan_outputs_and_array_carries(T_, N):
    """scan.py:200-294: an array as the initial carry; a long scan's stacked outputs computed with in the model"""
    from tests import cookbook
    cookbook.check_scan_outputs_and_array_carries(T_=T_, N=N)


def test_index_request_replacing_a_long_vector_site_inside_a_loop_is_refused_loudly():
    """`IndexRequest(idx, Update(C["y"].set(v)))` on a loop-form scan whose kernel holds a vector site of
    24 elements used to record the constraint at EVERY step (weights summed over all steps, silently).  The request now
    raises — naming the construct — for a launch-uniform [M] and for a per-particle [K, M] constraint alike; a scan the
    O(1) path takes (scan.py:325-416 on slices) still answers it bit-exact."""
    from genjax_amd import Diff, IndexRequest, Update
    T_, M, K = 20, 24, 5
    ones = np.ones(M, np.float32)

    def mk(o1):
        @genjax.gen
        def step(x, _):
            xn = genjax.normal(0.9 * x, 0.5) @ "x"
            genjax.normal(xn * jnp.array(ones), 1.0) @ "y"
            return (xn if o1 else xn + 0.1 * x), xn
        return step
    vec = np.linspace(-1, 1, M).astype(np.float32)
    per = torch.from_numpy(np.tile(vec, (K, 1)) + np.arange(K, dtype=np.float32)[:, None])
    a = (torch.zeros(K), None)
    sc = mk(False).scan(n=T_)
    tr = sc.simulate(genjax.split(genjax.key(1), K), a)
    for con in (jnp.array(vec), per):
        with pytest.raises(NotImplementedError, match="vector-valued site of more than 16 elements"):
            IndexRequest(7, Update(C["y"].set(con))).edit(genjax.split(genjax.key(2), K), tr, Diff.no_change(a))
    # the O(1) form (the kernel's return value depends on the carry through its choices alone)
    sc1 = mk(True).scan(n=T_)
    tr1 = sc1.simulate(genjax.split(genjax.key(1), K), a)

    @O.gen
    def ostep(x, _):
        xn = O.normal((np.float32(0.9) * np.asarray(x, np.float32)).astype(np.float32), np.float32(0.5)) @ "x"
        O.normal((np.asarray(xn, np.float32)[..., None] * ones).astype(np.float32), np.float32(1.0)) @ "y"
        return xn, xn
    osc = O.Scan(ostep, T_)
    otr = osc.simulate(O.split(O.key(1), K), (np.zeros(K, np.float32), None))
    new, w, _, _ = IndexRequest(7, Update(C["y"].set(jnp.array(vec)))).edit(genjax.split(genjax.key(2), K), tr1, Diff.no_change(a))
    onew, ow = O.scan_edit_index(osc, O.split(O.key(2), K), otr, (np.zeros(K, np.float32), None), 7,
                                 lambda k, sl, ar: ostep.update(k, sl, O.C.d({"y": vec}), ar)[:2])
    assert np.array_equal(w.numpy(), ow) and np.array_equal(new.get_choices()["y"].numpy(), onew.get_choices()["y"])


def test_traced_negative_index_wraps_like_jax_and_static_false_mask_gives_way():
    """`xs[z]` with z = -2 reads row n - 2 on values in registers (tracer.sym_take) as it does on a leaf in
    memory; `Choice(Mask(v, False)) | Choice(b)` is b (functional_types.py:312-316)"""
    from genjax_amd.core.choice_map import _or_values
    from genjax_amd.core.mask import Mask

    @genjax.gen
    def m(z):
        xs = jnp.stack([genjax.normal(float(j), 0.01) @ f"x{j}" for j in range(4)])
        return xs[z]
    tr = m.simulate(genjax.split(genjax.key(0), 6), (torch.tensor([-2, -1, 0, 1, 2, 3], dtype=torch.int32),))
    r = tr.get_retval().numpy()
    assert np.allclose(r, [2, 3, 0, 1, 2, 3], atol=0.1)
    assert _or_values(Mask(1.0, False), 2.0) == 2.0


@pytest.mark.parametrize("npts,J", [(100, 40), (500, 200), (5000, 1000)])
def test_hmc_and_regenerate_through_long_vector_sites(npts, J):
    """linear regression with 100 / 500 / 5 000 points `HMC(S["a"] | S["b"])`, 8-schools at J = 40 / 200 /
    1 000 `HMC(mu)`, `HMC(mu | log_tau)`, `Regenerate(theta)`, `HMC(theta)`, `HMC(mu | log_tau | theta)`, a latent vector
    read elementwise by two later sites, ONE trace — one launch each, bit-exact against the oracle; `jnp.sum(theta)` refused"""
    from tests import cookbook
    cookbook.check_hmc_through_long_vector_sites(npts=npts, J=J)


def test_support_matrix_is_current():
    """SUPPORT.md is GENERATED (tools/support_matrix.py runs every form x method x size cell on this CPU mirror): a sample of
    freshly run rows must be in the committed file word for word — the table cannot drift from the code"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("support_matrix", os.path.join(root, "tools", "support_matrix.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    committed = open(os.path.join(root, "SUPPORT.md")).read()
    rows = sm.generate(sample=7)
    assert len(rows) >= 6
    for name, size, batch, cells in rows:
        line = f"| {name} | {size} | {batch} | " + " | ".join(cells) + " |"
        assert line in committed, line
