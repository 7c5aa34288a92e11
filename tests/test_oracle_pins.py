"""Pins for the CPU oracle (no GPU): known-answer vectors, the one numeric
literal in the reference's tests, closed forms, independent f64 libraries, and
the committed golden fixtures (tests/golden/oracle_vectors.json)."""
import ctypes
import hashlib
import json
import math
import os

import numpy as np
import pytest
from scipy import stats
from scipy.special import erfinv, gammaln

from oracle import genjax_oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


def _tf(k0, k1, c0, c1):
    a = lambda v: np.array([v], dtype=np.uint32)
    o0, o1 = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    O.lib().orc_threefry2x32(ctypes.c_int64(1), O._p(a(k0)), O._p(a(k1)), O._p(a(c0)), O._p(a(c1)), O._p(o0), O._p(o1))
    return int(o0[0]), int(o1[0])


def test_threefry_random123_kat():
    """Random123 Threefry-2x32-20 known-answer vectors."""
    for kat in GOLD["threefry_kat"]:
        assert list(_tf(*kat["key"], *kat["ctr"])) == kat["out"]


def test_host_threefry_matches_kat():
    from genjax_amd.random import threefry2x32
    for kat in GOLD["threefry_kat"]:
        o0, o1 = threefry2x32(kat["key"][0], kat["key"][1], kat["ctr"][0], kat["ctr"][1])
        assert [int(o0), int(o1)] == kat["out"]


def test_key_algebra_golden():
    k = O.key(314159)
    assert [int(v) for v in k] == GOLD["key"] == [0, 314159]
    assert O.split(k, 8).tolist() == GOLD["split8"]
    assert [O.fold_in(k, d).tolist() for d in (1, 2, 3, 4)] == GOLD["fold_in_1_to_4"]
    # fold_in(key, d) is the same Threefry block as split child d (App. A.2)
    assert O.fold_in(k, 3).tolist() == O.split(k, 8)[3].tolist()
    assert O.bits32(k[None, :], np.arange(16, dtype=np.uint64)).tolist() == GOLD["bits32_first16"]


def _hexf(a):
    return [float(np.float32(v)).hex() for v in np.asarray(a).reshape(-1)]


def test_sample_streams_golden():
    k = O.key(314159)
    site = O.fold_in(k, 1)
    assert _hexf(O.random_uniform(site, (64,))) == GOLD["uniform64"]
    assert _hexf(O.random_normal(site, (64,))) == GOLD["normal64"]
    assert _hexf(O.random_gumbel(site, (64,))) == GOLD["gumbel64"]
    keys = O.split(k, 64)
    assert _hexf(O.normal.sample(O.fold_in(keys, 1), np.float32(1.5), np.float32(2.0))) == GOLD["normal_site_per_particle64"]
    assert [int(v) for v in O.flip.sample(O.fold_in(keys, 2), np.float32(0.3))] == GOLD["flip_per_particle64"]
    logits = np.log(np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float32))
    assert [int(v) for v in O.categorical.sample(O.fold_in(keys, 3), logits)] == GOLD["categorical_per_particle64"]
    assert _hexf(O.beta.sample(O.fold_in(keys, 4), np.float32(2.0), np.float32(2.0))) == GOLD["beta22_per_particle64"]


def test_uniform_is_mantissa_trick():
    """jax.random.uniform: bits >> 9 | 0x3F800000, minus 1 (App. A.2)."""
    bits = np.array([0, 1 << 9, 0xFFFFFFFF, 0x80000000], dtype=np.uint32)
    u = O.unit_from_bits(bits)
    assert u.tolist() == [0.0, 2.0 ** -23, 1.0 - 2.0 ** -23, 0.5]


def test_reference_assess_literal():
    """tests/generative_functions/test_static_gen_fn.py:317-318:
    assess({y1: 1.0, y2: -1.0}) of two normal(0,1) sites == -2.837877."""
    @O.gen
    def model():
        y1 = O.normal(0.0, 1.0) @ "y1"
        y2 = O.normal(0.0, 1.0) @ "y2"
        return y1 + y2
    score, _ = model.assess(O.C.kw(y1=np.float32(1.0), y2=np.float32(-1.0)), ())
    assert float(score) == pytest.approx(-2.837877, abs=5e-7)
    assert float(np.float32(score)) == GOLD["assess_literal"]


def test_logpdfs_against_scipy():
    xs = np.linspace(-4, 6, 101).astype(np.float32)
    np.testing.assert_allclose(O.normal.logpdf(xs, np.float32(0.5), np.float32(1.5)),
                               stats.norm.logpdf(xs.astype(np.float64), 0.5, 1.5), rtol=2e-6, atol=2e-6)
    ps = np.linspace(0.01, 0.99, 99).astype(np.float32)
    np.testing.assert_allclose(O.beta.logpdf(ps, np.float32(2.0), np.float32(3.0)),
                               stats.beta.logpdf(ps.astype(np.float64), 2.0, 3.0), rtol=1e-5, atol=1e-5)
    assert float(O.beta.logpdf(np.float32(0.3), np.float32(2.0), np.float32(2.0))) == pytest.approx(0.2311117, abs=2e-6)
    np.testing.assert_allclose(O.flip.logpdf(np.array([1, 0], np.int32), np.float32(0.7)),
                               [math.log(0.7), math.log(0.3)], rtol=1e-6)
    np.testing.assert_allclose(O.uniform.logpdf(np.array([0.5, 3.0], np.float32), np.float32(0.0), np.float32(2.0)),
                               [-math.log(2.0), -np.inf])
    np.testing.assert_allclose(O.bernoulli.logpdf(np.array([1, 0], np.int32), np.float32(0.4)),
                               [-math.log1p(math.exp(-0.4)), -math.log1p(math.exp(0.4))], rtol=1e-6)
    assert _hexf(O.normal.logpdf(np.array([-2.0, -0.5, 0.0, 0.3, 1.0, 4.0], np.float32), np.float32(0.5), np.float32(1.5))) == GOLD["normal_logpdf"]


def test_elementary_function_accuracy():
    x = np.linspace(-87, 88, 200_001).astype(np.float32)
    r = np.exp(x.astype(np.float64))
    assert np.max(np.abs(O.exp(x) - r) / r) < 1.5e-7
    x = np.exp(np.linspace(-80, 80, 200_001)).astype(np.float32)
    r = np.log(x.astype(np.float64))
    assert np.max(np.abs(O.log(x) - r) / np.maximum(np.abs(r), 1e-30)) < 1.5e-7
    x = np.linspace(-0.999, 50, 100_001).astype(np.float32)
    r = np.log1p(x.astype(np.float64))
    assert np.max(np.abs(O.log1p(x) - r) / np.maximum(np.abs(r), 1e-30)) < 3e-7
    x = np.exp(np.linspace(-6, 8, 100_001)).astype(np.float32)
    r = gammaln(x.astype(np.float64))
    assert np.max(np.abs(O.lgamma(x) - r) / np.maximum(1.0, np.abs(r))) < 4e-6
    x = np.linspace(-0.99999, 0.99999, 100_001).astype(np.float32)
    r = erfinv(x.astype(np.float64))
    assert np.max(np.abs(O.erfinv(x) - r) / np.maximum(np.abs(r), 1e-6)) < 1e-5
    x = np.linspace(-100, 100, 100_001).astype(np.float32)
    assert np.max(np.abs(O.cos(x) - np.cos(x.astype(np.float64)))) < 2e-7
    assert np.max(np.abs(O.sin(x) - np.sin(x.astype(np.float64)))) < 2e-7
    assert O.exp(np.float32(-200.0)) == 0.0 and np.isinf(O.exp(np.float32(100.0)))
    assert np.isneginf(O.log(np.float32(0.0))) and np.isnan(O.log(np.float32(-1.0)))


def test_sampler_distributions():
    """Distributional parity (the only kind available for Beta: SURVEY.md §7)."""
    n = 200_000
    keys = O.split(O.key(1), n)
    z = O.normal.sample(keys, np.float32(1.0), np.float32(2.0))
    assert stats.kstest(z[:20000], "norm", args=(1.0, 2.0)).pvalue > 1e-3
    assert abs(z.mean() - 1.0) < 0.02 and abs(z.std() - 2.0) < 0.02
    b = O.beta.sample(keys, np.float32(2.0), np.float32(5.0))
    assert stats.kstest(b[:20000], "beta", args=(2.0, 5.0)).pvalue > 1e-3
    b = O.beta.sample(keys, np.float32(0.5), np.float32(0.7))       # boosted branch (alpha < 1)
    assert stats.kstest(b[:20000], "beta", args=(0.5, 0.7)).pvalue > 1e-3
    f = O.flip.sample(keys, np.float32(0.3))
    assert abs(f.mean() - 0.3) < 5e-3
    c = O.categorical.sample(keys, np.log(np.array([0.5, 0.25, 0.125, 0.125], np.float32)))
    np.testing.assert_allclose(np.bincount(c, minlength=4) / n, [0.5, 0.25, 0.125, 0.125], atol=5e-3)


def test_smc_closed_forms():
    """tests/inference/test_smc.py:32-87: flip-flip exact log-marginals."""
    @O.gen
    def flip_flip_trivial():
        _ = O.flip(0.5) @ "x"
        _ = O.flip(0.7) @ "y"
    tgt = O.Target(flip_flip_trivial, (), O.C.kw(y=True))
    k = O.key(314159)
    z1 = O.log_marginal_likelihood_estimate(O.Importance(tgt), k)
    assert float(z1) == pytest.approx(math.log(0.7), rel=1e-1)
    zk = O.log_marginal_likelihood_estimate(O.ImportanceK(tgt, 1000), k)
    assert float(zk) == pytest.approx(math.log(0.7), rel=1e-3)

    @O.gen
    def flip_flip():
        v1 = O.flip(0.5) @ "x"
        p = np.where(v1, np.float32(0.9), np.float32(0.3))
        _ = O.flip(p) @ "y"
    tgt = O.Target(flip_flip, (), O.C.kw(y=True))
    zk = O.log_marginal_likelihood_estimate(O.ImportanceK(tgt, 2000), k)
    assert float(zk) == pytest.approx(math.log(0.5 * 0.9 + 0.5 * 0.3), rel=1e-1)


def test_resampling_golden_and_properties():
    rng = np.random.default_rng(0)
    for n_ in (8, 1024, 1_000_000):
        lw = rng.normal(0, 2, n_).astype(np.float32)
        g = GOLD["resample"][str(n_)]
        assert hashlib.sha256(lw.tobytes()).hexdigest() == g["lw_sha256"]
        cdf, total, M, shift = O.weight_cdf(lw)
        assert str(total) == g["total"] and shift == g["shift"] and float(M).hex() == g["max"]
        assert hashlib.sha256(cdf.tobytes()).hexdigest() == g["cdf_sha256"]
        for kind, name in ((O.SYSTEMATIC, "systematic"), (O.STRATIFIED, "stratified"), (O.MULTINOMIAL, "multinomial")):
            anc = O.ancestors(kind, O.key(99), cdf)
            assert hashlib.sha256(anc.tobytes()).hexdigest() == g[name + "_sha256"]
            if g[name] is not None:
                assert anc.tolist() == g[name]
            if kind != O.MULTINOMIAL:
                assert np.all(np.diff(anc) >= 0)
            # offspring counts track the weights
            if n_ == 1_000_000:
                w = np.exp(lw.astype(np.float64) - lw.max())
                w /= w.sum()
                cnt = np.bincount(anc, minlength=n_)
                top = np.argsort(w)[-100:]
                bound = {O.SYSTEMATIC: 1.0, O.STRATIFIED: 2.0}.get(kind, 8 * np.sqrt(n_ * w[top]) + 1)
                assert np.all(np.abs(cnt[top] - n_ * w[top]) <= bound)
        anc_t = O.ancestors_multinomial_tiled(O.key(99), cdf)
        assert hashlib.sha256(anc_t.tobytes()).hexdigest() == g["multinomial_tiled_sha256"]
        if g["multinomial_tiled"] is not None:
            assert anc_t.tolist() == g["multinomial_tiled"]
        # multinomial with sorted uniforms: the Python-integer statement, the C merge, the committed vectors
        anc_s = O.ancestors_multinomial_sorted(O.key(99), cdf) if cdf.size <= 4096 else O.ancestors_multinomial_sorted_c(O.key(99), cdf)
        assert np.array_equal(anc_s, O.ancestors_multinomial_sorted_c(O.key(99), cdf)) and np.all(np.diff(anc_s) >= 0)
        assert hashlib.sha256(anc_s.tobytes()).hexdigest() == g["multinomial_sorted_sha256"]
        if g["multinomial_sorted"] is not None:
            assert anc_s.tolist() == g["multinomial_sorted"]
        assert O.sorted_exponentials(O.key(99), 8).tolist() == g["sorted_exponentials_head"]
        assert np.all(np.diff(anc_t // 1024) >= 0)           # ordered by the ancestor's tile


def test_systematic_matches_float_definition_small():
    """Integer rule == the textbook definition: first i with CDF_i > (j + u0)/n * total."""
    from fractions import Fraction
    rng = np.random.default_rng(4)
    lw = rng.normal(0, 1, 50).astype(np.float32)
    cdf, total, _, _ = O.weight_cdf(lw)
    k = O.key(5)
    u0 = Fraction(int(O.bits32(k, 0)) >> 9, 1 << 23)
    anc = O.ancestors(O.SYSTEMATIC, k, cdf)
    for j in range(50):
        pos = (j + u0) / 50 * total
        i = next(i for i in range(50) if int(cdf[i]) > pos)
        assert anc[j] == i


def test_kalman_vs_oracle_particle_filter():
    from genjax_amd import workloads
    from tests import parity
    T, n = 20, 20_000
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(O)
    res = parity.oracle_bootstrap_sweep(init, step, n, T, ys, O.key(314159))
    assert res["log_ml"] == pytest.approx(workloads.kalman_log_ml(ys), abs=0.15)


def test_c_ancestors_equal_python_integer_ancestors():
    """orc_ancestors (unsigned __int128) == ancestors() (Python integers) for every resampling kind."""
    rng = np.random.default_rng(0)
    for n in (1, 7, 1000, 20000):
        lw = rng.normal(0, 3, n).astype(np.float32)
        cdf, total, M, shift = O.weight_cdf(lw)
        for kind in (O.SYSTEMATIC, O.STRATIFIED, O.MULTINOMIAL):
            assert np.array_equal(O.ancestors(kind, O.key(n + kind), cdf), O.ancestors_c(kind, O.key(n + kind), cdf))
    dead = np.zeros(5, np.uint64)
    assert np.array_equal(O.ancestors_c(O.SYSTEMATIC, O.key(1), dead), np.full(5, 4, np.int32))


# ---------------------------------------------------------------------------
# Outputs of the real jax.random, as printed in JAX's own documentation (the "Pseudorandom numbers"
# tutorial, both its classic-PRNG edition and its jax >= 0.5 / threefry_partitionable edition).
# There is no network and no jax here, so these constants are the builder's recollection of those
# pages — but an accidental match of every printed digit is not credible, so they pin:
#   * the Threefry-2x32 block exactly as jax keys it (rotations, key schedule, word order);
#   * bits -> uniform (mantissa trick) and bits -> normal (sqrt2 * erf_inv) float pipelines;
#   * the PARTITIONABLE layout jax 0.5.2 defaults to: split child i = threefry(key, (0, i)) (both
#     words), scalar draw bits = hi ^ lo of threefry(key, (0, 0)).
# ---------------------------------------------------------------------------
def _tf(k0, k1, c0, c1):
    import ctypes
    a = [np.array([v], np.uint32) for v in (k0, k1, c0, c1)]
    o0, o1 = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    O.lib().orc_threefry2x32(ctypes.c_int64(1), *[O._p(x) for x in a], O._p(o0), O._p(o1))
    return int(o0[0]), int(o1[0])


def _from_bits(fn, bits):
    import ctypes
    out = np.zeros(1, np.float32)
    getattr(O.lib(), fn)(ctypes.c_int64(1), O._p(np.array([bits], np.uint32)), O._p(out))
    return out[0]


def test_jax_docs_classic_prng_values():
    # jax.random.split(PRNGKey(0)) -> [[4146024105, 967050713], [2718843009, 1272950319]]
    a, b = _tf(0, 0, 0, 2), _tf(0, 0, 1, 3)          # counts [0,1,2,3] -> x0 = [0,1], x1 = [2,3]
    assert [a[0], b[0], a[1], b[1]] == [4146024105, 967050713, 2718843009, 1272950319]
    # jax.random.normal(PRNGKey(0)) -> -0.20584226 ; jax.random.uniform(PRNGKey(0)) -> 0.41845703
    bits0 = _tf(0, 0, 0, 0)[0]
    assert _from_bits("orc_std_normal_from_bits", bits0) == np.float32(-0.20584226)
    assert _from_bits("orc_unit_from_bits", bits0) == np.float32(0.41845703)
    # jax.random.normal(PRNGKey(42)) -> -0.18471177
    assert _from_bits("orc_std_normal_from_bits", _tf(0, 42, 0, 0)[0]) == np.float32(-0.18471177)
    # "JAX - The Sharp Bits", section on random numbers (classic PRNG): key = PRNGKey(0);
    #   key, subkey = split(key)  -> new key [4146024105 967050713], new subkey [2718843009 1272950319] --> normal [-1.2515389]
    #   key, subkey = split(key)  -> new key [2384771982 3928867769], new subkey [1278412471 2182328957] --> normal [-0.58665055]
    # i.e. a CHAIN of two splits and a draw from each child: pins the key word order through two levels
    assert _from_bits("orc_std_normal_from_bits", _tf(2718843009, 1272950319, 0, 0)[0]) == np.float32(-1.2515389)
    k = (4146024105, 967050713)
    a, b = _tf(*k, 0, 2), _tf(*k, 1, 3)
    assert [a[0], b[0], a[1], b[1]] == [2384771982, 3928867769, 1278412471, 2182328957]
    assert _from_bits("orc_std_normal_from_bits", _tf(a[1], b[1], 0, 0)[0]) == np.float32(-0.58665055)
    # JAX quickstart (classic PRNG): random.normal(PRNGKey(0), (10,)) ->
    #   [-0.3721109 0.26423115 -0.18252768 -0.7368197 -0.44030377 -0.1521442 -0.67135346 -0.5908641 0.73168886 0.5673026]
    # classic layout of a 10-element draw: counters 0..9 split in halves, x0 = [0..4], x1 = [5..9]; outputs
    # concatenated [y0..., y1...] -> element j < 5 is the FIRST word of threefry(key, (j, j + 5)), element 5 + j the second.
    # Ten more arguments of sqrt(2) erf_inv, every printed digit reproduced.
    blocks = [_tf(0, 0, j, j + 5) for j in range(5)]
    draw = [_from_bits("orc_std_normal_from_bits", blk[w]) for w in (0, 1) for blk in blocks]
    printed = [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641,
               0.73168886, 0.5673026]
    assert [float(v) for v in draw] == [float(np.float32(p)) for p in printed]
    # "Pseudorandom numbers" tutorial, classic edition: key = PRNGKey(42); three rounds of
    #   new_key, subkey = split(key); val = normal(subkey); print(f"draw {i}: {val}"); key = new_key
    # prints  draw 0: 1.369469404220581 / draw 1: -0.19947023689746857 / draw 2: -2.298278331756592
    # (and, after the first split, new_key [2465931498 3679230171], subkey [255383827 267815257])
    k, draws = (0, 42), []
    for i in range(3):
        a, b = _tf(*k, 0, 2), _tf(*k, 1, 3)
        k, sub = (a[0], b[0]), (a[1], b[1])
        if i == 0:
            assert k == (2465931498, 3679230171) and sub == (255383827, 267815257)
        draws.append(repr(float(_from_bits("orc_std_normal_from_bits", _tf(*sub, 0, 0)[0]))))
    assert draws == ["1.369469404220581", "-0.19947023689746857", "-2.298278331756592"]
    # Sharp Bits again, continuing from key [2384771982 3928867769]:  key, *subkeys = split(key, 4);
    # normal(subkey, (1,)) for the three subkeys prints [-0.37533438] [0.98645043] [0.14553197]
    # (classic split(key, 4): counters 0..7 in halves, outputs concatenated and reshaped (4, 2))
    k = (2384771982, 3928867769)
    blk = [_tf(*k, j, j + 4) for j in range(4)]
    flat = [x[0] for x in blk] + [x[1] for x in blk]
    subs = [(flat[2 * i], flat[2 * i + 1]) for i in range(1, 4)]
    got = [_from_bits("orc_std_normal_from_bits", _tf(*s, 0, 0)[0]) for s in subs]
    assert got == [np.float32(-0.37533438), np.float32(0.98645043), np.float32(0.14553197)]


def test_jax_docs_partitionable_prng_values():
    """jax >= 0.5 (the reference pins jax 0.5.2, poetry.lock:1627): key = jax.random.key(42)."""
    k = O.key(42)
    assert O.normal.sample(k, np.float32(0), np.float32(1)) == np.float32(-0.028304616)     # random.normal(key)
    new_key, subkey = O.split(k)
    assert new_key.tolist() == [1832780943, 270669613] and subkey.tolist() == [64467757, 2916123636]
    assert O.normal.sample(subkey, np.float32(0), np.float32(1)) == np.float32(0.60576403)  # random.normal(subkey)
    # "individually": [random.normal(k) for k in random.split(key, 3)] -> [0.07592554 0.60576403 0.4323065]
    ks = O.split(k, 3)
    ind = np.array([O.normal.sample(ks[i], np.float32(0), np.float32(1)) for i in range(3)], np.float32)
    assert np.all(np.abs(ind.astype(np.float64) - [0.07592554, 0.60576403, 0.4323065]) < 5.1e-9)    # digits as printed
    # "all at once": random.normal(key, shape=(3,)) -> [-0.02830462 0.46713185 0.29570296]
    # (element j of a vector draw takes counter j under the SAME key)
    allatonce = O.normal.sample(k, np.zeros(3, np.float32), np.float32(1))
    assert np.all(np.abs(allatonce.astype(np.float64) - [-0.02830462, 0.46713185, 0.29570296]) < 5.1e-9)
    # the same tutorial's loop  `new_key, subkey = split(key); val = normal(subkey); key = new_key`  prints
    #   draw 0: 0.6057640314102173 / draw 1: -0.21089035272598267 / draw 2: -0.3948981463909149
    # (a CHAIN of splits in the partitionable layout: child 0 carries on, child 1 draws)
    draws, kk = [], k
    for _ in range(3):
        kk, sub = O.split(kk)
        draws.append(repr(float(O.normal.sample(sub, np.float32(0), np.float32(1)))))
    assert draws == ["0.6057640314102173", "-0.21089035272598267", "-0.3948981463909149"]
    # "The Sharp Bits", jax >= 0.5 edition: key = random.key(0); random.normal(key, shape=(1,)) -> [1.6226422];
    # key, subkey = split(key): new key [1797259609 2579123966], new subkey [928981903 3453687069] --> normal [-2.4424558]
    k0 = O.key(0)
    assert O.normal.sample(k0, np.zeros(1, np.float32), np.float32(1)).tolist() == [float(np.float32(1.6226422))]
    nk, sk = O.split(k0)
    assert nk.tolist() == [1797259609, 2579123966] and sk.tolist() == [928981903, 3453687069]
    assert O.normal.sample(sk, np.zeros(1, np.float32), np.float32(1)).tolist() == [float(np.float32(-2.4424558))]


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 5000, 100_003])
def test_integer_cdf_numpy_and_c_statements_agree(n):
    """genjax_oracle.weight_cdf (numpy) and orc_core.c::orc_weight_cdf_tiled state the block-floating-point
    CDF independently; they must produce the same integers, the CDF must be non-decreasing, and
    ref + log(total * 2^-shift) must be the log-sum-exp of the weights."""
    rng = np.random.default_rng(n)
    lw = (rng.normal(0, 3, n) - 5).astype(np.float32)
    if n > 10:
        lw[3] = -np.inf
        lw[n // 2] = np.nan
    a, b = O.weight_cdf(lw), O.weight_cdf_c(lw)
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    assert np.all(a[0][1:] >= a[0][:-1])
    fin = lw[np.isfinite(lw)].astype(np.float64)
    lse = fin.max() + math.log(np.sum(np.exp(fin - fin.max())))
    est = O.cdf_reference(a[2]) + math.log(a[1]) - a[3] * math.log(2.0)
    assert abs(est - lse) < 1e-6
    # one shard of a larger population: the caller's global max instead of the local one
    M = float(np.float32(a[2] + 7.5))
    c, d = O.weight_cdf(lw, n_total=4 * n, M=M), O.weight_cdf_c(lw, n_total=4 * n, M=M)
    assert np.array_equal(c[0], d[0]) and c[1] == d[1]


def test_c_sweep_equals_numpy_sweep():
    """oracle/orc_sweep.c (bench.py's cpu_baseline) == the numpy oracle sweep: particles, ancestors, totals"""
    import ctypes
    import os
    from genjax_amd import workloads
    from tests import parity
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "liborc_sweep.so")
    lib = ctypes.CDLL(so)
    n, T, seed = 3000, 5, 314159
    ys = workloads.lgssm_data(T)
    shift = O.cdf_shift(n)
    f32, u64, i32 = np.float32, np.uint64, np.int32
    x, x2, lw = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, f32)
    cdf, anc = np.zeros(n, u64), np.zeros(n, i32)
    maxs, totals = np.zeros(T, f32), np.zeros(T, u64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.orc_lgssm_sweep(ctypes.c_int64(n), ctypes.c_int64(T), P(ys), ctypes.c_uint32(0), ctypes.c_uint32(seed),
                             ctypes.c_float(0.9), ctypes.c_float(0.5), ctypes.c_float(1.0), ctypes.c_float(1.0),
                             ctypes.c_int(shift), P(x), P(x2), P(lw), P(cdf), P(anc), P(maxs), P(totals))
    assert rc == 0
    oi, ost = workloads.make_lgssm(O)
    ref = parity.oracle_bootstrap_sweep(oi, ost, n, T, ys, O.key(seed))
    assert [int(t) for t in totals] == [h["total"] for h in ref["hist"]]
    assert np.array_equal(x, ref["x"]) and np.array_equal(anc, ref["anc"]) and np.array_equal(lw, ref["lw"])
    assert np.array_equal(cdf, ref["hist"][-1]["cdf"])
