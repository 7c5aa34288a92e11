"""`-m gpu` parity tests: every check goes through libgenmi_hip.so (the C-ABI)
on an MI355X and is compared with the CPU oracle on the same seeded inputs.
Integer / index results must be bit-exact; float results are bit-exact by
construction for elementwise work (same IEEE op sequence) and within the
stated tolerance for tree-ordered reductions."""
import numpy as np
import pytest
import torch

from oracle import genjax_oracle as O
from tests import parity

pytestmark = pytest.mark.gpu


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_library_is_hip(gpu):
    assert gpu.device.type == "cuda"
    assert gpu.c.gmx_version() == 1


def test_key_kernels_bit_exact(gpu):
    import genjax_amd as G
    from genjax_amd.random import lazy_split
    k = G.key(314159)
    for n in (1, 63, 64, 1000, 100_003):
        dev = lazy_split(k, n).data().cpu().numpy().view(np.uint32)
        ref = O.split(O.key(314159), n)
        assert np.array_equal(dev, ref)
    rows = G.split(k, 7)
    big = G.random.Key(lazy=("rowsplit", rows, 5000)).data().cpu().numpy().view(np.uint32).reshape(7, 5000, 2)
    assert np.array_equal(big, O.split(O.split(O.key(314159), 7), 5000))


def _models(g):
    @g.gen
    def model(x_prev):
        x = g.normal(0.9 * x_prev, 0.5) @ "x"
        y = g.normal(x, 1.0) @ "y"
        return x
    return model


@pytest.mark.parametrize("n", [1, 255, 256, 257, 10_000])
def test_gfi_simulate_importance_assess(gpu, n):
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C
    m, mo = _models(G), _models(O)
    keys, keyso = G.split(G.key(7), n), O.split(O.key(7), n)
    xp = np.linspace(-3, 3, n).astype(np.float32)
    tr = m.simulate(keys, (_dev(xp),))
    tro = mo.simulate(keyso, (xp,))
    for a in ("x", "y"):
        assert np.array_equal(tr.get_choices()[a].cpu().numpy(), tro.get_choices()[a])
    assert np.array_equal(tr.get_score().cpu().numpy(), tro.get_score())
    tr, w = m.importance(keys, C["y"].set(0.7), (_dev(xp),))
    tro, wo = mo.importance(keyso, O.C.d({"y": np.float32(0.7)}), (xp,))
    assert np.array_equal(tr.get_choices()["x"].cpu().numpy(), tro.get_choices()["x"])
    assert np.array_equal(w.cpu().numpy(), wo)
    s, _ = m.assess(C.kw(x=_dev(xp), y=0.3), (_dev(xp),))
    so, _ = mo.assess(O.C.kw(x=xp, y=np.float32(0.3)), (xp,), batch_shape=(n,))
    assert np.array_equal(s.cpu().numpy(), so)


def test_specialized_equals_interpreter(gpu):
    """gmx_program_specialize: same bits as the interpreter on a program that
    exercises most op classes (Beta / categorical / trig included)."""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from genjax_amd import static

    @G.gen
    def m(a, b):
        p = G.beta(2.0, 3.0) @ "p"
        f = G.flip(p) @ "f"
        z = G.normal(jnp.where(f, a, b) + jnp.cos(a) * jnp.tanh(b), jnp.exp(0.1 * b)) @ "z"
        u = G.uniform(-1.0, 2.0) @ "u"
        k = G.categorical(logits=jnp.array([0.1, 0.2, 0.3]) * u) @ "k"
        w = G.bernoulli(logits=z) @ "w"
        return z * u + jnp.lgamma(jnp.abs(z) + 1.0)
    n = 20_000
    keys = G.split(G.key(3), n)
    a = _dev(np.linspace(-2, 2, n).astype(np.float32))
    b = _dev(np.linspace(1, -1, n).astype(np.float32))
    tr1 = m.simulate(keys, (a, b))
    comps = [c[0] for c in static._CACHE.values() if hasattr(c[0], "specialize")]
    assert comps and all(c.specialize() for c in comps), gpu.c.gmx_last_error()
    tr2 = m.simulate(keys, (a, b))
    for addr in ("p", "f", "z", "u", "k", "w"):
        v1, v2 = tr1.get_choices()[addr], tr2.get_choices()[addr]
        assert torch.equal(v1, v2), addr
    assert torch.equal(tr1.get_score(), tr2.get_score())
    assert torch.equal(tr1.get_retval(), tr2.get_retval())


def test_elementary_functions_bit_exact(gpu):
    """exp/log/... on the device == the oracle's C restatement, bit for bit."""
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 10, 50_000), rng.uniform(-88, 88, 50_000)]).astype(np.float32)
    pos = np.abs(x) + np.float32(1e-3)
    unit = rng.uniform(-0.999, 0.999, x.size).astype(np.float32)
    cases = [("exp", jnp.exp, O.exp, x), ("log", jnp.log, O.log, pos), ("log1p", jnp.log1p, O.log1p, pos - 1),
             ("sqrt", jnp.sqrt, O.sqrt, pos), ("sin", jnp.sin, O.sin, x), ("cos", jnp.cos, O.cos, x),
             ("tanh", jnp.tanh, O.tanh, x / 10), ("sigmoid", jnp.sigmoid, O.sigmoid, x),
             ("softplus", jnp.softplus, O.softplus, x), ("lgamma", jnp.lgamma, O.lgamma, pos)]
    for name, f, fo, arg in cases:
        @G.gen
        def m(v, f=f):
            G.normal(f(v), 1.0) @ "z"
            return f(v)
        _, r = m.assess(G.ChoiceMap.kw(z=0.0), (_dev(arg),))
        assert np.array_equal(r.cpu().numpy().view(np.uint32), fo(arg).view(np.uint32)), name


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 4096, 100_000, 1_000_000])
def test_weight_cdf_bit_exact(gpu, n):
    from genjax_amd.inference import smc
    rng = np.random.default_rng(n)
    lw = (rng.normal(0, 3, n) - 5).astype(np.float32)
    if n > 10:
        lw[3] = -np.inf
    cdf, total, mx, shift = smc.weight_cdf(_dev(lw))
    rc, rt, rm, rs = O.weight_cdf(lw)
    assert shift == rs
    assert float(mx.item()) == rm
    assert np.array_equal(cdf.cpu().numpy().view(np.uint64), rc)
    assert int(total.item()) == rt


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 7, 1000, 65_537])
def test_ancestors_bit_exact(gpu, kind, n):
    import genjax_amd as G
    from genjax_amd.inference import smc
    rng = np.random.default_rng(100 + n)
    lw = rng.normal(0, 2, n).astype(np.float32)
    cdf, total, _, _ = smc.weight_cdf(_dev(lw))
    anc = smc.ancestors_from_cdf(kind, G.key(99), cdf, total)
    rc, _, _, _ = O.weight_cdf(lw)
    ref = O.ancestors(kind, O.key(99), rc)
    assert np.array_equal(anc.cpu().numpy(), ref)
    if kind != 2:
        assert np.all(np.diff(anc.cpu().numpy()) >= 0)        # sortedness property


def test_ancestors_degenerate_weights(gpu):
    """all mass on one particle -> every ancestor is that particle"""
    import genjax_amd as G
    from genjax_amd.inference import smc
    n = 5000
    lw = np.full(n, -1e30, dtype=np.float32)
    lw[1234] = 0.0
    cdf, total, _, _ = smc.weight_cdf(_dev(lw))
    for kind in (0, 1, 2):
        anc = smc.ancestors_from_cdf(kind, G.key(1), cdf, total).cpu().numpy()
        assert np.all(anc == 1234)


@pytest.mark.parametrize("shape", [(1, 10), (1, 4097), (1, 1_000_000), (50, 50), (3, 70_000), (1000, 7)])
def test_logsumexp(gpu, shape):
    from genjax_amd import engine
    rng = np.random.default_rng(5)
    x = rng.normal(0, 5, shape).astype(np.float32)
    got = engine.logsumexp_rows(_dev(x)).cpu().numpy()
    ref = np.log(np.sum(np.exp(x.astype(np.float64) - x.max(-1, keepdims=True)), -1)) + x.max(-1)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)     # f32 tree sum vs f64


def test_gather_and_categorical(gpu):
    import genjax_amd as G
    from genjax_amd import engine
    rng = np.random.default_rng(3)
    n = 10_000
    a = rng.normal(size=n).astype(np.float32)
    b = rng.integers(0, 100, size=(n, 3)).astype(np.int32)
    c = rng.integers(0, 2, size=n).astype(bool)
    anc = rng.integers(0, n, size=7777).astype(np.int32)
    ga, gb, gc = engine.gather_leaves([_dev(a), _dev(b), _dev(c)], _dev(anc))
    assert np.array_equal(ga.cpu().numpy(), a[anc])
    assert np.array_equal(gb.cpu().numpy(), b[anc])
    assert np.array_equal(gc.cpu().numpy(), c[anc])
    logits = rng.normal(size=(64, 500)).astype(np.float32)
    keys = G.split(G.key(11), 64)
    idx = engine.categorical_rows(keys, _dev(logits)).cpu().numpy()
    ref = O.categorical.sample(O.split(O.key(11), 64), logits)
    assert np.array_equal(idx, ref)


@pytest.mark.parametrize("n,T,capture,specialize", [(4096, 5, False, False), (4096, 5, True, False),
                                                    (4096, 5, False, True), (100_000, 8, True, True),
                                                    (100_000, 8, True, False)])
def test_lgssm_sweep_matches_oracle(gpu, n, T, capture, specialize):
    """interpreter AND hiprtc-specialised kernels vs the oracle, bit for bit"""
    res = parity.check_lgssm_sweep(n=n, T=T, capture=capture, specialize=specialize)
    assert res["ancestors_equal"] and res["x_equal"] and res["totals_equal"], res
    assert res["lw_max_abs_diff"] == 0.0
    assert res["log_ml"] == res["log_ml_oracle"]


def test_full_size_sweep_properties(gpu):
    """BASELINE config 2 at full size (1e6 particles, T = 100): size-independent
    properties — sorted ancestors, determinism, and log-ML within Monte-Carlo
    error of the exact Kalman answer."""
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference.smc import BootstrapSweep
    n, T = 1_000_000, 100
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    sw = BootstrapSweep(init, step, n, T).prepare(G.key(314159), torch.from_numpy(ys)).capture()
    sw.launch()
    a = sw.log_ml()
    anc1 = sw.anc.clone()
    sw.launch()
    assert sw.log_ml() == a and torch.equal(anc1, sw.anc)          # idempotent / deterministic
    assert bool(torch.all(anc1[1:] >= anc1[:-1]))                  # systematic => sorted
    assert int(sw.ws.view(torch.int32)[1].item()) == 0             # scan never hit its spin bound
    kal = workloads.kalman_log_ml(ys)
    assert abs(a - kal) < 0.05, (a, kal)
