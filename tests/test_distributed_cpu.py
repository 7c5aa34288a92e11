"""N > 1 path on CPU: world_size-2 (and 4) gloo jobs of ShardedBootstrapSweep
must reproduce the single-process oracle sweep BIT FOR BIT (global-index keys,
integer CDF, exact slot bounds => results independent of the rank count)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import genjax_oracle as O
from tests import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, worker_args, attempts=2, extra_env=None):
    """torch.distributed.run with `world` gloo ranks of tests/dist_worker.py on 127.0.0.1.  The rendezvous port
    is picked free and then released, so another process can grab it in between: a failed launch is retried
    once on a new port."""
    r = None
    for _ in range(attempts):
        port = str(_free_port())
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1", **(extra_env or {}))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.join(ROOT, "tests", "dist_worker.py")] + [str(a) for a in worker_args]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode == 0:
            break
    return r


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,capacity,tiles,prog_stats,na,fused", [
    (2, 0, 1, "1", 0, 1), (4, 0, 1, "1", 0, 1), (2, 3, 1, "1", 0, 1), (2, 0, 0, "1", 0, 1),
    (2, 0, 1, "0", 0, 1), (2, 0, 1, "1", 0, 0), (2, 0, 1, "1", 1, 1), (4, 3, 1, "1", 1, 1),
    (8, 0, 1, "1", 0, 1), (8, 3, 1, "1", 1, 1)])        # EIGHT ranks (BASELINE's node): plain, and overflow re-run + noise ahead
def test_sharded_sweep_equals_single_process_oracle(tmp_path, world, capacity, tiles, prog_stats, na, fused):
    """capacity 0 = default (fast path, no overflow); capacity 3 forces the
    overflow flag and the full-capacity re-run.  tiles 1 = the two-collective step (all-gather of tile
    statistics + all-to-all), 0 = ShardedBootstrapSweep(cdf_form=True): max all-reduce + local CDF + totals all-gather +
    all-to-all.  prog_stats "0": the site program does not write the tile statistics itself, a gmx_tile_stats launch does.
    fused 0: gmx_shard_totals + gmx_shard_step_tiles instead of the one-launch gmx_shard_step_fused.
    na 1: NOISE AHEAD on the sharded sweep — the step's draws come from background programs keyed by the GLOBAL
    particle index (lazy_split offset; GMX_KEY_ROWSPLIT rows + index_offset), launched a group of steps ahead."""
    from genjax_amd import workloads
    n_total, T = (8192 if world == 8 else 4096), 6          # (1024 per rank: shards start on a CDF tile boundary)
    out = str(tmp_path / "shard")
    r = _launch(world, [out, str(n_total // world), str(T)] + ([str(capacity)] if capacity else []),
                extra_env={"GENMI_HOSTSIM_TILE_STATS": prog_stats, "GENMI_NOISE_GROUP": "3",
                           "GENMI_TEST_OPTS": json.dumps({"noise_ahead": na, "cdf_form": not tiles, "fused": fused})})
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    ys = workloads.lgssm_data(T)
    oi, ost = workloads.make_lgssm(O)
    ref = parity.oracle_bootstrap_sweep(oi, ost, n_total, T, ys, O.key(314159))
    assert [int(t) for t in meta["totals"]] == [h["total"] for h in ref["hist"]]
    assert meta["log_ml"] == ref["log_ml"]
    assert meta["reruns"] == (1 if capacity else 0), meta
    # the sharded sweep returns the RESAMPLED particles of the last step
    assert np.array_equal(x, ref["x"][ref["anc"]])


@pytest.mark.parametrize("world,capacity", [(2, 0), (4, 0), (4, 3)])
def test_sharded_sweep_with_the_sorted_multinomial_equals_single_process_oracle(tmp_path, world, capacity):
    """resample="multinomial_sorted" on the sharded router — every rank draws the same table of the
    N global slots (order statistics, integers) and routes against it (gmx_shard_step_sorted); equal to the
    single-process oracle sweep whatever the rank count, also through the capacity-overflow re-run."""
    from genjax_amd import workloads
    n_total, T = 4096, 6
    out = str(tmp_path / "shard")
    r = _launch(world, [out, str(n_total // world), str(T)] + ([str(capacity)] if capacity else []),
                extra_env={"GENMI_TEST_OPTS": json.dumps({"resample": "multinomial_sorted"})})
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    ys = workloads.lgssm_data(T)
    oi, ost = workloads.make_lgssm(O)
    ref = parity.oracle_bootstrap_sweep(oi, ost, n_total, T, ys, O.key(314159), kind=O.MULTINOMIAL_SORTED)
    assert [int(t) for t in meta["totals"]] == [h["total"] for h in ref["hist"]]
    assert meta["log_ml"] == ref["log_ml"]
    assert meta["reruns"] == (1 if capacity else 0), meta
    assert np.array_equal(x, ref["x"][ref["anc"]])


@pytest.mark.parametrize("comm,world,capacity,na,prog_stats", [
    ("p2p", 2, 0, 0, "1"), ("p2p", 4, 3, 0, "1"), ("p2p", 2, 0, 1, "1"),
    ("peer", 2, 0, 0, "1"), ("peer", 4, 3, 0, "1"), ("peer", 2, 0, 1, "1"), ("peer", 4, 0, 1, "0")])
def test_sharded_sweep_over_the_peer_mapped_communicator(tmp_path, comm, world, capacity, na, prog_stats):
    """GENMI_COMM=p2p (include/genmi.h "Peer-mapped exchange"; comm.P2PComm): every collective of the sharded step is
    ONE exchange over peer-mapped memory — put into the peers' buffers, a flag per peer, a wait on the own flags; the
    epoch lives with the flags.  GENMI_COMM=peer ("Fused peer exchange"; comm.PeerComm): NO collective launch — the
    site program's epilogue puts its tile statistics into the peers' landing tables (prog_stats "0": gmx_tile_stats +
    gmx_peer_put_stats do), gmx_shard_step_peer reads them as tagged granules, puts the offspring states and waits for
    the ones its own slots need.  The CPU mirror runs both protocols over POSIX shared memory, so the gloo ranks really
    write into each other's buffers: world 2 and 4 (the latter through the capacity-overflow re-run, which re-allocates
    the exchange buffers collectively) equal the single-process oracle bit for bit, also with noise ahead."""
    from genjax_amd import workloads
    n_total, T = 4096, 6
    out = str(tmp_path / "shard_p2p")
    r = _launch(world, [out, str(n_total // world), str(T)] + ([str(capacity)] if capacity else []),
                extra_env={"GENMI_COMM": comm, "GENMI_NOISE_GROUP": "3", "GENMI_HOSTSIM_TILE_STATS": prog_stats,
                           "GENMI_TEST_OPTS": json.dumps({"noise_ahead": na})})
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    assert meta["communicator"].startswith(comm)
    ys = workloads.lgssm_data(T)
    oi, ost = workloads.make_lgssm(O)
    ref = parity.oracle_bootstrap_sweep(oi, ost, n_total, T, ys, O.key(314159))
    assert [int(t) for t in meta["totals"]] == [h["total"] for h in ref["hist"]]
    assert meta["log_ml"] == ref["log_ml"] and meta["reruns"] == (1 if capacity else 0)
    assert np.array_equal(x, ref["x"][ref["anc"]])


def test_sharded_importancek_over_the_peer_mapped_communicator(tmp_path):
    """BASELINE config 4 sharded with GENMI_COMM=p2p: one all-gather (tile statistics) + ONE all-to-all of the packed
    10-latent trace, both as peer-mapped exchanges"""
    k_total, world = 4096, 2
    out = str(tmp_path / "schools_p2p")
    r = _launch(world, [out, str(k_total // world), "0", "0", "schools"], extra_env={"GENMI_COMM": "p2p"})
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out + ".npz")
    info = json.load(open(out + ".info.json"))
    assert info["collectives"]["all_gather"] == 1 and info["collectives"]["all_to_all"] == 1
    sig = np.array(parity.SCHOOL_SIGMA, np.float32)

    @O.gen
    def o_schools():
        mu = O.normal(0.0, 5.0) @ "mu"
        log_tau = O.normal(0.0, 1.0) @ "log_tau"
        theta = O.normal(mu[..., None] * np.ones(8, np.float32), O.exp(log_tau)[..., None] * np.ones(8, np.float32)) @ "theta"
        _ = O.normal(theta, sig) @ "y"
        return theta
    oc = O.ImportanceK(O.Target(o_schools, (), O.C.d({"y": parity.SCHOOL_Y})), k_total).run_smc(O.key(2))
    cdf, total, M, shift = O.weight_cdf(oc.get_log_weights())
    anc = O.ancestors(O.SYSTEMATIC, O.split(O.key(2))[0], cdf)
    assert np.array_equal(got["theta"], oc.get_particles().get_choices()["theta"][anc])


@pytest.mark.parametrize("world,capacity,comm", [(2, 0, None), (4, 7, None), (2, 0, "peer"), (4, 7, "peer"), (8, 0, None), (8, 7, "peer")])
def test_sharded_mh_sweep_equals_single_process_oracle(tmp_path, world, capacity, comm):
    """BASELINE config 3 sharded: nonlinear SSM + one MH move per step, two routed leaves (the particle and
    the state it was extended from).  Must equal the single-process oracle for any rank count; capacity 7
    forces the overflow re-run."""
    n_total, T = (8192 if world == 8 else 4096), 4          # 1024 per rank at world 4 / 8: shards start on a CDF tile boundary
    out = str(tmp_path / "shard_mh")
    r = _launch(world, [out, str(n_total // world), str(T), str(capacity), "mh"], extra_env={"GENMI_COMM": comm} if comm else None)
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    ref = parity.oracle_nlssm_mh_sweep(n_total, T, 7)
    assert np.array_equal(x, ref["resampled"])
    assert abs(meta["log_ml"] - sum(ref["terms"])) < 1e-9 * max(1.0, abs(sum(ref["terms"])))
    assert meta["reruns"] == (1 if capacity else 0), meta


@pytest.mark.parametrize("world,capacity,comm", [(2, 0, None), (2, 5, None), (2, 5, "peer")])
def test_sharded_vector_state_sweep_equals_single_process_oracle(tmp_path, world, capacity, comm):
    """a 2-vector state (constant-velocity tracker): every component is one routed leaf (same plan, one
    gmx_shard_step + one all-to-all each); equals the single-process oracle, also through the overflow re-run."""
    n_total, T = 2048, 5
    out = str(tmp_path / "shard_vec")
    r = _launch(world, [out, str(n_total // world), str(T), str(capacity), "vec"], extra_env={"GENMI_COMM": comm} if comm else None)
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    ref = parity.oracle_tracker_sweep(n_total, T, 5)
    assert x.shape == (n_total, 2)
    assert np.array_equal(x, ref["x"][ref["anc"]])
    assert meta["log_ml"] == ref["log_ml"]
    assert meta["reruns"] == (1 if capacity else 0), meta


@pytest.mark.parametrize("world,capacity,comm,D", [(2, 0, None, 2), (2, 6, None, 2), (2, 0, "peer", 2), (2, 6, "peer", 2),
                                                   (2, 0, "peer", 6), (2, 6, "peer", 6)])
def test_sharded_vector_state_mh_sweep_equals_single_process_oracle(tmp_path, world, capacity, comm, D):
    """a D-vector state AND one MH move per step: the particle and the state it was extended from travel as
    2 x D routed leaves (D = 6: twelve, past the eight the fused peer exchange was built for — GMX_PEER_MAX_LEAVES is
    32); equals the single-process oracle, also through the overflow re-run."""
    n_total, T = 2048, 4
    out = str(tmp_path / "shard_vecmh")
    r = _launch(world, [out, str(n_total // world), str(T), str(capacity), {2: "vecmh", 6: "vec6mh"}[D]],
                extra_env={"GENMI_COMM": comm} if comm else None)
    assert r.returncode == 0, r.stderr[-3000:]
    x = np.load(out + ".npy")
    meta = json.load(open(out + ".json"))
    oi, ost = parity.make_vec_mh(O, lambda *v: np.stack(v, axis=-1), np.ones(D, np.float32))
    oreq = {"x": O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(0.2)))}
    ref = parity.oracle_mh_sweep(oi, ost, oreq, parity.tracker_data(T), n_total, T, 11, extra=lambda t: (np.float32(t),))
    assert x.shape == (n_total, D) and np.array_equal(x, ref["x"][ref["anc"]])
    assert abs(meta["log_ml"] - sum(ref["terms"])) < 1e-9 * max(1.0, abs(sum(ref["terms"])))
    assert meta["reruns"] == (1 if capacity else 0), meta


@pytest.mark.parametrize("world,capacity,kind", [(2, 0, "systematic"), (4, 5, "systematic"), (2, 0, "multinomial_sorted"),
                                                 (4, 5, "multinomial_sorted"), (8, 0, "systematic"), (8, 5, "systematic")])
def test_sharded_importancek_global_resample_equals_oracle(tmp_path, world, capacity, kind):
    """BASELINE config 4 sharded (8-schools ImportanceK, ONE global resample of a 10-latent trace — systematic, or the
    sorted multinomial against the table of all K slots): the concatenated ranks equal the single-process oracle, for
    any rank count."""
    k_total = 8192 if world == 8 else 4096
    out = str(tmp_path / "schools")
    r = _launch(world, [out, str(k_total // world), "0", str(capacity), "schools"],
                extra_env={"GENMI_TEST_OPTS": json.dumps({"resample": kind})})
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out + ".npz")
    sig = np.array(parity.SCHOOL_SIGMA, np.float32)

    @O.gen
    def o_schools():
        mu = O.normal(0.0, 5.0) @ "mu"
        log_tau = O.normal(0.0, 1.0) @ "log_tau"
        theta = O.normal(mu[..., None] * np.ones(8, np.float32), O.exp(log_tau)[..., None] * np.ones(8, np.float32)) @ "theta"
        _ = O.normal(theta, sig) @ "y"
        return theta
    oc = O.ImportanceK(O.Target(o_schools, (), O.C.d({"y": parity.SCHOOL_Y})), k_total).run_smc(O.key(2))
    assert np.array_equal(got["lw"], oc.get_log_weights())
    cdf, total, M, shift = O.weight_cdf(oc.get_log_weights())
    anc = O.ancestors_of_kind(O.MULTINOMIAL_SORTED if kind == "multinomial_sorted" else O.SYSTEMATIC,
                              O.split(O.key(2))[0], cdf)                 # the algorithm's leftover key resamples
    assert np.array_equal(got["theta"], oc.get_particles().get_choices()["theta"][anc])
    assert np.array_equal(got["mu"], oc.get_particles().get_choices()["mu"][anc])
    assert abs(float(got["log_ml"]) - float(oc.get_log_marginal_likelihood_estimate())) < 2e-5
    # ONE plan for the whole 10-latent trace: one all-gather (tile statistics), ONE all-to-all (every row packed),
    # one 8-byte all-reduce per capacity attempt — however many leaves the trace has
    info = json.load(open(out + ".info.json"))
    assert info["rows"] >= 10 and info["collectives"]["all_to_all"] == 1
    if kind == "systematic":
        assert info["form"] == "tile statistics"
        assert info["collectives"]["all_gather"] == 1 and 1 <= info["collectives"]["all_reduce_max"] <= 2


def test_slot_bounds_match_ancestors():
    """systematic_slot_bounds (host, exact integers) == counting the oracle's ancestors per mass interval"""
    from genjax_amd.inference.sharded import systematic_slot_bounds
    rng = np.random.default_rng(1)
    n = 3000
    lw = rng.normal(0, 2, n).astype(np.float32)
    cdf, total, _, _ = O.weight_cdf(lw)
    k = O.key(5)
    anc = O.ancestors(O.SYSTEMATIC, k, cdf)
    u0 = int(O.bits32(k, 0)) >> 9
    cuts = [0, 700, 1500, 2999, 3000]
    offs = [0] + [int(cdf[c - 1]) for c in cuts[1:]]
    b = systematic_slot_bounds(offs, total, n, u0)
    for (lo, hi), (s, e) in zip(zip(cuts[:-1], cuts[1:]), zip(b[:-1], b[1:])):
        assert e - s == int(np.sum((anc >= lo) & (anc < hi)))


@pytest.mark.parametrize("world,weak,bare", [(1, False, False), (2, False, False), (2, True, False), (2, False, True)])
def test_bench_contract_control_flow(world, weak, bare):
    """bench.py's N > 1 control flow (rendezvous on 127.0.0.1, barriers, max over ranks, exactly ONE JSON line on
    stdout from rank 0, communicator teardown after the line is out) — bench.py unchanged, started through
    tests/bench_on_cpu.py (gloo + the CPU mirror of the C-ABI); values are not timings.
    Default = STRONG scaling (BASELINE's metric): --particles is the TOTAL, rounded up to a multiple of world x 1024;
    --weak: --particles per GPU, rounded up to a multiple of 1024.
    bare: `python bench.py --gpus 2` with NO launcher and no WORLD_SIZE — what the driver runs: the process spawns its
    own ranks (bench.spawn_ranks) and relays rank 0's line."""
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    if bare:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--particles", "3000", "--T", "4", "--no-cpu-baseline",
            "--no-graph"] + (["--weak"] if weak else [])
    if world == 1 or bare:
        cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_on_cpu.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "bench_on_cpu.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in out
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == ("weak" if weak else "strong")
    if world == 1:
        per, total = 3000, 3000
    elif weak:
        per, total = 3072, 3072 * world
    else:
        per, total = 2048, 4096               # 3000 in total -> 2 x 2048 (a multiple of world x 1024)
    assert out["config"]["particles_per_gpu"] == per and out["config"]["particles_total"] == total
    if world > 1:
        assert str(per if weak else total) in out["config"]["workload"] and "communicator" in out["config"]
        # the OTHER scaling rides along: a strong line carries the weak record (3000 per GPU -> 3072), a weak line the strong one
        other = out["strong" if weak else "weak"]
        assert "error" not in other, other
        assert other["scaling"] == ("strong" if weak else "weak") and other["value"] > 0
        assert other["particles_per_gpu"] == (2048 if weak else 3072) and other["particles_total"] == other["particles_per_gpu"] * world
        assert abs(other["log_ml"] - out["log_ml_kalman"]) < 0.5
    assert out["value"] > 0 and abs(out["log_ml"] - out["log_ml_kalman"]) < 0.5
