#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

Workload (BASELINE config 2): linear-Gaussian state-space model, T = 100,
bootstrap SMC with 1e6 particles, systematic resampling every step.
One "step" of this harness = one whole sweep (N * T particle-steps), issued as
one hipGraph replay with every input already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU — either launched by torch.distributed.run, or bare: without WORLD_SIZE in the
   environment this process only spawns the N ranks (spawn_ranks; it never touches a GPU itself) and relays
   rank 0's JSON line)

N > 1 is STRONG scaling by default — BASELINE's metric: the same 1e6-particle sweep split over the N ranks
(contiguous blocks of global particle indices; the total is rounded up to a multiple of N x 1024 because a shard
starts on a tile of the integer CDF; the JSON names the exact count).  `--weak` keeps ~1e6 particles PER GPU instead.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the
site-program interpreter k_vm): algorithmic bytes per launch / its average
duration measured here with HIP events on the launch stream.  `cpu_baseline`
times the oracle's C statement of the same sweep on the host cores.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# cpu_baseline leg only: libgomp reads this when it is first loaded; a passive team pays a wake-up per parallel
# region (three per SMC step), which made the port's rate swing by 10x between runs
os.environ.setdefault("OMP_WAIT_POLICY", "ACTIVE")

N_PARTICLES = 1_000_000
T_STEPS = 100
VM_VALU_PER_WAVE = 325.65           # VALU instructions per 64 particles: SQ_INSTS_VALU / waves / 4 particles per
                                    # thread (1302.6 per wave, profiles/r02c_pmc_summary.txt; 1438.6 before the
                                    # one-instruction DPP scans / integer fixed-point weights, 1496.7 at the end of round 1)
# the noise-ahead (two-stream) sweep, SQ_INSTS_VALU per wave of 256 particles (profiles/r02h_pmc_summary.txt):
NA_VALU_PER_WAVE = {"gmx_jit_background_kernel": 1135.8, "gmx_jit_kernel": 216.6, "k_offspring_tile": 540.8}
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9   # integer / unpacked-f32 vector instructions: 16 lanes per clock per SIMD (a
                                            # wave64 instruction holds its SIMD for 4 cycles; only packed f32 math doubles
                                            # that).  tools/calib.hip on MI355X (profiles/r02_calib.txt): one Threefry-like
                                            # add/rotate/xor chain per lane 33.6 T lane-ops/s, two independent chains 35.5 T
                                            # — no gain from ILP, i.e. the pipe is full — against 39.3 T nominal at 2.4 GHz
VALU_CALIBRATED_LANE_OPS = 35.5e12
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# Algorithmic bytes per particle-step (8*D + 24 = 32 B, D = 1):
#   site program  ancestor 4 + gathered state 4 in, state 4 + log-weight 4 out    = 16 B
#   CDF           log-weight 4 in, CDF 4 out                                       =  8 B
#   offspring     CDF 4 in, ancestor 4 out                                         =  8 B
# (what runs: the CDF is never materialised — the site program leaves two numbers per 1024-particle tile and
#  k_offspring_tile rebuilds its tile's CDF in registers from the log-weights: 24 B of actual traffic)
VM_BYTES_PER_PARTICLE = 16
SWEEP_BYTES_PER_PARTICLE_STEP = 32


def cpu_baseline(n, T, ys, seed, budget_s=25.0):
    """Oracle 'port' timed on this box's host cores (bounded sample)."""
    src_dir = os.path.join(ROOT, "oracle")
    so = None
    try:     # native build for the cores it is timed on
        tmp = tempfile.mkdtemp(prefix="orc_")
        so = os.path.join(tmp, "liborc_sweep.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=c11",
                               "-ffp-contract=off", "-fno-fast-math", os.path.join(src_dir, "orc_sweep.c"),
                               "-o", so, "-lm"], stderr=subprocess.DEVNULL)
    except Exception:
        so = os.path.join(src_dir, "_build", "liborc_sweep.so")
        if not os.path.exists(so):
            return None
    import numpy as np
    lib = ctypes.CDLL(so)
    cores = int(lib.orc_threads())
    # the cores this process may actually use: affinity mask and cgroup quota (a 128-thread team on a box that
    # grants fewer CPUs spends its time in barriers: measured 1.7e7 instead of 1e8 particle-steps/s)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota[0] != "max":
            cores = max(1, min(cores, int(int(quota[0]) / int(quota[1]))))
    except Exception:
        pass
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except Exception:
        pass
    from oracle.genjax_oracle import cdf_shift
    shift = cdf_shift(n)
    f32, u64, i32 = np.float32, np.uint64, np.int32
    x, x2, lw = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, f32)
    cdf, anc = np.zeros(n, u64), np.zeros(n, i32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def run(Tn):
        maxs, totals = np.zeros(Tn, f32), np.zeros(Tn, u64)
        t0 = time.perf_counter()
        rc = lib.orc_lgssm_sweep(ctypes.c_int64(n), ctypes.c_int64(Tn), P(ys), ctypes.c_uint32(0),
                                 ctypes.c_uint32(seed), ctypes.c_float(0.9), ctypes.c_float(0.5),
                                 ctypes.c_float(1.0), ctypes.c_float(1.0), ctypes.c_int(shift), P(x), P(x2),
                                 P(lw), P(cdf), P(anc), P(maxs), P(totals))
        assert rc == 0
        return time.perf_counter() - t0
    t_probe = run(2)                          # 2 steps to size the sample
    Tn = int(max(2, min(T, budget_s / max(t_probe / 2, 1e-6))))
    dt = run(Tn)
    out = {"value": n * Tn / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
           "sample": f"{Tn} of {T} SMC steps x {n} particles, oracle/orc_sweep.c (OpenMP, gcc -O3 -march=native)"}
    try:        # the same port on ONE core (SURVEY.md 8d asks for both), a few steps only
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        t1 = run(2)
        T1 = int(max(2, min(T, 4.0 / max(t1 / 2, 1e-6))))
        out["one_core"] = {"value": n * T1 / run(T1), "sample": f"{T1} steps"}
        gomp.omp_set_num_threads(cores)
    except Exception as e:
        out["one_core"] = {"error": repr(e)}
    # the reference's own backend, if this box happens to have it (it never travels with the repo): a jax.numpy
    # restatement of the same sweep written by this build (tools/jax_cpu_restatement.py)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import jax_cpu_restatement as jr
        if jr.available():
            out["jax_cpu"] = jr.time_sweep(n, T, ys, seed, budget_s=15.0)
        else:
            out["jax_cpu"] = "jax not importable on this box: the reference's jax[cpu] path cannot be timed here"
    except Exception as e:
        out["jax_cpu"] = {"error": repr(e)}
    return out


def spawn_ranks(world: int) -> int:
    """`python bench.py --gpus N` with no launcher: THIS process becomes the launcher.  It never imports torch, never
    loads the HIP library and never touches a GPU; it starts N children of the same command line — one rank per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, exactly what
    `torch.distributed.run --nproc-per-node N` would set — relays rank 0's stdout (the ONE JSON line) and returns
    non-zero if any rank does.  Children are started as child processes (never exec'd over a process that has
    initialised the GPU) and are ended by their exact PIDs if one of them fails."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.environ.get("GENMI_BENCH_ENTRY", os.path.abspath(__file__))     # tests: tests/bench_on_cpu.py
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GENMI_BENCH_SPAWNED="1")
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is this process's stdout; the other ranks print nothing there (their fd 1 -> stderr)
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {procs.index(p)} exited with status {code}; ending the other ranks\n")
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
            if rc != 0:
                deadline = time.time() + 20
                while any(q.poll() is None for q in live) and time.time() < deadline:
                    time.sleep(0.1)
                for q in live:
                    if q.poll() is None:
                        q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=N_PARTICLES)
    ap.add_argument("--T", type=int, default=T_STEPS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="use the multi-GPU code path even at world size 1")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: --particles PER GPU (weak scaling) instead of in total (strong scaling, the default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: spawn the ranks BEFORE anything here imports torch or touches the GPU
        raise SystemExit(spawn_ranks(args.gpus))

    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 (the RCCL
    # banner librccl prints on communicator creation, library chatter) is sent to stderr
    import ctypes
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE={world}: they must agree")
    # The C-ABI backend decides where this runs: genjax_amd._lib.get() loads the HIP library and fails loudly
    # without a GPU.  (tests/bench_on_cpu.py installs the tests' CPU mirror of the C-ABI BEFORE running this file,
    # to exercise the launch / barrier / reporting logic under gloo; nothing here knows about it.)
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist_
        dist = dist_
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import genjax_amd as G
    from genjax_amd import _lib, workloads
    from genjax_amd.inference.smc import BootstrapSweep
    be = _lib.get()
    n, T = args.particles, args.T
    seed = 314159
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    single = world == 1 and not args.sharded
    requested = n
    if world > 1:
        # shards start on a 1024-particle tile of the global CDF (include/genmi.h "Resampling"), so that the N-GPU
        # population is bit-identical to a single-process one of the same size.
        #   strong (default): --particles in TOTAL, rounded up to a multiple of world x 1024 (1e6 -> 1 001 472 at 2,
        #                     1 003 520 at 4, 1 007 616 at 8 ranks); n = the per-rank share
        #   --weak:           --particles PER GPU, rounded up to a multiple of 1024 (1e6 -> 1 000 448)
        if args.weak:
            n = ((n + 1023) // 1024) * 1024
        else:
            n = ((n + world * 1024 - 1) // (world * 1024)) * 1024
    if single:
        sw = BootstrapSweep(init, step, n, T).prepare(G.key(seed), torch.from_numpy(ys))
        if not args.no_graph:
            sw.capture()
        launch = sw.launch
    else:
        from genjax_amd.inference.sharded import ShardedBootstrapSweep
        sw = ShardedBootstrapSweep(init, step, n, T, dist, always_communicate=True).prepare(
            G.key(seed), torch.from_numpy(ys))

        # the whole sharded sweep — kernels, both streams of the noise-ahead form AND the RCCL collectives (issued by
        # comm.RcclComm on the kernels' own stream) — as one hipGraph, like the single-GPU sweep.  Not with the
        # torch.distributed fallback communicator (its watchdog queries events recorded inside a capture), not on
        # the CPU mirror; GENMI_SHARDED_GRAPH=0 keeps eager launches.
        if (on_gpu and not args.no_graph and os.environ.get("GENMI_SHARDED_GRAPH", "1") != "0"
                and (sw.cx is None or sw.cx.graph_safe)):
            sw.capture()

        def launch():
            sw.launch()
            sw.finish()          # the once-per-sweep overflow check (one sync, one 8-byte all-reduce)

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        launch()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=be.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_particles = n * world
    value = args.steps * total_particles * T / dt
    log_ml = sw.log_ml()
    kal = workloads.kalman_log_ml(ys)

    out = {
        "metric": "particles/sec (particle-steps/s), bootstrap SMC sweep, linear-Gaussian SSM T=100",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak" if (args.weak and world > 1) else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 2: linear-Gaussian state-space (T=100), bootstrap SMC, "
                               "systematic resampling every step"
                               + ("" if world == 1 else
                                  (f"; {requested} particles per GPU requested (weak scaling), {n} run (a multiple of 1024)"
                                   if args.weak else
                                   f"; {requested} particles in total requested, {total_particles} run "
                                   f"(a multiple of {world} x 1024: shards start on a tile of the integer CDF)")),
                   "particles_per_gpu": n, "particles_total": total_particles, "T": T,
                   "resampler": "systematic", "graph": not args.no_graph and single,
                   "path": ("BootstrapSweep (hipGraph" + (", noise ahead on a second stream)" if getattr(sw, "noise_ahead", False)
                                                         else ")")) if single else
                   ("ShardedBootstrapSweep (RCCL" + (", noise ahead on a second stream" if getattr(sw, "noise_ahead", False) else "") + ")"),
                   "key": seed},
        "log_ml": log_ml, "log_ml_kalman": kal, "log_ml_abs_err": abs(log_ml - kal),
        "log_ml_rel_err": abs(log_ml - kal) / abs(kal),
    }

    if not single:
        out["config"]["communicator"] = sw.cx.name
        out["config"]["GENMI_COMM"] = os.environ.get("GENMI_COMM", "(unset: rccl on a GPU box, torch.distributed otherwise)")
        out["config"]["graph"] = sw.graph is not None
        out["config"]["capacity_per_peer"] = sw.capacity
        out["config"]["full_capacity_reruns"] = sw.reruns

    if rank == 0 and on_gpu:
        # ---- per-kernel durations, HIP events on the launch stream ----
        from ctypes import c_float, c_void_p
        timer = c_void_p()
        be.check(be.c.gmx_timer_create(timer), "timer")

        def time_launches(fn, reps=100):
            """Average duration of one launch of `fn`: `reps` back-to-back launches captured
            into a hipGraph (so the host is out of the loop) and timed with HIP events on the
            launch stream.  Includes the dependent-launch boundary (~1.5 us)."""
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
                side.synchronize()
                be.check(be.c.gmx_capture_begin(be.stream()), "capture")
                try:
                    for _ in range(reps):
                        fn()
                finally:
                    g = c_void_p()
                    rc = be.c.gmx_capture_end(be.stream(), g)
                be.check(rc, "capture_end")
                be.check(be.c.gmx_graph_launch(g, be.stream()), "graph")     # warm
                side.synchronize()
                be.check(be.c.gmx_timer_start(timer, be.stream()), "timer")
                for _ in range(5):
                    be.check(be.c.gmx_graph_launch(g, be.stream()), "graph")
                be.check(be.c.gmx_timer_stop(timer, be.stream()), "timer")
                ms = c_float()
                be.check(be.c.gmx_timer_elapsed_ms(timer, ms), "timer")
                be.c.gmx_graph_destroy(g)
            torch.cuda.current_stream().wait_stream(side)
            return ms.value * 1e3 / (5 * reps)          # us per launch
        kt = sw.kernel_timers()
        us = {name: time_launches(fn) for name, fn in kt.items()}
        vm_us = us["k_vm"]
        one = torch.zeros((2,), dtype=torch.float32, device="cuda")
        gap = time_launches(lambda: be.check(be.c.gmx_reduce_max(be.ptr(one), 1, be.ptr(one[1:]), be.stream()),
                                             "gmx_reduce_max"))
        us["launch_boundary"] = gap                               # a chain of trivial launches, for reference
        traffic = traffic_noise = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, traffic_noise = tj.get("k_vm_hbm_bytes_per_launch"), tj.get("noise_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        specialised = bool(sw.p_step.comp._be.c.gmx_program_is_specialized(sw.p_step.comp.handle))
        sweep_hbm_frac = SWEEP_BYTES_PER_PARTICLE_STEP * value / world / 1e9 / HBM_PEAK_GBS
        if single and getattr(sw, "noise_ahead", False):
            # ---- the two-stream (noise-ahead) sweep: three kernels per step, two of them concurrent ----
            # Durations: each kernel alone, back to back on one stream (HIP events); the whole sweep; the chain alone
            # (noise launches left out: the site programs then read stale noise — same work, same traffic).
            # Kernels overlap inside the sweep, so "x in the sweep" is no longer a difference of two sweeps: the
            # in-sweep per-kernel averages are rocprofv3's (profiles/*_kernel_stats.csv of this same command).
            us["sweep"] = time_launches(lambda: sw.enqueue(), reps=1)
            us["chain_only_sweep"] = time_launches(lambda: sw._enqueue_noise_ahead(skip_noise=True), reps=1)
            waves = (n + 1023) // 1024 * 4
            lane_ops = {k: v * 64 * waves for k, v in NA_VALU_PER_WAVE.items()}          # per launch
            step_us = us["sweep"] / T
            noise_rate = lane_ops["gmx_jit_background_kernel"] / (us["k_noise"] * 1e-6)
            sweep_rate = sum(lane_ops.values()) / (step_us * 1e-6)
            achieved = VM_BYTES_PER_PARTICLE * n / (vm_us * 1e-6) / 1e9
            out["roofline"] = {
                # The dominant kernel by GPU time is the noise program (gmx_jit_background_kernel, ~40 %): keys in,
                # 4 bytes per particle out, ZERO algorithmic bytes ("RNG contributes 0 B") — it is pure
                # integer / f32 vector work, so it is priced against the vector-issue peak, not against HBM (the
                # schema's "hbm" | "mfma" has no name for that; "valu" says what it is).
                "bound": "valu", "kernel": "gmx_jit_background_kernel (noise program: 3 Threefry-2x32 blocks + erf_inv "
                                           "per draw; BootstrapSweep noise-ahead form)",
                "achieved": noise_rate / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s",
                "frac": noise_rate / VALU_PEAK_LANE_OPS, "traffic": traffic_noise,
                "limiter": "vector-instruction issue (16 lanes/clk/SIMD for integer and unpacked f32): the whole sweep "
                           "keeps the vector ALUs busy for sweep.valu_frac of the time",
                "valu_source": "instructions per wave: SQ_INSTS_VALU of profiles/r02h_pmc_summary.txt (constants in "
                               "bench.py, not measured in this run); durations: HIP events in this run",
                "traffic_source": "profiles/traffic.json (rocprofv3 TCC passes of an earlier run of this workload, "
                                  "calibrated against copy kernels; NOT measured in this run); the noise program writes "
                                  "4 B per particle and reads nothing",
                "calibrated_ceiling_T_lane_ops": VALU_CALIBRATED_LANE_OPS / 1e12,
                # the data-path kernels against HBM (algorithmic bytes per launch / isolated launch duration)
                "hbm": {"gmx_jit_kernel": {"algorithmic_bytes_per_launch": VM_BYTES_PER_PARTICLE * n,
                                           "traffic": traffic,
                                           "achieved_GBps": achieved, "frac": achieved / HBM_PEAK_GBS,
                                           "note": "site program without its draws: ancestor 4 + gathered state 4 in, "
                                                   "state 4 + log-weight 4 out (+ 4 B of noise read, not algorithmic)",
                                           "specialised": specialised},
                        "k_offspring_tile": {"algorithmic_bytes_per_launch": 8 * n,
                                             "achieved_GBps": 8 * n / (us["k_offspring_tile"] * 1e-6) / 1e9
                                             if "k_offspring_tile" in us else None,
                                             "frac": 8 * n / (us["k_offspring_tile"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                                             if "k_offspring_tile" in us else None},
                        "peak_GBps": HBM_PEAK_GBS},
                "sweep": {"us_per_step": step_us, "chain_only_us_per_step": us["chain_only_sweep"] / T,
                          "valu_insts_per_wave_per_step": sum(NA_VALU_PER_WAVE.values()),
                          "valu_lane_ops_per_s": sweep_rate, "valu_frac": sweep_rate / VALU_PEAK_LANE_OPS,
                          "valu_frac_of_calibrated_ceiling": sweep_rate / VALU_CALIBRATED_LANE_OPS,
                          "hbm_frac": sweep_hbm_frac},
                "kernel_us": us,
                "sweep_frac_of_hbm_roofline": sweep_hbm_frac}
        else:
            if single:
                # the site program's duration IN the sweep: (sweep) - (sweep without its T launches), both
                # as hipGraphs timed with HIP events; this is the figure rocprofv3's per-kernel average
                # reproduces (the isolated back-to-back figure above runs ~10 % shorter: warm caches)
                full = time_launches(lambda: sw.enqueue(), reps=1) * 1.0
                rest = time_launches(lambda: sw.enqueue(skip_vm=True), reps=1) * 1.0
                us["sweep"] = full
                us["sweep_without_k_vm"] = rest
                # What rocprofv3 reports as this kernel's duration: in its kernel trace of a graph replay consecutive
                # kernels abut (median end -> next start = 0 ns, profiles/r01_o_kernel_stats.csv's trace), i.e. a
                # kernel's span includes its dependent-launch ramp.  So the roofline uses the marginal cost as is.
                us["k_vm_in_sweep"] = (full - rest) / T
                us["k_vm_in_sweep_minus_launch_boundary"] = (full - rest) / T - gap
                vm_us = us["k_vm_in_sweep"]
            achieved = VM_BYTES_PER_PARTICLE * n / (vm_us * 1e-6) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               # `bound` names the roofline `achieved` / `peak` are priced against (the schema has "hbm" |
                               # "mfma").  What actually limits this kernel is vector-instruction ISSUE: see `valu` below.
                               "limiter": "VALU issue: integer Threefry-2x32 (3 blocks per draw) at 16 lanes/clk/SIMD; "
                                          "HBM traffic equals the algorithmic bytes (no re-reads)",
                               "traffic_source": "profiles/traffic.json (rocprofv3 TCC pass of an earlier run of this "
                                                 "workload, calibrated against copy kernels; NOT measured in this run)",
                               "kernel": "gmx_jit_kernel (site program specialised from k_vm)" if specialised else "k_vm<gmx_regs_vgpr<16>, false>",
                               "algorithmic_bytes_per_launch": VM_BYTES_PER_PARTICLE * n,
                               # the kernel is VALU-issue bound, not HBM bound: 3 Threefry-2x32 blocks per draw
                               # (split child, fold_in, bits).  SQ_INSTS_VALU per wave from profiles/*_pmc_summary.txt.
                               "valu": {"source": "insts_per_64_particles: SQ_INSTS_VALU of profiles/r02c_pmc_summary.txt "
                                                  "(a constant in bench.py, not measured in this run); peak_int: "
                                                  "tools/calib.hip on this part (profiles/r02_calib.txt)",
                                        "insts_per_64_particles": VM_VALU_PER_WAVE,
                                        "lane_ops_per_s": VM_VALU_PER_WAVE * n / (vm_us * 1e-6),
                                        "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                                        "frac": VM_VALU_PER_WAVE * n / (vm_us * 1e-6) / VALU_PEAK_LANE_OPS,
                                        "calibrated_ceiling_lane_ops_per_s": VALU_CALIBRATED_LANE_OPS,
                                        "frac_of_calibrated_ceiling": VM_VALU_PER_WAVE * n / (vm_us * 1e-6) / VALU_CALIBRATED_LANE_OPS,
                                        "note": "peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz: integer (Threefry) and unpacked f32 "
                                                "instructions issue at 16 lanes/clk/SIMD on gfx950; the calibrated ceiling is what "
                                                "dependent add/rotate/xor chains sustain on this part (tools/calib.hip)"},
                               "kernel_us": us,
                               "sweep_frac_of_hbm_roofline": sweep_hbm_frac}
        be.c.gmx_timer_destroy(timer)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n, T, ys, seed)
            except Exception as e:                              # report, never fail the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
    if not on_gpu:
        out["data"] = "synthetic; NOT A MEASUREMENT (no GPU: the C-ABI was not the HIP library)"
    if dist is not None:
        dist.barrier()           # rank 0's kernel timing is done before anyone tears the communicator down
    # The JSON line goes out BEFORE the teardown: whatever a communicator does while it is destroyed, the
    # measurement is already on stdout.
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)       # C stdio buffers (the banner) go where fd 1 points NOW: stderr
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)                            # teardown chatter again to stderr
    if dist is not None:
        if hasattr(sw, "close") and getattr(sw, "cx", None) is not None:
            sw.close()           # graph first, then the RCCL communicator (the other order hangs)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
