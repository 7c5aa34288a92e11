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

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel — the hiprtc-specialised site program of the
SMC step (gmx_jit_kernel; the interpreter k_vm only when GENMI_JIT=0): algorithmic bytes per launch / its average
duration, measured here with HIP events on the launch stream, plus its VALU-issue fraction (the limiter that binds:
`roofline.valu`).  `cpu_baseline` times the oracle's C statement of the same sweep on the host cores.  The other
BASELINE configs (3: nonlinear SSM + MH, 4: 64-d mixture importance, 5: mixture Gibbs sweep) ride along in
`config.other_configs`, each with its own valu record.
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
VALU_PEAK_LANE_OPS = 256 * 4 * 64 * 2.4e9 / 2.0   # 78.6 T lane-ops/s: TWO vector instructions per 4-cycle slot of a SIMD — the
                                            # ceiling no instruction stream exceeds.  Calibrated with inline-asm streams the
                                            # compiler cannot fold or pack (tools/calib_valu_gen.py, profiles/r06_calib.txt):
                                            # f32 add / mul / fma / mov and simple integer ops (xor, and, or, add_u32, lshr)
                                            # retire one per ~2.2 cycles per SIMD with >= 2 waves (66-70 T measured: the
                                            # clock sags to 2.15-2.37 GHz); v_alignbit, shifts left, min / max, cvt, cmp,
                                            # cndmask, mul_lo, add3, anything with an SGPR source: one per 4.07 cycles; and
                                            # an integer instruction next to such a one (a Threefry round: add, alignbit,
                                            # xor) takes a whole slot too: 3.97-4.03 cycles per instruction, 38 T lane-ops/s.
                                            # Round 5's "39.3 T nominal peak" was the one-per-slot rate: config 4's kernel
                                            # exceeded it (47.8 T) because its f32 arithmetic rides beside the Threefry.
VALU_SLOT_CYCLES = 4.07                     # profiles/r06_calib.txt


def slot_model(kernel_key_prefix: str):
    """cycles per vector instruction [lo, hi] the SLOT model gives a kernel of this instruction mix (tools/valu_model.py over
    its offline disassembly, profiles/r06_valu_classes.json): lo = every simple-integer instruction finds a partner in its
    slot, hi = none does (what Threefry-interleaved code measures)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_classes.json")))
    except (OSError, ValueError):
        return None
    for k, v in d.items():
        if isinstance(v, dict) and k.startswith(kernel_key_prefix):
            return {"classes": {c: v[c] for c in ("F", "I", "X", "T")}, "cycles_per_inst": [v["cycles_per_inst_lo"], v["cycles_per_inst_hi"]],
                    "source": k}
    return None


def slot_frac(rate_lane_ops: float, sm):
    """the fraction [lo, hi] of the slot model's bound a measured instruction rate reaches: rate x cycles-per-instruction /
    (SIMDs x 64 lanes x 2.4 GHz) — 1.0 = the SIMDs' slots are full for this instruction mix"""
    if sm is None or rate_lane_ops is None:
        return None
    cap = 256 * 4 * 64 * 2.4e9
    return [rate_lane_ops * c / cap for c in sm["cycles_per_inst"]]


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


def _file_sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def code_identity(sw):
    """What code the numbers of this run were taken on: the FNV-1a hash of every hiprtc code object the sweep launches
    (gmx_program_code_hash) and the sha256 of the AOT library (k_offspring_tile and the other hand-written kernels).
    tools/prof_summary.py records the same identities next to the counters it extracts, and profile-sourced
    figures (instructions per wave, HBM bytes per launch, in-sweep durations) are used ONLY when they match."""
    from genjax_amd import _lib
    be = _lib.get()
    ids = {"libgenmi_hip.so": _file_sha16(_lib.LIB_PATH)}
    progs = {"gmx_jit_kernel": getattr(sw, "p_step", None)}
    for plist in getattr(sw, "_noise_progs", {}).values():
        for _root, q, _idx in plist:
            progs.setdefault("gmx_jit_background_kernel", q)
    if getattr(sw, "noise_ahead", False) and sw._noise_progs.get(id(sw.p_step)):
        progs["gmx_jit_background_kernel"] = sw._noise_progs[id(sw.p_step)][0][1]
    for name, p_ in progs.items():
        if p_ is not None:
            ids[name] = "%016x" % int(be.c.gmx_program_code_hash(p_.comp.handle))
    return ids


def load_profile_counters(ids):
    """profiles/counters.json (tools/prof_summary.py --json, one tools/reproduce.sh run): per-kernel SQ / TCC counters
    and rocprofv3 in-sweep durations, each with the identity of the code it was measured on.  Returns
    {kernel: {...}} for the kernels whose identity matches this run's, and a note for those that do not."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return {}, {"file": None}
    try:
        cj = json.load(open(path))
    except Exception as e:
        return {}, {"file": "profiles/counters.json", "error": repr(e)}
    ok, notes = {}, {"file": "profiles/counters.json", "tag": cj.get("tag"), "commit": cj.get("commit"), "stale": []}
    for k, d in cj.get("kernels", {}).items():
        short = "k_offspring_tile" if "k_offspring_tile" in k else k
        want = ids.get(short if short in ids else "libgenmi_hip.so")
        if d.get("code_id") is not None and d.get("code_id") == want:
            ok[short] = d
        else:
            notes["stale"].append(short)
    return ok, notes


def measure_roofline(be, sw, n, T, world, single, value):
    """The `roofline` object.  Top level: the WHOLE SWEEP against HBM — algorithmic bytes (32 B per
    particle-step) / the driver-visible time of the timed region / 8 TB/s.  `kernels`: each kernel of the step with its
    algorithmic bytes per launch over (a) its isolated launch duration, (b) its duration inside the dependent chain —
    both measured HERE with HIP events on the launch stream — and (c) rocprofv3's in-sweep average, when
    profiles/counters.json was taken on this very code.  `valu`: vector-instruction issue, the limiter of this path
    (instructions per wave are SQ_INSTS_VALU of the same profile, used only under the same identity check)."""
    import torch
    from ctypes import c_float, c_void_p
    timer = c_void_p()
    be.check(be.c.gmx_timer_create(timer), "timer")

    def time_launches(fn, reps=100):
        """Average duration of one launch of `fn`: `reps` back-to-back launches captured into a hipGraph (so the host
        is out of the loop) and timed with HIP events on the launch stream.  Includes the dependent-launch boundary."""
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

    ids = code_identity(sw)
    prof, prof_notes = load_profile_counters(ids)
    us = {name: time_launches(fn) for name, fn in sw.kernel_timers().items()}
    one = torch.zeros((2,), dtype=torch.float32, device="cuda")
    us["launch_boundary"] = time_launches(lambda: be.check(be.c.gmx_reduce_max(be.ptr(one), 1, be.ptr(one[1:]), be.stream()),
                                                           "gmx_reduce_max"))
    specialised = bool(be.c.gmx_program_is_specialized(sw.p_step.comp.handle))
    na = single and getattr(sw, "noise_ahead", False)
    sweep_gbs = SWEEP_BYTES_PER_PARTICLE_STEP * value / world / 1e9          # per GPU
    kern = {}

    def kernel_entry(name, alg_bytes, iso_us, chain_us=None, note=None):
        e = {"algorithmic_bytes_per_launch": alg_bytes, "us_isolated": iso_us}
        if alg_bytes:
            e["GBps_isolated"] = alg_bytes / (iso_us * 1e-6) / 1e9
            e["frac_isolated"] = e["GBps_isolated"] / HBM_PEAK_GBS
        if chain_us is not None:
            e["us_in_chain"] = chain_us
            if alg_bytes:
                e["frac_in_chain"] = alg_bytes / (chain_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        p_ = prof.get(name)
        if p_ is not None:
            if p_.get("avg_us_in_sweep"):
                e["us_in_sweep_rocprofv3"] = p_["avg_us_in_sweep"]
                if alg_bytes:
                    e["frac_in_sweep_rocprofv3"] = alg_bytes / (p_["avg_us_in_sweep"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            e["traffic"] = p_.get("hbm_bytes_per_launch")
            e["valu_insts_per_wave"] = p_.get("valu_per_wave")
            e["wait_any_over_wave_cycles"] = p_.get("wait_any_frac")
        else:
            e["traffic"] = None
            e["valu_insts_per_wave"] = None
        if note:
            e["note"] = note
        kern[name] = e
        return e

    if single:
        us["sweep"] = time_launches(lambda: sw.enqueue(), reps=1)
    fuse = single and bool(getattr(sw, "fuse", False))
    if na and fuse:
        # ONE chain launch per step: [resample step t-1 ; site program of step t] (gmx_run_args.rs)
        us["chain_only_sweep"] = time_launches(lambda: sw._enqueue_noise_ahead(skip_noise=True), reps=1)
        kernel_entry("gmx_jit_kernel", (VM_BYTES_PER_PARTICLE + 8) * n, us["k_vm"], us["chain_only_sweep"] / T,
                     "ONE launch per step: the previous step's resampling (log-weight 4 in, ancestor 4 out) as the "
                     "prologue of the site program (ancestor 4 + gathered state 4 in, state 4 + log-weight 4 out; + 4 B of "
                     "noise read, not algorithmic); specialised: %s" % specialised)
        kernel_entry("gmx_jit_background_kernel", 0, us["k_noise"], None,
                     "noise program: 3 Threefry-2x32 blocks + erf_inv per draw, second stream; zero algorithmic bytes "
                     "(SURVEY 8d: RNG contributes 0 B), writes 4 B per particle")
    elif na:
        us["chain_only_sweep"] = time_launches(lambda: sw._enqueue_noise_ahead(skip_noise=True), reps=1)
        us["chain_only_sweep_without_site_program"] = time_launches(
            lambda: sw._enqueue_noise_ahead(skip_vm=True, skip_noise=True), reps=1)
        vm_chain = (us["chain_only_sweep"] - us["chain_only_sweep_without_site_program"]) / T
        rs_chain = us["chain_only_sweep_without_site_program"] / T
        kernel_entry("gmx_jit_kernel", VM_BYTES_PER_PARTICLE * n, us["k_vm"], vm_chain,
                     "site program without its draws: ancestor 4 + gathered state 4 in, state 4 + log-weight 4 out "
                     "(+ 4 B of noise read, not algorithmic); specialised: %s" % specialised)
        if "k_offspring_tile" in us:
            kernel_entry("k_offspring_tile", 8 * n, us["k_offspring_tile"], rs_chain, "log-weight 4 in, ancestor 4 out")
        kernel_entry("gmx_jit_background_kernel", 0, us["k_noise"], None,
                     "noise program: 3 Threefry-2x32 blocks + erf_inv per draw, second stream; zero algorithmic bytes "
                     "(SURVEY 8d: RNG contributes 0 B), writes 4 B per particle")
    elif single and fuse:
        kernel_entry("gmx_jit_kernel", (VM_BYTES_PER_PARTICLE + 8) * n, us["k_vm"], us["sweep"] / T,
                     "ONE launch per step: the previous step's resampling as the prologue of the site program")
    elif single:
        us["sweep_without_k_vm"] = time_launches(lambda: sw.enqueue(skip_vm=True), reps=1)
        vm_chain = (us["sweep"] - us["sweep_without_k_vm"]) / T
        kernel_entry("gmx_jit_kernel" if specialised else "k_vm", VM_BYTES_PER_PARTICLE * n, us["k_vm"], vm_chain)
        for k_ in us:
            if k_.startswith(("k_offspring", "resample", "k_weight_cdf", "k_ancestors")):
                kernel_entry(k_, 8 * n, us[k_], None)
    else:
        kernel_entry("gmx_jit_kernel" if specialised else "k_vm", VM_BYTES_PER_PARTICLE * n, us["k_vm"], None)

    # ---- vector-instruction issue: the limiter ----
    waves = (n + 1023) // 1024 * 4
    sm = slot_model("gmx_jit_kernel (config 2 step")
    valu = {"peak_T_lane_ops": VALU_PEAK_LANE_OPS / 1e12,
            "peak_note": "two vector instructions per 4-cycle slot of each SIMD (256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 2): the "
                         "ceiling no stream exceeds, reached by f32 add / mul / fma and simple integer streams only "
                         "(profiles/r06_calib.txt: 66-70 T measured).  `slot_model`: what THIS instruction mix can reach — a "
                         "Threefry round (add, alignbit, xor) costs three whole slots",
            "slot_model": sm,
            "insts_per_wave_source": prof_notes}
    per_wave = {k: kern[k].get("valu_insts_per_wave") for k in kern}
    if single and all(v for v in per_wave.values()):
        step_us = us["sweep"] / T
        rate = sum(per_wave.values()) * 64 * waves / (step_us * 1e-6)
        valu.update(insts_per_wave_per_step=sum(per_wave.values()), lane_ops_per_s=rate,
                    valu_frac=rate / VALU_PEAK_LANE_OPS, slot_frac=slot_frac(rate, sm))
        if na:
            nrate = per_wave["gmx_jit_background_kernel"] * 64 * waves / (us["k_noise"] * 1e-6)
            valu["noise_program_isolated_valu_frac"] = nrate / VALU_PEAK_LANE_OPS
    else:
        valu["valu_frac"] = None
        valu["note"] = ("no instruction counts for this exact code (profiles/counters.json absent or taken on other "
                        "code: see insts_per_wave_source.stale) — not estimated")
    traffic = None
    if kern and all(kern[k].get("traffic") is not None for k in kern):
        traffic = sum(kern[k]["traffic"] for k in kern)             # HBM bytes per STEP (one launch of each kernel)
    be.c.gmx_timer_destroy(timer)
    return {
        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
        "achieved": sweep_gbs, "frac": sweep_gbs / HBM_PEAK_GBS, "traffic": traffic,
        "scope": "whole sweep per GPU: 32 algorithmic bytes per particle-step (SURVEY.md 8d) x particle-steps/s of the "
                 "timed region (the figure `value` is computed from); `traffic` = TCC bytes per STEP summed over the "
                 "step's kernels (vs 32 B x n algorithmic), from profiles/counters.json when taken on this code",
        "algorithmic_bytes_per_step": SWEEP_BYTES_PER_PARTICLE_STEP * n,
        "limiter": "the SIMDs' instruction slots for THIS instruction mix (3 Threefry-2x32 blocks per draw fixed by jax's key "
                   "tree: integer instructions next to v_alignbit take a slot each) and the chain's dependent round trips — not "
                   "HBM, and not the 2-per-slot issue ceiling: see valu.slot_frac against valu.valu_frac",
        "kernels": kern, "valu": valu, "kernel_us": us, "code_identity": ids,
        "us_per_step_event_timed": (us["sweep"] / T) if "sweep" in us else None,
        "chain_only_us_per_step_event_timed": (us["chain_only_sweep"] / T) if "chain_only_sweep" in us else None,
        "us_per_step_note": "HIP-event-timed replays of the captured sweep taken AFTER the timed region, in isolation; the "
                            "figure `value` / `ms_per_step` come from is the timed region itself (ms_per_step / T)",
    }


def config_workload(which: int):
    """(run, units per call, unit name) of BASELINE config 3 / 4 / 5 exactly as `other_configs` times it — shared with
    tools/run_config.py, which runs it under rocprofv3 for the instruction counts `other_configs` quotes."""
    import numpy as np
    import torch
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp, workloads
    from genjax_amd.inference import gibbs, smc
    if which == 3:
        n, T = N_PARTICLES, T_STEPS
        init, step = workloads.make_nlssm(G)
        req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
        sw = smc.BootstrapSweep(init, step, n, T, step_extra=lambda t: (float(t),), rejuvenate=req).prepare(
            G.key(7), torch.from_numpy(workloads.nlssm_data(T))).capture()
        return {"run": sw.launch, "units": T, "unit": "step", "sweep": sw, "n": n, "T": T, "warm_units": T}         # capture() runs one eager sweep
    if which == 4:
        sig = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
        ysch = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)

        @G.gen
        def schools():
            mu = G.normal(0.0, 5.0) @ "mu"
            log_tau = G.normal(0.0, 1.0) @ "log_tau"
            theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
            _ = G.normal(theta, jnp.array(sig)) @ "y"
            return theta
        k = 10_000_000
        alg = smc.ImportanceK(G.Target(schools, (), C["y"].set(ysch)), k_particles=k)
        box = {}

        def run4():
            c = alg.run_smc(G.key(2))
            r = smc.resample(G.key(3), c, "systematic")
            box["c"], box["theta"] = c, r.get_particles().get_choices()["theta"]     # materialises the gathered latents
        return {"run": run4, "units": 1, "unit": "run", "alg": alg, "box": box, "k": k, "warm_units": 0}
    if which == 5:
        n, K = 1_000_000, 64
        x, guess, probs, z = workloads.mixture_data(n, K)
        gd = workloads.make_mixture(G)
        args5 = (torch.from_numpy(probs).cuda(), torch.from_numpy(guess).cuda())
        chm = C["obs"].set(torch.from_numpy(x).cuda())
        box = {}

        def run5():
            box["idx"] = gibbs.gibbs_categorical(G.key(1), gd, args5, chm, "idx", K)
        return {"run": run5, "units": 1, "unit": "sweep", "box": box, "n": n, "K": K, "z": z, "gd": gd, "args": args5,
                "chm": chm, "warm_units": 0}
    raise ValueError(which)


def config_valu(name: str, seconds_per_unit: float, programs: str = None):
    """The vector-instruction issue of one of the other configs (they are ALU-bound: an HBM fraction says little):
    SQ_INSTS_VALU of every kernel of the workload per unit (step / run / sweep), from the rocprofv3 PMC pass
    tools/reproduce.sh took (profiles/counters.json `configs`, used only when taken on THIS code: the library's sha256
    AND engine.program_digest of the site programs the workload creates — a change on the Python side that alters a
    site program changes the specialised kernel under an unchanged library), over the unit's time measured in this run; plus the dominant kernel's name and its duration in that profile."""
    from genjax_amd import _lib
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        ent = json.load(open(path)).get("configs", {}).get(name)
    except Exception:
        ent = None
    if not ent:
        return {"valu_frac": None, "note": "no rocprofv3 counters for this config in profiles/counters.json"}
    lib = _file_sha16(_lib.LIB_PATH)
    if ent.get("lib") != lib:
        return {"valu_frac": None, "note": f"profiles/counters.json `{name}` was taken on another library build "
                                           f"({ent.get('lib')} != {lib}): not used"}
    if ent.get("programs") != programs:
        return {"valu_frac": None, "note": f"profiles/counters.json `{name}` was taken on other site programs "
                                           f"({ent.get('programs')} != {programs}): not used"}
    wave_insts = float(ent["valu_wave_insts_per_unit"])
    rate = wave_insts * 64.0 / seconds_per_unit
    sm = slot_model("gmx_jit_kernel (config 4") if name == "config4" else None
    return {"valu_wave_insts_per_unit": wave_insts, "unit": ent.get("unit"), "lane_ops_per_s": rate,
            "valu_frac": rate / VALU_PEAK_LANE_OPS, "slot_model": sm, "slot_frac": slot_frac(rate, sm),
            "dominant_kernel": ent.get("dominant_kernel"), "dominant_kernel_valu_insts_per_wave": ent.get("dominant_valu_per_wave"),
            "dominant_kernel_us_in_profile": ent.get("dominant_avg_us"), "kernels": ent.get("kernels"),
            "source": {"file": "profiles/counters.json", "tag": ent.get("tag"), "lib": lib}}


def other_configs():
    """BASELINE configs 3 / 4 / 5 on this GPU, AFTER and outside the headline's timed region (a few seconds in all),
    so that the driver-visible record carries them too.  Each with its SURVEY 8(d) algorithmic bytes against 8 TB/s.
    Synthetic data, random-init everything, same models / keys as the parity tests (tests/test_gpu_parity.py holds each
    of these workloads to the oracle bit for bit)."""
    import numpy as np
    import torch
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, engine, numpy as jnp, workloads
    from genjax_amd.inference import gibbs, smc
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    try:        # ---- config 3: nonlinear SSM, 1e6 particles, T = 100, one Gaussian-drift MH move per step ----
        with engine.program_digest() as progs:       # the site programs the counters of this config are held to
            engine.clear_caches()                    # ... every one of them created inside the block, as in a fresh process
            w3 = config_workload(3)
            sw, n, T = w3["sweep"], w3["n"], w3["T"]
            dt = timed(sw.launch, 5)
        b = 52 * n * T          # 8(d): 32 B bootstrap step + 20 B MH move per particle-step
        out["config3"] = {"workload": "nonlinear SSM + one Rejuvenate MH move per step, BootstrapSweep(rejuvenate=...), "
                                      "1e6 particles x 100 steps, one hipGraph", "us_per_step": 1e6 * dt / T,
                          "particle_steps_per_s": n * T / dt, "algorithmic_bytes_per_particle_step": 52,
                          "hbm_frac": b / dt / 1e9 / HBM_PEAK_GBS, "valu": config_valu("config3", dt / T, progs.hex()),
                          "log_ml": sw.log_ml(), "noise_ahead": bool(sw.noise_ahead), "one_launch_per_step": bool(sw.fuse),
                          "accept_rate_last_step": float(sw.accept.float().mean())}
        del sw, w3
    except Exception as e:
        out["config3"] = {"error": repr(e)[:300]}
    try:        # ---- config 4: 8-schools, ImportanceK k = 1e7 + one global systematic resample ----
        with engine.program_digest() as progs:
            engine.clear_caches()                    # ... every one of them created inside the block, as in a fresh process
            w4 = config_workload(4)
            alg, box, k = w4["alg"], w4["box"], w4["k"]
            dt = timed(w4["run"], 10)          # (ten back-to-back runs: three of them were mostly pipeline fill and drain)
        dti = timed(lambda: alg.run_smc(G.key(2)), 10)
        out["config4"] = {"workload": "8-schools ImportanceK k = 1e7 + one systematic resample + gather of theta",
                          "ms_total": 1e3 * dt, "ms_importance": 1e3 * dti, "particles_per_s": k / dt,
                          "algorithmic_bytes_per_particle": {"importance": 48, "resample_and_gather": 104},
                          "hbm_frac_importance": 48.0 * k / dti / 1e9 / HBM_PEAK_GBS,
                          "hbm_frac_total": 152.0 * k / dt / 1e9 / HBM_PEAK_GBS, "valu": config_valu("config4", dt, progs.hex()),
                          "log_ml": float(box["c"].get_log_marginal_likelihood_estimate())}
        box.clear()
        del w4
    except Exception as e:
        out["config4"] = {"error": repr(e)[:300]}
    try:        # ---- config 5: mixture, K = 64 clusters, 1e6 datapoints: one assignment sweep ----
        with engine.program_digest() as progs:
            engine.clear_caches()                    # ... every one of them created inside the block, as in a fresh process
            w5 = config_workload(5)
            n, K, z, gd, args5, chm, box = w5["n"], w5["K"], w5["z"], w5["gd"], w5["args"], w5["chm"], w5["box"]
            dt = timed(w5["run"], 5)
        # ... and THROUGH THE GFI: the datapoints as a `generate_datapoint.repeat(n=N)` plate called directly (its
        # elements on the launch axis), the sweep = gibbs.enumerative_gibbs on the plate's trace (the fused draw + the
        # plate's Update): the notebook's update_datapoint_assignment for a model written with the Vmap combinator
        plate = gd.repeat(n=n)
        tr5, _w5 = plate.importance(G.key(2), chm, args5)

        def run5p():
            box["tr"], box["idxp"], _ = gibbs.enumerative_gibbs(G.key(1), tr5, "idx", K)
        dtp = timed(run5p, 5)
        dts = timed(lambda: plate.simulate(G.key(3), args5), 5)
        out["config5"] = {"workload": "Dirichlet-categorical mixture, K = 64, 1e6 datapoints: one cluster-assignment "
                                      "sweep (gibbs_categorical: one launch, no [N, K] matrix)", "ms": 1e3 * dt,
                          "datapoints_per_s": n / dt, "gumbels_per_s": n * K / dt, "algorithmic_bytes_per_datapoint": 8,
                          "hbm_frac": 8.0 * n / dt / 1e9 / HBM_PEAK_GBS, "valu": config_valu("config5", dt, progs.hex()),
                          "note": "ALU-bound by design (64 Gumbels + 64 log-densities per datapoint; SURVEY 8d)",
                          "agrees_with_generating_component": float((box["idx"].cpu().numpy() == z).mean()),
                          "through_the_gfi": {
                              "workload": "generate_datapoint.repeat(n = 1e6) called directly (elements on the launch "
                                          "axis) + gibbs.enumerative_gibbs on its trace (fused draw + plate Update)",
                              "ms": 1e3 * dtp, "ratio_to_gibbs_categorical": dtp / dt,
                              "ms_plate_simulate": 1e3 * dts}}
    except Exception as e:
        out["config5"] = {"error": repr(e)[:300]}
    try:        # ---- config 2 under the other resampling schemes (the headline is systematic) ----
        n, T = N_PARTICLES, T_STEPS
        ys = workloads.lgssm_data(T)
        init, step = workloads.make_lgssm(G)
        kinds = {}
        for kind in ("stratified", "multinomial_sorted", "multinomial_tiled", "multinomial"):
            sw = smc.BootstrapSweep(init, step, n, T, resample=kind).prepare(G.key(314159), torch.from_numpy(ys)).capture()
            dt = timed(sw.launch, 3)
            kinds[kind] = {"us_per_step": 1e6 * dt / T, "particle_steps_per_s": n * T / dt, "log_ml": sw.log_ml()}
            del sw
        out["config2_resampling_kinds"] = dict(
            kinds, workload="BASELINE config 2 (1e6 particles x 100 steps, one hipGraph) with resample=<kind>; the three "
                            "multinomial forms share the offspring law and differ in slot order (iid / by tile / sorted)",
            kalman_log_ml=workloads.kalman_log_ml(ys))
    except Exception as e:
        out["config2_resampling_kinds"] = {"error": repr(e)[:300]}
    out["config2_sizes"] = config2_sizes()
    return out


CONFIG2_SIZES = (125_000, 250_000, 500_000, 1_000_000, 2_000_000, 8_000_000)


def config2_sizes():
    """BASELINE config 2 (linear-Gaussian SSM, T = 100, systematic resampling every step, one hipGraph) at six particle
    counts: where the step stops being launch- / latency-bound, and what a rank holding N / 8 particles of a sharded
    sweep has to work with (DESIGN.md §6's scaling budget).  us/step, the bytes fraction (32 algorithmic bytes per
    particle-step against 8 TB/s) and the vector-instruction issue fraction (profile-sourced instructions per wave, used
    only for the step form they were counted on: one launch per step, n <= 2^20).  Every size's log-ML is held to the
    C oracle bit for bit in tests/test_gpu_parity.py::test_config2_sizes_log_ml_bit_exact."""
    import torch
    import genjax_amd as G
    from genjax_amd import workloads
    from genjax_amd.inference import smc
    T = T_STEPS
    ys = workloads.lgssm_data(T)
    init, step = workloads.make_lgssm(G)
    out = {"workload": "BASELINE config 2 at N particles x 100 steps, one hipGraph per sweep; one launch per step up to "
                       "2^20 particles, two beyond", "kalman_log_ml": workloads.kalman_log_ml(ys), "sizes": {}}
    prof = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    except Exception:
        pass
    for n in CONFIG2_SIZES:
        try:
            sw = smc.BootstrapSweep(init, step, n, T).prepare(G.key(314159), torch.from_numpy(ys)).capture()
            sw.launch()
            torch.cuda.synchronize()
            reps = 5 if n <= 2_000_000 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                sw.launch()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            rec = {"us_per_step": 1e6 * dt / T, "particle_steps_per_s": n * T / dt,
                   "hbm_frac": SWEEP_BYTES_PER_PARTICLE_STEP * n * T / dt / 1e9 / HBM_PEAK_GBS,
                   "one_launch_per_step": bool(sw.fuse), "noise_ahead": bool(sw.noise_ahead), "log_ml": sw.log_ml()}
            valu = None
            if prof is not None and sw.fuse and sw.noise_ahead:
                ids = code_identity(sw)
                cnt, _notes = load_profile_counters(ids)
                pw = [cnt.get(k, {}).get("valu_per_wave") for k in ("gmx_jit_kernel", "gmx_jit_background_kernel")]
                if all(pw):
                    waves = (n + 1023) // 1024 * 4
                    valu = sum(pw) * 64 * waves / (dt / T) / VALU_PEAK_LANE_OPS
            rec["valu_frac"] = valu
            out["sizes"][str(n)] = rec
            del sw
        except Exception as e:      # noqa: BLE001
            out["sizes"][str(n)] = {"error": repr(e)[:300]}
    return out


def other_configs_sharded(dist, world, rank, cx, be):
    """BASELINE's two multi-GPU configs on the sharded path (every rank takes part; rank 0 reports), each beside the
    SAME run's single-GPU code path as its strong-scaling denominator (every rank runs that one on its own GPU at the
    same time; the slowest rank's time is quoted):
      config 3  nonlinear SSM + one MH move per step, 1e6 particles in total: ShardedBootstrapSweep(rejuvenate=...)
                over the communicator the headline chose, vs BootstrapSweep(rejuvenate=...) on one GPU
      config 4  8-schools ImportanceK k = 1e7 in total + ONE global systematic resample:
                sharded_importance_resample (k / N per rank), vs ImportanceK + smc.resample on one GPU"""
    import numpy as np
    import torch
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp, workloads
    from genjax_amd.inference import smc
    from genjax_amd.inference.sharded import ShardedBootstrapSweep, sharded_importance_resample
    out = {}

    def slowest(dt):
        t = torch.tensor([dt], device=be.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return slowest((time.perf_counter() - t0) / reps)

    try:
        T = T_STEPS
        n = ((N_PARTICLES + world * 1024 - 1) // (world * 1024)) * 1024
        init, step = workloads.make_nlssm(G)
        req = G.StaticRequest({"x": G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))})
        ysn = torch.from_numpy(workloads.nlssm_data(T))
        one = smc.BootstrapSweep(init, step, n * world, T, step_extra=lambda t: (float(t),), rejuvenate=req).prepare(
            G.key(7), ysn).capture()
        dt1 = timed(one.launch, 3)
        lm1 = one.log_ml()
        del one
        sh = ShardedBootstrapSweep(init, step, n, T, dist, always_communicate=True, rejuvenate=req,
                                   step_extra=lambda t: (float(t),), comm=cx).prepare(G.key(7), ysn)
        if cx is not None and cx.graph_safe and (world == 1 or getattr(cx, "fused", False)):
            sh.capture()

        def run3():
            sh.launch(); sh.finish()
        dtN = timed(run3, 3)
        out["config3"] = {"workload": f"nonlinear SSM + one Rejuvenate MH move per step, {n * world} particles x {T} steps",
                          "sharded_us_per_step": 1e6 * dtN / T, "single_gpu_us_per_step": 1e6 * dt1 / T,
                          "strong_scaling_speedup": dt1 / dtN, "particle_steps_per_s": n * world * T / dtN,
                          "log_ml_rel_diff_vs_single_gpu": abs(sh.log_ml() - lm1) / max(1.0, abs(lm1)), "graph": sh.graph is not None,
                          "communicator": getattr(cx, "name", None), "full_capacity_reruns": sh.reruns,
                          "one_launch_per_step": bool(getattr(sh, "fuse_sh", False) and getattr(sh, "chain_mh", False))}
        sh.close()
    except Exception as e:          # noqa: BLE001
        out["config3"] = {"error": repr(e)[:300]}
    try:
        sig = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
        ysch = np.array([28, 8, -3, 7, -1, 1, 18, 12], np.float32)

        @G.gen
        def schools():
            mu = G.normal(0.0, 5.0) @ "mu"
            log_tau = G.normal(0.0, 1.0) @ "log_tau"
            theta = G.normal(mu * jnp.ones(8), jnp.exp(log_tau) * jnp.ones(8)) @ "theta"
            _ = G.normal(theta, jnp.array(sig)) @ "y"
            return theta
        tgt = G.Target(schools, (), C["y"].set(ysch))
        kr = ((10_000_000 + world * 1024 - 1) // (world * 1024)) * 1024
        k = kr * world
        alg = smc.ImportanceK(tgt, k_particles=k)
        box = {}

        def run1():
            c = alg.run_smc(G.key(2))
            r = smc.resample(G.split(G.key(2))[0], c, "systematic")
            box["theta1"] = r.get_particles().get_choices()["theta"]
        dt1 = timed(run1, 8)
        stats = {}

        def runN():
            coll, _lw = sharded_importance_resample(tgt, kr, G.key(2), dist, comm=cx, stats=stats)
            box["thetaN"] = coll.get_particles().get_choices()["theta"]
        dtN = timed(runN, 8)
        same = bool(torch.equal(box["thetaN"], box["theta1"][rank * kr:(rank + 1) * kr]))
        out["config4"] = {"workload": f"8-schools ImportanceK k = {k} + one global systematic resample of the 10-latent trace",
                          "sharded_ms": 1e3 * dtN, "single_gpu_ms": 1e3 * dt1, "strong_scaling_speedup": dt1 / dtN,
                          "particles_per_s": k / dtN, "resampled_latents_equal_single_gpu": same,
                          "collectives": stats.get("collectives"), "form": stats.get("form"),
                          "communicator": getattr(cx, "name", None)}
    except Exception as e:          # noqa: BLE001
        out["config4"] = {"error": repr(e)[:300]}
    return out


def ShardedSweep(init, step, n, T, dist, key, ys, comm=None, **kw):
    """a prepared ShardedBootstrapSweep over the communicator `comm` ("peer" / "rccl" / "p2p" / "torch"; None: whatever
    GENMI_COMM / the default says)"""
    import torch
    from genjax_amd.inference.sharded import ShardedBootstrapSweep
    old = os.environ.get("GENMI_COMM")
    if comm is not None:
        os.environ["GENMI_COMM"] = comm
    try:
        return ShardedBootstrapSweep(init, step, n, T, dist, always_communicate=True, **kw).prepare(key, torch.from_numpy(ys))
    finally:
        if comm is not None:
            if old is None:
                os.environ.pop("GENMI_COMM", None)
            else:
                os.environ["GENMI_COMM"] = old


def pick_sharded_sweep(args, dist, world, rank, on_gpu, be, make):
    """The sharded sweep the timed region runs, and the start-up A/B that chose its communicator (outside the timed
    region, agreed across ranks, recorded in `config`).

    Candidates on a GPU box (GENMI_COMM set: that one alone): the FUSED PEER EXCHANGE (comm.PeerComm: no collective
    launch per step; every wait bounded by the device's wall clock, so a candidate that does not work costs seconds,
    not the job) as one hipGraph, and the direct RCCL communicator — as one hipGraph at world size 1, EAGER across GPUs:
    captured RCCL calls have never run at world > 1 in the build loop and a replay that hangs cannot be bounded
    (`--rccl-graph` opts in).  Each candidate runs one warm-up sweep and three timed ones; it is dropped if it
    raises, times out, or its evidence differs from the first working candidate's (every communicator must produce
    the SAME bits).  The fastest survivor (MAX over ranks of its time) is kept, the others are closed."""
    import torch
    from genjax_amd.inference.comm import _Deadline
    forced = os.environ.get("GENMI_COMM")
    graph_ok = on_gpu and not args.no_graph
    rccl_graph = graph_ok and (world == 1 or args.rccl_graph)
    if not on_gpu:
        cands = [(forced, False)]
    elif forced:
        cands = [(forced, graph_ok and (forced != "rccl" or rccl_graph) and forced != "torch")]
    else:
        cands = [("rccl", rccl_graph), ("peer", graph_ok)]
    timeout = float(os.environ.get("GENMI_COMM_TIMEOUT", "120"))

    def agree(ok, dt):
        if dist is None or world == 1:
            return ok, dt
        t = torch.tensor([1.0 if ok else 0.0, -dt], device=be.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t[0].item() > 0.5), -float(t[1].item())

    results, best, ref_log_ml = [], None, None
    for comm, graph in cands:
        sw, ok, dt, log_ml, err = None, True, float("inf"), None, None
        try:
            with _Deadline(timeout, f"the start-up sweeps over the {comm} communicator"):
                sw = make(comm=comm)
                if graph and (sw.cx is None or sw.cx.graph_safe):
                    sw.capture()
                sw.launch(); sw.finish()
                if on_gpu:
                    torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    sw.launch(); sw.finish()
                if on_gpu:
                    torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                log_ml = sw.log_ml()
        except Exception as e:          # noqa: BLE001
            ok, err = False, repr(e)[:200]
            print(f"bench.py: rank {rank}: communicator {comm!r} dropped: {e!r}", file=sys.stderr, flush=True)
        if ok and ref_log_ml is not None and log_ml != ref_log_ml:
            ok, err = False, f"log_ml {log_ml!r} differs from the first communicator's {ref_log_ml!r}"
        ok, dt = agree(ok, dt)
        if ok and ref_log_ml is None:
            ref_log_ml = log_ml
        results.append({"communicator": comm, "graph": bool(sw is not None and sw.graph is not None), "ok": ok,
                        "us_per_step": (1e6 * dt / sw.T) if ok else None, **({"error": err} if err else {})})
        if ok and (best is None or dt < best[1]):
            if best is not None:
                best[0].close()
            best = (sw, dt)
        elif sw is not None:
            try:
                sw.close()
            except Exception:       # noqa: BLE001
                pass
    if best is None:
        # nothing worked: the torch.distributed communicator, eager (the last resort; it fails loudly if it cannot run)
        sw = make(comm="torch")
        results.append({"communicator": "torch", "graph": False, "ok": True, "us_per_step": None, "fallback": True})
        return sw, results
    return best[0], results


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
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs 3 / 4 / 5 block after the headline")
    ap.add_argument("--no-roofline", action="store_true",
                    help="only the timed sweeps (rocprofv3's kernel-trace pass: its per-kernel averages are then IN-SWEEP "
                         "durations, not mixed with the isolated timing loops of the roofline block)")
    ap.add_argument("--sharded", action="store_true", help="use the multi-GPU code path even at world size 1")
    ap.add_argument("--rccl-graph", action="store_true",
                    help="N > 1: capture the RCCL collectives into the sweep's hipGraph too (default: eager across GPUs)")
    ap.add_argument("--two-launches", action="store_true",
                    help="single GPU: the two-launch step [site program -> resampler] instead of the default ONE launch "
                         "per step (the site program resamples the previous step first: BootstrapSweep(fuse_resample=...))")
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
        sw = BootstrapSweep(init, step, n, T, fuse_resample=False if args.two_launches else None).prepare(G.key(seed), torch.from_numpy(ys))
        if not args.no_graph:
            sw.capture()
        launch = sw.launch
    else:
        sw, comm_ab = pick_sharded_sweep(args, dist, world, rank, on_gpu, be,
                                         lambda **kw: ShardedSweep(init, step, n, T, dist, G.key(seed), ys, **kw))

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
                   "path": ("BootstrapSweep (hipGraph" + (", one launch per step" if getattr(sw, "fuse", False) else "")
                            + (", noise ahead on a second stream)" if getattr(sw, "noise_ahead", False) else ")")) if single else
                   ("ShardedBootstrapSweep (" + str(getattr(getattr(sw, "cx", None), "name", "no communicator"))
                    + (", noise ahead on a second stream" if getattr(sw, "noise_ahead", False) else "") + ")"),
                   "key": seed},
        "log_ml": log_ml, "log_ml_kalman": kal, "log_ml_abs_err": abs(log_ml - kal),
        "log_ml_rel_err": abs(log_ml - kal) / abs(kal),
    }

    if not single:
        out["config"]["communicator"] = sw.cx.name
        out["config"]["communicator_ab"] = comm_ab       # the start-up A/B (3 sweeps each, outside the timed region)
        out["config"]["GENMI_COMM"] = os.environ.get("GENMI_COMM", "(unset: rccl on a GPU box, torch.distributed otherwise)")
        out["config"]["graph"] = sw.graph is not None
        out["config"]["capacity_per_peer"] = sw.capacity
        out["config"]["full_capacity_reruns"] = sw.reruns

    if not single and world > 1 and dist is not None and not args.no_other_configs:
        # the OTHER scaling beside the headline's: a strong-scaling line also carries the weak
        # record (--particles PER GPU), a --weak line the strong one — same communicator, three sweeps, MAX over ranks
        from genjax_amd.inference.sharded import ShardedBootstrapSweep
        want_weak = not args.weak
        n2 = ((requested + 1023) // 1024) * 1024 if want_weak else \
            ((requested + world * 1024 - 1) // (world * 1024)) * 1024
        rec = {"scaling": "weak" if want_weak else "strong", "particles_per_gpu": n2, "particles_total": n2 * world}
        try:
            sw2 = ShardedBootstrapSweep(init, step, n2, T, dist, always_communicate=True, comm=sw.cx).prepare(
                G.key(seed), torch.from_numpy(ys))
            if sw.graph is not None:
                sw2.capture()
            sw2.launch(); sw2.finish()
            barrier()
            t2 = time.perf_counter()
            for _ in range(3):
                sw2.launch(); sw2.finish()
            barrier()
            dt2 = (time.perf_counter() - t2) / 3
            tt = torch.tensor([dt2], device=be.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt.item())
            rec.update(value=n2 * world * T / dt2, unit="particle-steps/s", us_per_step=1e6 * dt2 / T, log_ml=sw2.log_ml(),
                       one_launch_per_step=bool(getattr(sw2, "fuse_sh", False)))
            if sw2.graph is not None:
                be.c.gmx_graph_destroy(sw2.graph)
                sw2.graph = None
        except Exception as e:          # noqa: BLE001  (reported, never fails the headline)
            rec["error"] = repr(e)[:300]
        out["weak" if want_weak else "strong"] = rec
    if not single:
        out["config"]["one_launch_per_step"] = bool(getattr(sw, "fuse_sh", False))
    if not single and on_gpu and not args.no_other_configs and dist is not None and n * world >= N_PARTICLES and T == T_STEPS:
        oc = other_configs_sharded(dist, world, rank, sw.cx, be)       # COLLECTIVE: every rank
        if rank == 0:
            out["other_configs"] = oc
    if rank == 0 and on_gpu:
        out["roofline"] = ({"code_identity": code_identity(sw), "note": "--no-roofline"} if args.no_roofline
                           else measure_roofline(be, sw, n, T, world, single, value))
        if single and not args.no_other_configs and n == N_PARTICLES and T == T_STEPS:
            try:
                out["other_configs"] = other_configs()
            except Exception as e:
                out["other_configs"] = {"error": repr(e)[:300]}
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
