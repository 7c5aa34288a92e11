"""tests/fuzz_models.py on the MI355X beyond the seeds the test suite runs: random `@gen` models built with the
product (through libgenmi_hip.so) and with the oracle, every GFI method / edit / MH move / ImportanceK / resampling
compared bit for bit.  Prints one JSON line of counts.
  python tools/experiments/fuzz_on_device.py [n_interpreter_seeds] [n_jit_seeds] [n_jit_smc_seeds] [seed_offset]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import _lib
from tests import fuzz_models as F

_lib.get()                                   # the HIP library, or a loud failure
if os.environ.get("FUZZ_FORCE_JIT") == "1":
    # EVERY program through hiprtc, at the interpreter seeds' small batch: what a specialised kernel computes against the
    # oracle for many programs per minute (at 2^18 particles the numpy oracle is the clock: a plate of plates of long
    # vector sites is minutes per model)
    from genjax_amd import engine
    engine.JIT_MIN_PARTICLES = 1
n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_jit = int(sys.argv[2]) if len(sys.argv) > 2 else 24
off = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # fresh seeds: every range below shifted by it
budget = float(os.environ.get("FUZZ_SECONDS", "1e9"))       # no NEW seed is started after this many seconds
out = dict(seed_offset=off, models=0, over_the_limits=0, failures=[], smc_models=0, big_plate_models=0, jit_models=0)
t0 = time.time()
for seed in range(10_000 + off, 10_000 + off + n_small):
    if time.time() - t0 > budget:
        out['stopped_at_budget'] = True
        break
    try:
        F.run_one(seed)
        out["models"] += 1
    except F.OverTheLimits:
        out["over_the_limits"] += 1
    except Exception as e:      # noqa: BLE001
        out["failures"].append((seed, "run_one", repr(e)[:200])); print("FAIL", out["failures"][-1], file=sys.stderr, flush=True)
    if seed % 50 == 49:          # (a run that writes nothing for minutes is taken to be hung)
        print(f"# interpreter seed {seed} done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
for seed in range(20_000 + off, 20_000 + off + n_small // 2):
    if time.time() - t0 > budget:
        out['stopped_at_budget'] = True
        break
    try:
        F.run_smc_one(seed)
        out["smc_models"] += 1
    except F.OverTheLimits:
        out["over_the_limits"] += 1
    except Exception as e:      # noqa: BLE001
        out["failures"].append((seed, "run_smc_one", repr(e)[:200])); print("FAIL", out["failures"][-1], file=sys.stderr, flush=True)
    if seed % 50 == 49:
        print(f"# smc seed {seed} done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
for seed in range(30_000 + off, 30_000 + off + 16):
    if time.time() - t0 > budget:
        out['stopped_at_budget'] = True
        break
    try:
        F.run_big_one(seed)
        out["big_plate_models"] += 1
    except Exception as e:      # noqa: BLE001
        out["failures"].append((seed, "run_big_one", repr(e)[:200])); print("FAIL", out["failures"][-1], file=sys.stderr, flush=True)
    print(f"# big-plate seed {seed} done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
for seed in range(40_000 + off, 40_000 + off + n_jit):          # 2^18 particles: the hiprtc-specialised programs
    if time.time() - t0 > budget:
        out['stopped_at_budget'] = True
        break
    try:
        F.run_one(seed, B=1 << 18)
        out["jit_models"] += 1
    except F.OverTheLimits:
        out["over_the_limits"] += 1
    except Exception as e:      # noqa: BLE001
        out["failures"].append((seed, "run_one at 2^18", repr(e)[:200])); print("FAIL", out["failures"][-1], file=sys.stderr, flush=True)
    print(f"# jit seed {seed} done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
n_jit_smc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
out["jit_smc_models"] = 0
for seed in range(50_000 + off, 50_000 + off + n_jit_smc):      # ImportanceK with 2^18 particles, then a resampling of them
    if time.time() - t0 > budget:
        out['stopped_at_budget'] = True
        break
    try:
        F.run_smc_one(seed, K=1 << 18)
        out["jit_smc_models"] += 1
    except F.OverTheLimits:
        out["over_the_limits"] += 1
    except Exception as e:      # noqa: BLE001
        out["failures"].append((seed, "run_smc_one at 2^18", repr(e)[:200])); print("FAIL", out["failures"][-1], file=sys.stderr, flush=True)
    print(f"# jit smc seed {seed} done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))
