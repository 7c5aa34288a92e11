"""AddressSanitizer + UndefinedBehaviorSanitizer over the C / C++ that runs on the CPU side (SURVEY.md §5 "race
detection / sanitizers"): oracle/orc_core.c, oracle/orc_sweep.c and tests/hostsim/hostsim.cpp — which compiles the
PRODUCT's device headers (gmx_vm.h, gmx_dist.h, gmx_rng.h, gmx_math.h) for the host, so the interpreter, samplers
and log-densities themselves are checked.  CPU only (GPU sanitizers are not available on this pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_side_c_code_is_clean_under_asan_and_ubsan():
    import tests.hostsim as hs
    hs_so = hs.build_sanitized()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ,
               LD_PRELOAD=asan,                       # the runtime must come first in a non-instrumented python
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               PYTHONMALLOC="malloc", OMP_NUM_THREADS="2",
               GENMI_HOSTSIM_SO=hs_so, GENMI_ORACLE_SO=os.path.join(ROOT, "oracle", "_build", "liborc_san.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_driver.py")], env=env,
                       capture_output=True, text=True, timeout=1200)
    bad = [ln for ln in r.stderr.splitlines() if "AddressSanitizer" in ln or "runtime error:" in ln]
    assert r.returncode == 0 and not bad, (r.returncode, bad[:5], r.stderr[-3000:])
    assert "sanitize_driver ok" in r.stdout
