"""Runs bench.py's control flow on the tests' CPU mirror of the C-ABI (tests/hostsim) under gloo — used by
tests/test_distributed_cpu.py::test_bench_contract_control_flow only.  Installs the mirror, then executes
bench.py unchanged with the same argv."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.hostsim as hs  # noqa: E402

hs.install()
# a bare `--gpus N` makes bench.py spawn its own ranks: they must come back through THIS file (the CPU mirror)
os.environ["GENMI_BENCH_ENTRY"] = os.path.abspath(__file__)
sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
