set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02a_pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; cat gpurun_out/r02a_bench.json
GENMI_JIT_PREFETCH=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02a_bench_nopre.json 2>/dev/null; cat gpurun_out/r02a_bench_nopre.json
