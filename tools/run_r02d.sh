cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02d_pytest_gpu.log 2>&1; tail -15 gpurun_out/r02d_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; tail -c 3000 gpurun_out/r02d_bench.json
timeout 600 bash tools/prof.sh r02d > gpurun_out/r02d_pmc_summary.txt 2>&1; head -30 gpurun_out/r02d_pmc_summary.txt
