cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02h_pytest_gpu.log 2>&1; tail -5 gpurun_out/r02h_pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02h_bench.json 2> gpurun_out/r02h_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r02h_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
timeout 600 bash tools/prof.sh r02h > gpurun_out/r02h_pmc_summary.txt 2>&1; head -24 gpurun_out/r02h_pmc_summary.txt
