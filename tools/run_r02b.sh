cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02b_pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r02b_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
GENMI_TILE_Q=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02b_bench_noq.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r02b_bench_noq.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
timeout 600 bash tools/prof.sh r02b > gpurun_out/r02b_pmc_summary.txt 2>&1; cat gpurun_out/r02b_pmc_summary.txt
