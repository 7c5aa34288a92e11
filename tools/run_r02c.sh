cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02c_pytest_gpu.log
for v in "" "GENMI_TILE_Q=0"; do
env $v timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02c_bench_$v.json 2> gpurun_out/r02c_bench.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02c_bench_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
done
