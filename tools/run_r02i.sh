cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02i_pytest_gpu.log 2>&1; tail -2 gpurun_out/r02i_pytest_gpu.log
for v in "" "GENMI_TILE_Q=1"; do
env $v timeout 300 python bench.py --no-cpu-baseline > "gpurun_out/r02i_bench_$v.json" 2> gpurun_out/r02i_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r02i_bench_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
done
