cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" "GENMI_JIT_DEFS=-DGMX_DIAG_SHORT_THREEFRY=1"; do
env $v timeout 300 python bench.py --no-cpu-baseline > "gpurun_out/r02j_bench_$v.json" 2> gpurun_out/r02j_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r02j_bench_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['kernel_us'])"
done
