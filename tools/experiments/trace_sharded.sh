set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
GENMI_SHARDED_GRAPH=$mode rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03j_trace_g$mode -- python3 $R/bench.py --sharded --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline > $R/gpurun_out/r03j_trace_g$mode.json 2> $R/gpurun_out/r03j_trace_g$mode.err
python3 $R/tools/trace_timeline.py $R/gpurun_out/r03j_trace_g$mode --skip 150 --count 30 > $R/gpurun_out/r03j_timeline_g$mode.txt
done
cat $R/gpurun_out/r03j_timeline_g1.txt
