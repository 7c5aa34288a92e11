#!/bin/bash
# usage (GPU box, repo root): tools/experiments/trace_sharded.sh <tag> [comm]
# rocprofv3 kernel trace (+ --stats) of `bench.py --sharded` at world size 1 over one communicator (default: peer),
# a timeline of consecutive kernels of one step, and the per-kernel averages.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; comm=${2:-peer}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
GENMI_COMM=$comm rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace_$comm -- python3 $R/bench.py --sharded --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline > $R/gpurun_out/${tag}_trace_$comm.json 2> $R/gpurun_out/${tag}_trace_$comm.err
python3 $R/tools/trace_timeline.py $R/gpurun_out/${tag}_trace_$comm --skip 150 --count 24 > $R/gpurun_out/${tag}_timeline_$comm.txt
f=$(find $R/gpurun_out/${tag}_trace_$comm -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats_$comm.csv
cat $R/gpurun_out/${tag}_timeline_$comm.txt
