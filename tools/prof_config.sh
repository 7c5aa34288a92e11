#!/bin/bash
# usage (GPU box, repo root): tools/prof_config.sh <tag> <3|4|5>
# kernel trace + stats, then the SQ PMC passes (each its own rocprofv3 run, counters only) of ONE of the other BASELINE
# configs (tools/run_config.py) -> gpurun_out/prof_<tag>_config<k>/{trace,pmc}, <tag>_config<k>_{units,pmc_summary}
set -e
: ${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT
tag=$1; k=$2
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_${tag}_config$k
mkdir -p $R/gpurun_out
# hiprtc inside a profiled process compiles the same source to different code (engine.program_digest): the kernels are
# compiled by a PLAIN run first; the profiled runs below then load them from the JIT cache
python3 $R/tools/run_config.py $k 1 > $R/gpurun_out/${tag}_config${k}_units_plain.json 2> $out.plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/run_config.py $k 3 > $R/gpurun_out/${tag}_config${k}_units_trace.json 2> $out.trace.err
rocprofv3 -i $R/profiles/pmc/sq_pass.txt --kernel-trace --output-format csv -d $out/pmc -- python3 $R/tools/run_config.py $k 1 > $R/gpurun_out/${tag}_config${k}_units_pmc.json 2> $out.pmc.err
cd $R && python3 tools/prof_summary.py $out > $R/gpurun_out/${tag}_config${k}_pmc_summary.txt
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_config${k}_kernel_stats.csv
tail -3 $R/gpurun_out/${tag}_config${k}_pmc_summary.txt
