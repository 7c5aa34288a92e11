#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# kernel trace + stats, then the SQ PMC passes, each in its own rocprofv3 run.
set -e
: ${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
# (hiprtc inside a profiled process compiles the same source to different code, DESIGN section 5: compile by a plain run first)
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-roofline --T 10 "$@" > /dev/null 2> $out.plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline "$@" > $out.bench.json 2> $out.trace.err
rocprofv3 -i $GRAFT_REPO_ROOT/profiles/pmc/sq_pass.txt --kernel-trace --output-format csv -d $out/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-roofline --T 10 "$@" > /dev/null 2> $out.pmc.err
cd $GRAFT_REPO_ROOT && python3 tools/prof_summary.py $out
