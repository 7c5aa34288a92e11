#!/bin/bash
# usage (on the GPU box, from the repo root): tools/traffic.sh <tag>
# FETCH_SIZE / WRITE_SIZE passes (profiles/pmc/tcc_pass.txt, counters only) over the bench sweep and over the
# calibration copy kernels (tools/calib, built here with hipcc), then tools/traffic.py -> gpurun_out/traffic_<tag>.json
set -e
: ${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/traffic_$tag
[ -x $root/tools/calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $root/tools/calib $root/tools/calib.hip
cd /tmp && export TMPDIR=/tmp
export GENMI_NOISE_GROUP=1    # one noise launch per step, so that "bytes per launch" is per step for every kernel
# (hiprtc inside a profiled process compiles the same source to different code, DESIGN section 5: compile by a plain run first)
python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-roofline --T 10 > /dev/null 2> $out.plain.err
rocprofv3 -i $root/profiles/pmc/tcc_pass.txt --kernel-trace --output-format csv -d $out/bench -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-roofline --T 10 > /dev/null 2> $out.bench.err
rocprofv3 -i $root/profiles/pmc/tcc_pass.txt --kernel-trace --output-format csv -d $out/calib -- $root/tools/calib > /dev/null 2> $out.calib.err
cd $root && python3 tools/traffic.py $out/bench $out/calib $root/gpurun_out/traffic_$tag.json
