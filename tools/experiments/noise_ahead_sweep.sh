#!/bin/bash
# EXPERIMENT: tools/experiments/noise_ahead.py over residency caps (PAD), group sizes (BATCH), launch forms (GROUP),
# steps per noise launch (ROWS) and chain priority (PRIO); one JSON line per run
out=${1:-gpurun_out/noise_ahead_sweep.txt}
: > $out
run() {  # PAD BATCH GROUP ROWS PRIO [CHAIN_ONLY]
  PAD=$1 BATCH=$2 RING=$((2 * $2)) GROUP=$3 ROWS=$4 PRIO=$5 CHAIN_ONLY=${6:-0} timeout -k 10 200 python tools/experiments/noise_ahead.py 2>gpurun_out/noise_ahead_last.err | tail -1 >> $out || { echo "{\"failed\": \"$*\"}" >> $out; tail -3 gpurun_out/noise_ahead_last.err >> $out; }
}
run 56000 10 0 10 0 1
for prio in 0 3; do
  for pad in 56000 64000; do
    run $pad 10 2 1 $prio
    run $pad 10 2 2 $prio
    run $pad 10 2 10 $prio
  done
done
cat $out
