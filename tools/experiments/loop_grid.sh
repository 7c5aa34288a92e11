#!/bin/bash
# the one-launch step's LOOPED kernel at BASELINE config 2's own size (1e6 particles), by workgroup count:
#   tools/experiments/loop_grid.sh > gpurun_out/r06d_loop_grid.txt
for cfg in "0 0" "1 1024" "1 768" "1 512" "1 384" "1 256"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset GENMI_RS_LOOP GENMI_RS_LOOP_GRID; else export GENMI_RS_LOOP=1 GENMI_RS_LOOP_GRID=$2; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('loop=$1 grid=$2', 'value %.4e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step'] / 100), 'log_ml', d.get('log_ml'))"
done
