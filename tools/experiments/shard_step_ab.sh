# A/B of the fused sharded routing launch at world size 1 (RCCL issued): LDS-routed form vs the per-thread runs
set -e
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
for f in 1 0 1 0; do
  GENMI_SHARD_FILL=$f python3 $R/bench.py --sharded --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GENMI_SHARD_FILL=$f', round(d['ms_per_step']*10,2), 'us/step', d['log_ml'])"
done
