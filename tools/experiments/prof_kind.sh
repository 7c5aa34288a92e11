# per-kernel times of the config-2 sweep under one resampling kind: tools/experiments/prof_kind.sh <kind> <tag>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
kind=$1; tag=$2
mkdir -p $R/gpurun_out
# (hiprtc inside a profiled process compiles the same source to different code, DESIGN section 5: compile by a plain run first)
RESAMPLE=$kind REPS=1 ROUNDS=1 python3 $R/tools/bench_ab.py noise_ahead True > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
RESAMPLE=$kind REPS=3 ROUNDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/bench_ab.py noise_ahead True > $R/gpurun_out/$tag.json 2> $R/gpurun_out/$tag.err
cat $R/gpurun_out/$tag.json
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
