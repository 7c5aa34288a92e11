#!/bin/bash
# tools/bench_noise_ahead.py over the residency cap of the noise programs and the group size; one JSON line per run
out=${1:-gpurun_out/noise_ahead_tune.txt}
: > $out
for pad in 26000 32000 40000 50000 56000; do
  for grp in 5 10 25; do
    GENMI_NOISE_LDS_PAD=$pad GENMI_NOISE_GROUP=$grp REPS=10 timeout -k 10 120 python tools/bench_noise_ahead.py 2>/dev/null | tail -1 >> $out
  done
done
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print("pad", d["lds_pad"], "group", d["group"], "one-stream %.2f  noise-ahead %.2f us/step  same=%s" % (d["one_stream"]["us_per_step"], d["noise_ahead"]["us_per_step"], d["bit_identical"]))
PY
