#!/bin/bash
# config 3 (BootstrapSweep(rejuvenate=...)): which draws of the chained MH + extension program go to the background
# stream (GENMI_NOISE_ROOTS) x the noise programs' residency cap; tools/bench_sweep3.py prints both forms per run
out=${1:-gpurun_out/config3_noise_roots.txt}
: > $out
for r in LDKEY KSPLITU all; do
  for pad in 1 32000 56000 81920; do
    GENMI_NOISE_ROOTS=$r GENMI_NOISE_LDS_PAD=$pad timeout -k 10 200 python tools/bench_sweep3.py 2>/dev/null | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('roots $r lds_pad $pad: one stream %.2f, noise ahead %.2f us/step, bit_identical %s' % (d['one_stream']['us_per_step'], d['noise_ahead']['us_per_step'], d['bit_identical']))" >> $out
  done
done
cat $out
