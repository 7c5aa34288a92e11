#!/bin/bash
# Everything DESIGN.md quotes, in one go, on an MI355X box (from the repo root; ~30 min — past gpurun's 20-minute
# limit per call: run the blocks below in two or three calls, as round 5 did: r05s = the counter passes, r05t = suite + bench):
#   tools/reproduce.sh <tag>      -> gpurun_out/<tag>_*
# Each rocprofv3 pass is its own invocation (kernel trace + stats, SQ counters, TCC counters: never combined);
# every step runs under `timeout` so that a wedged process cannot hold the box.
tag=${1:-run}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$root
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1; tail -1 $out/${tag}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; tail -1 $out/${tag}_smoke.log | cut -c1-80
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 300 python bench.py --sharded --no-cpu-baseline > $out/${tag}_bench_sharded_world1.json 2> /dev/null
timeout 300 python tools/bench_noise_ahead.py 2> /dev/null | tail -1 > $out/${tag}_noise_ahead_ab.json
timeout 300 python tools/experiments/skewed_resample.py 2> /dev/null | tail -1 > $out/${tag}_skewed_resample.json
timeout 600 bash tools/prof.sh $tag > $out/${tag}_pmc_summary.txt 2>&1
timeout 600 bash tools/traffic.sh $tag > $out/${tag}_traffic.txt 2>&1
for k in 3 4 5; do timeout 400 bash tools/prof_config.sh $tag $k > $out/${tag}_config${k}_prof.log 2>&1; done   # -> <tag>_config<k>_{pmc_summary.txt,kernel_stats.csv}
python3 tools/make_counters.py $tag > $out/${tag}_counters.log 2>&1     # -> gpurun_out/<tag>_counters.json (copy to profiles/counters.json)
timeout 300 python tools/bench_configs.py > $out/${tag}_configs_3_4.json 2> /dev/null
timeout 300 python tools/bench_sweep3.py 2> /dev/null | grep "^{" > $out/${tag}_config3_native_sweep.json
timeout 300 python tools/bench_mixture.py 2> /dev/null | tail -1 > $out/${tag}_mixture_config5.json
timeout 300 python tools/bench_hmc.py 2> /dev/null | tail -1 > $out/${tag}_hmc.json
timeout 300 python tools/bench_kinds.py 2> /dev/null | tail -1 > $out/${tag}_kinds.json
# round 5: the one-launch sharded step against two launches; a long vector-valued site against its vmap-plate spelling;
# O(1) IndexRequest on a long scan; the bound on what a destination-centric resampling prologue could save
timeout 300 python tools/bench_sharded_fuse_ab.py 2> /dev/null | tail -1 > $out/${tag}_sharded_fuse_ab.json
timeout 300 bash tools/experiments/prof_sharded_one_launch.sh > $out/${tag}_sh_prof.log 2>&1      # SQ counters of the one-launch sharded kernel -> r05m_sh_pmc_summary.txt
# hiprtc inside a profiled process compiles the same source to different code: config 5, JIT cache off
GENMI_JIT_CACHE=0 timeout 120 python tools/experiments/config5_timing.py 4 > $out/${tag}_c5_plain.json 2> /dev/null
(cd /tmp && GENMI_JIT_CACHE=0 TMPDIR=/tmp timeout 120 rocprofv3 --kernel-trace --output-format csv -d $out/prof_${tag}_c5 -- python3 $root/tools/experiments/config5_timing.py 4 > $out/${tag}_c5_under_rocprofv3.json 2> /dev/null)
timeout 300 python tools/vector_site_cost.py 100000 2> /dev/null | tail -1 > $out/${tag}_vector_site_cost.json
timeout 300 python tools/scan_index_request_cost.py 100000 4096 2> /dev/null | tail -1 > $out/${tag}_scan_index_request_cost.json
python tools/experiments/build_diag_lib.py nopoll > /dev/null 2>&1 && timeout 300 python tools/experiments/poll_hop_bound.py 2> /dev/null | tail -1 > $out/${tag}_poll_hop_bound.json
# the ordered resamplers beside each other: the chain without the background stream, the kernel alone, the sorted kind's kernels
for k in systematic stratified multinomial_sorted; do RESAMPLE=$k timeout 120 python tools/experiments/chain_only.py 2> /dev/null | tail -1; done > $out/${tag}_chain_only.txt
timeout 120 python tools/experiments/sorted_micro.py 2> /dev/null | tail -1 > $out/${tag}_sorted_micro.json
timeout 300 bash tools/experiments/prof_kind.sh multinomial_sorted ${tag}_kind_multinomial_sorted > $out/${tag}_kind_multinomial_sorted.txt 2>&1
python - <<PY
import json
b = json.load(open("$out/${tag}_bench.json"))
print("config 2:", "%.3e" % b["value"], b["unit"], "| %.1f us/step" % (1e3 * b["ms_per_step"] / 100),
      "| roofline frac %.3f" % b["roofline"]["frac"], "| cpu_baseline %.2e on %d cores" % (b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
PY
