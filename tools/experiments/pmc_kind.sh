# SQ instruction counters of every kernel of the config-2 sweep under one resampling kind (a separate rocprofv3 --pmc pass):
#   tools/experiments/pmc_kind.sh <kind> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
kind=$1; tag=$2
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
printf 'pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES\n' > /tmp/pmc_kind.txt
RESAMPLE=$kind REPS=1 ROUNDS=1 T=10 rocprofv3 -i /tmp/pmc_kind.txt --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/bench_ab.py noise_ahead True > /dev/null 2> $R/gpurun_out/$tag.err
python3 - <<PY > $R/gpurun_out/$tag.txt
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); calls[k] += 1
print("kind $kind: per WAVE averages over all launches (T = 10 sweep)")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = c.get("SQ_WAVES", 0) or 1
    print(f"{k:44s} launches={calls[k]:4d} waves={int(w):8d} valu={c.get('SQ_INSTS_VALU',0)/w:8.1f} salu={c.get('SQ_INSTS_SALU',0)/w:7.1f} vmem_rd={c.get('SQ_INSTS_VMEM_RD',0)/w:6.1f} vmem_wr={c.get('SQ_INSTS_VMEM_WR',0)/w:6.1f} lds={c.get('SQ_INSTS_LDS',0)/w:6.1f} wait_any/wave_cycles={c.get('SQ_WAIT_ANY',0)/(c.get('SQ_WAVE_CYCLES',0) or 1):.2f}")
PY
cat $R/gpurun_out/$tag.txt
