# the sorted multinomial's background kernels under different residency caps: tools/experiments/sorted_pad.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
for pad in 0 8000 16000 32000 56000; do
  echo "pad $pad"; GENMI_SORTED_LDS_PAD=$pad KINDS=multinomial_sorted python3 $R/tools/bench_kinds.py 2>/dev/null
done > $R/gpurun_out/${tag}_pads.txt
cat $R/gpurun_out/${tag}_pads.txt
