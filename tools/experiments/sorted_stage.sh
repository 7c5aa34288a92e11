# the sorted multinomial resampler reading its table directly (0) or through LDS windows (1): <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
for st in 0 1 0 1; do
  echo "stage $st"; GENMI_SORTED_STAGE=$st KINDS=multinomial_sorted python3 $R/tools/bench_kinds.py 2>/dev/null
done > $R/gpurun_out/${tag}_stage.txt
cat $R/gpurun_out/${tag}_stage.txt
