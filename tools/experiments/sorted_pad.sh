# the sorted multinomial's sweep under different residency caps of the background kernels (normals' programs x table kernels): <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
for npad in 40000 56000 80000; do for spad in 0 16000 40000; do
  echo "noise_pad $npad sorted_pad $spad"; GENMI_NOISE_LDS_PAD=$npad GENMI_SORTED_LDS_PAD=$spad KINDS=multinomial_sorted python3 $R/tools/bench_kinds.py 2>/dev/null
done; done > $R/gpurun_out/${tag}_pads.txt
for grp in 5 10 20; do echo "noise_group $grp"; GENMI_NOISE_GROUP=$grp KINDS=multinomial_sorted python3 $R/tools/bench_kinds.py 2>/dev/null; done >> $R/gpurun_out/${tag}_pads.txt
cat $R/gpurun_out/${tag}_pads.txt
