set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_r05m_sh
cd /tmp && export TMPDIR=/tmp
export GENMI_COMM=peer ONLY=one_launch REPS=1
# (hiprtc inside a profiled process compiles the same source to different code, DESIGN section 5: compile by a plain run first)
python3 $R/tools/bench_sharded_fuse_ab.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/bench_sharded_fuse_ab.py > $R/gpurun_out/r05m_sh_trace.json 2> $out.trace.err
rocprofv3 -i $R/profiles/pmc/sq_pass.txt --kernel-trace --output-format csv -d $out/pmc -- python3 $R/tools/bench_sharded_fuse_ab.py > /dev/null 2> $out.pmc.err
cd $R && python3 tools/prof_summary.py $out > gpurun_out/r05m_sh_pmc_summary.txt
cat gpurun_out/r05m_sh_pmc_summary.txt
