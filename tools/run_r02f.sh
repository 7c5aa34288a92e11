cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02f_pytest_gpu.log 2>&1; tail -12 gpurun_out/r02f_pytest_gpu.log
timeout 600 python tools/bench_configs.py > gpurun_out/r02f_configs_3_4.json 2> gpurun_out/r02f_configs.err; cat gpurun_out/r02f_configs_3_4.json; tail -3 gpurun_out/r02f_configs.err
