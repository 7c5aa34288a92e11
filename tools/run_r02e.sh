cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02e_pytest_gpu.log 2>&1; tail -12 gpurun_out/r02e_pytest_gpu.log
