cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/experiments/noise_ahead.py > gpurun_out/r02g_noise_ahead.json 2> gpurun_out/r02g_noise_ahead.err; cat gpurun_out/r02g_noise_ahead.json; tail -5 gpurun_out/r02g_noise_ahead.err
