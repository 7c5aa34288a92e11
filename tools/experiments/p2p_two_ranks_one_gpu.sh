# two ranks of the sharded sweep on ONE GPU over the peer-mapped communicator (tests/dist_worker.py with GENMI_TEST_OPTS on_gpu)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
GENMI_COMM=${COMM:-p2p} GENMI_TEST_OPTS="{\"on_gpu\": 1, \"noise_ahead\": ${NA:-0}, \"capture\": ${CAP:-0}}" GENMI_NOISE_GROUP=3 GENMI_COMM_TIMEOUT=60 \
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 \
  tests/dist_worker.py $R/gpurun_out/p2p2 4096 6 > $R/gpurun_out/p2p2.log 2>&1
echo "rc=$?"
grep -v "^\[Gloo\]\|^$\|amdgpu.ids" $R/gpurun_out/p2p2.log | tail -40
