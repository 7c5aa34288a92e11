"""Host-side profile of one ShardedBootstrapSweep sweep at world size 1 (RCCL calls issued)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import genjax_amd as G
from genjax_amd import workloads
from genjax_amd.inference.sharded import ShardedBootstrapSweep
T = 100
ys = workloads.lgssm_data(T)
init, step = workloads.make_lgssm(G)
sw = ShardedBootstrapSweep(init, step, 1_000_000, T, dist, always_communicate=True).prepare(G.key(1), torch.from_numpy(ys))
print("communicator:", sw.cx.name)
for _ in range(3):
    sw.launch(); sw.finish()
torch.cuda.synchronize()
t0 = time.perf_counter(); sw.launch(); t1 = time.perf_counter(); sw.finish(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue host time {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms")
pr = cProfile.Profile(); pr.enable(); sw.launch(); pr.disable(); sw.finish()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
if "--graph" in sys.argv:
    ref = sw.state().clone(); ref_ml = sw.log_ml()
    sw.capture()
    for _ in range(3):
        sw.launch(); sw.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sw.launch(); sw.finish()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"graph: {1e3*(t1-t0)/10:.3f} ms per sweep; same state {bool(torch.equal(ref, sw.state()))}, same log_ml {ref_ml == sw.log_ml()}")
dist.destroy_process_group()
