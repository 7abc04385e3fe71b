import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = pkg.workloads.make("darcy64")
eng = post.HipEngine(pkg, w)
job = post.ShardedPosterior(eng, dist=dist, rank=0, world=1, k_samples=16, group=4, replicate_factor=False)
job.replicate = False
job.prepare()
job.step(0)
def t(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("factor+share", t(job._factor_and_share), "ms")
print("mean", t(eng.mean), "ms")
mu = eng.mean()
print("sample16", t(lambda: eng.sample(16, mu, 1, 0)), "ms")
job.replicate = True
print("replicated factor", t(job._factor_and_share), "ms")
dist.destroy_process_group()
