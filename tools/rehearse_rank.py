"""One rank of the two-rank rehearsal of the shared-factor job on ONE GPU (both ranks on cuda:0,
gloo moves the CUDA tensors through the host): rank 0 factors block ranges, every finished range of
Linv / C blocks is broadcast, both ranks take the mean and draw their own sample ids.  Started by
tools/rehearse_driver.py through torch.distributed.run; writes r<rank>.npz into argv[1]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import __graft_entry__ as g

outdir = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
from importlib import import_module
pkg = g.load_package()
post = import_module(g.PKG_NAME + ".posterior")
torch.cuda.set_device(0)
w = pkg.workloads.make("darcy64")
batch = 2
vals = np.stack([w.Q.data, 1.5 * w.Q.data])
rhs = np.stack([w.rhs, -w.rhs])
eng = post.HipEngine(pkg, w, device_index=0, batch=batch, values=vals, rhs=rhs, keep_l=(rank == 0), transport="torch")
job = post.ShardedPosterior(eng, dist=dist, rank=rank, world=world, k_samples=6, seed=42, group=5)
job.prepare()
out = {}
for step in range(2):                       # the second step re-uses buffers the first step's sweeps read
    mu, X = job.step(step)
    torch.cuda.synchronize()
    out[f"mu{step}"] = mu.cpu().numpy()
    out[f"X{step}"] = X.cpu().numpy()
acc = ((X - mu[:, None, :]) ** 2).sum(dim=1)          # variance accumulators: one all-reduce (R2)
dist.all_reduce(acc)
out["acc"] = acc.cpu().numpy()
out["solves"] = job.solves_per_step()
lds = []
for p in range(batch):                      # rank 1 adopted the factor without its L blocks: the log-determinant parts
    eng.F.select_problem(p)                 # travelled inside the packed transport image
    lds.append(eng.F.logdet())
out["logdet"] = np.array(lds)
out["bytes"] = np.array([eng.transport_bytes()])
out["layout"] = eng.F.get_layout()
# Round 4: the all-gather form of the same batch -- every rank factors ONE of the two problems (HipGatherEngine), block ranges are
# all-gathered (gloo moves the packed images through the host), both ranks take both means and draw their own sample ids
eng.F.close()
geng = post.HipGatherEngine(pkg, w, device_index=0, batch_total=batch, world=world, rank=rank, values_all=vals, rhs_all=rhs,
                            keep_l=False, transport="torch")
gjob = post.ShardedPosterior(geng, dist=dist, rank=rank, world=world, k_samples=6, seed=42, group=5, share="allgather")
gjob.prepare()
for step in range(2):
    gmu, gX = gjob.step(step)
    torch.cuda.synchronize()
    out[f"gmu{step}"] = gmu.cpu().numpy()
    out[f"gX{step}"] = gX.cpu().numpy()
glds = []
for p in range(batch):
    geng.F_all.select_problem(p)
    glds.append(geng.F_all.logdet())
out["glogdet"] = np.array(glds)
out["gbytes"] = np.array([geng.transport_bytes()])
out["gsolves"] = gjob.solves_per_step()
np.savez(os.path.join(outdir, f"r{rank}.npz"), **out)
dist.barrier()
dist.destroy_process_group()
