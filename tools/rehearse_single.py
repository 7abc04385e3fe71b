"""Single process on cuda:0, started by tools/rehearse_driver.py: (1) the reference results of the
two-rank rehearsal (same sample ids drawn by one rank); (2) the library's own RCCL communicator
(gmrf_comm_*, the path a Julia host uses) with a world of ONE rank: unique id, create, layout record
broadcast, block-range factor broadcast beside the factorisation, wait, all-reduce.
Writes single.npz into argv[1]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g
from importlib import import_module

outdir = sys.argv[1]
pkg = g.load_package()
post = import_module(g.PKG_NAME + ".posterior")
torch.cuda.set_device(0)
w = pkg.workloads.make("darcy64")
vals = np.stack([w.Q.data, 1.5 * w.Q.data])
rhs = np.stack([w.rhs, -w.rhs])
out = {}
# (1) one rank, both ranks' sample ids: ids of rank r at step s start at (s * 2 + r) * k * batch
F = pkg.TridiagonalCholeskyFactor(batch=2).factor(w.Q, w.n_blocks, values=vals)
mu = F.solve_batch(torch.from_numpy(rhs[:, None, :]).cuda())[:, 0, :]
out["mu"] = mu.cpu().numpy()
for step in range(2):
    for r in range(2):
        out[f"X{step}_{r}"] = F.sample_batch(6, mean=mu, seed=42, first_id=(step * 2 + r) * 6 * 2, like=mu).cpu().numpy()
# (2) C-ABI communicator, world of one rank, transport "cabi" through the same driver
uid = pkg.api.Comm.unique_id()
comm = pkg.api.Comm(0, 0, 1, uid)
lay = np.arange(5, dtype=np.int64)
comm.bcast_host(lay, 0)
assert list(lay) == [0, 1, 2, 3, 4]
eng = post.HipEngine(pkg, w, device_index=0, batch=2, values=vals, rhs=rhs, keep_l=False, transport="cabi", comm=comm)
job = post.ShardedPosterior(eng, rank=0, world=1, k_samples=6, seed=42, group=5, force_shared=True)
job.prepare()
for step in range(2):
    mu_c, X_c = job.step(step)
torch.cuda.synchronize()
out["mu_cabi"] = mu_c.cpu().numpy()
out["X_cabi"] = X_c.cpu().numpy()           # ids (1 * 1 + 0) * 6 * 2 = 12 ..: step 0 of rank 1 above... see driver
acc = torch.ones(1000, dtype=torch.float64, device="cuda")
comm.allreduce_sum(acc, eng.F)
torch.cuda.synchronize()
out["acc_cabi"] = acc.cpu().numpy()
comm.close()
np.savez(os.path.join(outdir, "single.npz"), **out)
print("rehearse_single ok", flush=True)
