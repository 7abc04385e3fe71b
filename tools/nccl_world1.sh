#!/bin/bash
# shared-factor (RCCL broadcast) code path with a world of ONE rank: exercises init, caller-owned
# factor storage, ranged factorisation + async broadcast handles, adopt/commit.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = pkg.workloads.make("darcy64")
eng = post.HipEngine(pkg, w)
job = post.ShardedPosterior(eng, dist=dist, rank=0, world=1, k_samples=16, group=4, replicate_factor=False)
job.replicate = False            # force the broadcast path even with one rank
job.prepare()
mu, X = job.step(0)
sys.path.insert(0, os.getcwd())
from oracle import bt_oracle as O
Fo = O.tridiagonal_cholesky(w.Q, w.n_blocks)
err = np.linalg.norm(mu[0].cpu().numpy() - O.ldiv(Fo, w.rhs)) / np.linalg.norm(O.ldiv(Fo, w.rhs))
t = time.perf_counter(); [job.step(s) for s in range(1, 4)]; torch.cuda.synchronize()
print(f"shared-factor path, world 1: mean rel err {err:.2e}; {1e3*(time.perf_counter()-t)/3:.2f} ms/step")
dist.destroy_process_group()
assert err < 1e-9
PY
