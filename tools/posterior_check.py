"""Development aid (round 5): gmrf_bt_posterior (mean + samples in one call, the samples' sweep beside the mean's) against ldiv + sample:
bitwise comparison and wall / device times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
F = None
for name in (sys.argv[1:] or ["darcy256"]):
    w = pkg.workloads.make(name)
    F = None
    import gc; gc.collect()
    F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
    rhs = torch.from_numpy(w.rhs).cuda()
    sep, one = [], []
    for it in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        mu = pkg.ldiv(F, rhs); X = F.sample(64, mean=mu, seed=3, like=rhs)
        torch.cuda.synchronize(); sep.append((time.perf_counter() - t) * 1e3)
    for it in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        mu1, X1 = F.posterior(rhs, 64, seed=3)
        torch.cuda.synchronize(); one.append(((time.perf_counter() - t) * 1e3, F.stats()["solve_ms"]))
    st = F.stats()
    print(f"{name}: ldiv + sample64 wall {min(sep[1:]):.3f} ms | posterior wall {min(o[0] for o in one[1:]):.3f} ms (dev {min(o[1] for o in one[1:]):.3f}) | "
          f"equal {torch.equal(mu, mu1) and torch.equal(X, X1)} | sweep_persist {st['sweep_persist']} aborts {st['persist_aborts']}", flush=True)
