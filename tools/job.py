"""One or more darcy posterior jobs (refactor + mean + samples) for profiling runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "darcy256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eager = len(sys.argv) > 3 and sys.argv[3] == "eager"
w = pkg.workloads.make(name)
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
F.set_eager(eager)
nz = torch.from_numpy(w.Q.data).cuda(); rhs = torch.from_numpy(w.rhs).cuda()
for _ in range(steps):
    F.refactor(nz)
    mu = pkg.ldiv(F, rhs)
    X = F.sample(64, mean=mu, seed=1, like=rhs)
torch.cuda.synchronize()
print("done", float(mu.abs().max()))
