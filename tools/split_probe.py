import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
w = pkg.workloads.make("darcy256")
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
nz = torch.from_numpy(w.Q.data).cuda(); rhs = torch.from_numpy(w.rhs).cuda()
for flag, name in [(0, "fused step"), (2, "split step (tile, panel, update)")]:
    pkg._cabi.check(pkg._cabi.load().gmrf_bt_set_eager(F._h, flag))
    for it in range(3):
        F.refactor(nz)
    st = F.stats()
    mu = pkg.ldiv(F, rhs)
    r = w.Q @ mu.cpu().numpy() - w.rhs
    print(f"{name}: factor {st['factor_ms']:.2f} ms, residual {np.linalg.norm(r)/np.linalg.norm(w.rhs):.2e}")
