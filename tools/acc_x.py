import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
pkg = g.load_package()
w = pkg.workloads.make(sys.argv[1] if len(sys.argv) > 1 else "burgers512x64"); bs = w.block_size; A = w.Q.tocsr()
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
I = np.eye(bs)
for i in [0, 1, 2, 8, 32, 63]:
    L = np.tril(F.chos[i]); X = np.tril(F.inverses[i])
    Xs = np.tril(sla.solve_triangular(L, I, lower=True))
    msg = "blk %2d cond(L) %.1f: gpu X: ||XL-I|| %.2e ||LX-I|| %.2e | substitution X: %.2e %.2e" % (i, np.linalg.cond(L), np.abs(X @ L - I).max(), np.abs(L @ X - I).max(), np.abs(Xs @ L - I).max(), np.abs(L @ Xs - I).max())
    if i > 0:
        B = A[i*bs:(i+1)*bs, (i-1)*bs:i*bs].toarray()
        Lp = np.tril(F.chos[i-1])
        Cg = F.Cs[i-1]
        Cs_ = sla.solve_triangular(Lp, B.T, lower=True).T
        msg += " | C_{i-1}: gpu vs trsm(gpu L) %.2e ; residual ||C L^T - B||/||B|| gpu %.2e trsm %.2e" % (np.abs(Cg - Cs_).max() / np.abs(Cs_).max(), np.linalg.norm(Cg @ Lp.T - B) / np.linalg.norm(B), np.linalg.norm(Cs_ @ Lp.T - B) / np.linalg.norm(B))
    print(msg)
