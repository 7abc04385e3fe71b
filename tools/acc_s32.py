import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package(); lib = pkg._cabi.load()
ld = np.longdouble
def chol_ld(M):
    n = M.shape[0]; L = np.zeros_like(M)
    for j in range(n):
        L[j, j] = np.sqrt(M[j, j] - L[j, :j] @ L[j, :j])
        L[j+1:, j] = (M[j+1:, j] - L[j+1:, :j] @ L[j, :j]) / L[j, j]
    return L
w = pkg.workloads.make("burgers512x64"); bs = w.block_size; A = w.Q.tocsr()
Fo = O.tridiagonal_cholesky(w.Q, w.n_blocks)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for i in [1, 8, 32, 60, 63]:
    D = A[i*bs:(i+1)*bs, i*bs:(i+1)*bs].toarray(); Cm = Fo.Cs[i-1]
    S = D - Cm @ Cm.T; S = np.tril(S) + np.tril(S, -1).T
    Lt = chol_ld(S.astype(ld)).astype(float)
    Sg = np.tril(S).copy(); Li = np.zeros((bs, bs)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(Sg), pkg._cabi.ptr(Li), C.byref(info)))
    Lg = np.tril(Sg); Xg = np.tril(Li)
    Ln = np.linalg.cholesky(S)
    I = np.eye(bs)
    B = A[(i+1)*bs:(i+2)*bs, i*bs:(i+1)*bs].toarray() if i < w.n_blocks - 1 else None
    msg = "S_%d cond %.1e: L err vs long double: numpy %.2e HIP %.2e | HIP X: ||L X - I|| %.2e" % (i, np.linalg.cond(S), relm(Ln, Lt), relm(Lg, Lt), np.abs(Lt @ Xg - I).max())
    if B is not None:
        Ct = sla.solve_triangular(Lt, B.T, lower=True).T
        msg += " | next C = B X^T err %.2e (trsm with numpy L: %.2e)" % (relm(B @ Xg.T, Ct), relm(sla.solve_triangular(Ln, B.T, lower=True).T, Ct))
    print(msg, flush=True)
