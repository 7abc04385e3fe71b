import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
ld = np.longdouble
def chol_ld(M):
    n = M.shape[0]; L = np.zeros_like(M)
    for j in range(n):
        L[j, j] = np.sqrt(M[j, j] - L[j, :j] @ L[j, :j])
        L[j+1:, j] = (M[j+1:, j] - L[j+1:, :j] @ L[j, :j]) / L[j, j]
    return L
rng = np.random.default_rng(0)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for kappa in [1e1, 1e3, 1e5, 1e7]:
    Qm, _ = np.linalg.qr(rng.standard_normal((64, 64)))
    A = (Qm * np.geomspace(1, kappa, 64)) @ Qm.T; A = 0.5 * (A + A.T)
    Lref = chol_ld(A.astype(ld)).astype(float)
    t = A.copy(); inv = np.zeros((64, 64)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
    Lnp = np.linalg.cholesky(A)
    Xref = sla.solve_triangular(Lref, np.eye(64), lower=True)
    print("tile  cond %.0e: L err gpu %.2e numpy %.2e | inverse: gpu ||X L - I|| %.2e, numpy trtri %.2e" % (kappa, relm(np.tril(t), Lref), relm(Lnp, Lref), np.abs(inv @ Lref - np.eye(64)).max(), np.abs(Xref @ Lref - np.eye(64)).max()))
for kappa in [1e1, 1e3, 1e5]:
    n = 256
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Qm * np.geomspace(1, kappa, n)) @ Qm.T; A = 0.5 * (A + A.T)
    Lref = chol_ld(A.astype(ld)).astype(float)
    S = np.tril(A).copy(); Li = np.zeros((n, n)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_block(0, n, pkg._cabi.ptr(S), pkg._cabi.ptr(Li), C.byref(info)))
    Lnp = np.linalg.cholesky(A)
    print("block256 cond %.0e: L err gpu %.2e numpy %.2e | gpu ||X L - I|| %.2e" % (kappa, relm(np.tril(S), Lref), relm(Lnp, Lref), np.abs(np.tril(Li) @ Lref - np.eye(n)).max()))
