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
D1 = A[bs:2*bs, bs:2*bs].toarray(); C0 = Fo.Cs[0]
S1 = D1 - C0 @ C0.T; S1 = np.tril(S1) + np.tril(S1, -1).T
Lt = chol_ld(S1.astype(ld)).astype(float)
Lnp = np.linalg.cholesky(S1)
S = np.tril(S1).copy(); Li = np.zeros((bs, bs)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Li), C.byref(info)))
Lg = np.tril(S)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
print("S1 (cond %.1e): L err vs long double: numpy %.2e  gpu %.2e" % (np.linalg.cond(S1), relm(Lnp, Lt), relm(Lg, Lt)))
for j in range(0, bs, 64):
    print("  tile col %3d: diag tile err numpy %.2e gpu %.2e | panel below: numpy %.2e gpu %.2e" % (j, relm(Lnp[j:j+64, j:j+64], Lt[j:j+64, j:j+64]), relm(Lg[j:j+64, j:j+64], Lt[j:j+64, j:j+64]),
          relm(Lnp[j+64:, j:j+64], Lt[j+64:, j:j+64]) if j + 64 < bs else 0, relm(Lg[j+64:, j:j+64], Lt[j+64:, j:j+64]) if j + 64 < bs else 0))
T0 = S1[:64, :64]
print("  cond of first 64 tile %.2e ; cond of its factor %.2e" % (np.linalg.cond(T0), np.linalg.cond(Lt[:64,:64])))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "S1.npy"), S1)
