import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package(); lib = pkg._cabi.load()
w = pkg.workloads.make(sys.argv[1] if len(sys.argv) > 1 else "burgers512x64")
bs = w.block_size
A = w.Q.tocsr()
D0 = A[:bs, :bs].toarray()
# (1) single block through the potrf_block hook
S = np.tril(D0).copy(); Li = np.zeros((bs, bs)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Li), C.byref(info)))
L = np.linalg.cholesky(D0)
print("block0: cond(D0) %.2e  |L-Lref|max/|L|max %.2e   ||L L^T - D||/||D|| gpu %.2e ref %.2e   ||Linv L - I|| %.2e (ref %.2e)" % (
    np.linalg.cond(D0), np.abs(np.tril(S) - L).max() / np.abs(L).max(),
    np.linalg.norm(np.tril(S) @ np.tril(S).T - D0) / np.linalg.norm(D0), np.linalg.norm(L @ L.T - D0) / np.linalg.norm(D0),
    np.abs(np.tril(Li) @ L - np.eye(bs)).max(), np.abs(sla.solve_triangular(L, np.eye(bs), lower=True) @ L - np.eye(bs)).max()))
# (2) first 64x64 tile alone
T0 = D0[:64, :64].copy(); t = T0.copy(); inv = np.zeros((64, 64))
pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
Lt = np.linalg.cholesky(T0)
print("tile0: cond %.2e |L-Lref|/|L| %.2e  ||LL^T-T||/||T|| gpu %.2e ref %.2e  ||inv L - I|| %.2e" % (np.linalg.cond(T0), np.abs(np.tril(t) - Lt).max() / np.abs(Lt).max(),
      np.linalg.norm(np.tril(t) @ np.tril(t).T - T0) / np.linalg.norm(T0), np.linalg.norm(Lt @ Lt.T - T0) / np.linalg.norm(T0), np.abs(inv @ Lt - np.eye(64)).max()))
# (3) whole factor: blockwise errors
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks); Fo = O.tridiagonal_cholesky(w.Q, w.n_blocks)
for i in [0, 1, 2, w.n_blocks // 2, w.n_blocks - 1]:
    Lg, Lo = np.tril(F.chos[i]), Fo.chos[i]
    Xg = np.tril(F.inverses[i])
    msg = "blk %3d: |L-Lo|/|L| %.2e  ||Xg Lo - I|| %.2e" % (i, np.abs(Lg - Lo).max() / np.abs(Lo).max(), np.abs(Xg @ Lo - np.eye(bs)).max())
    if i < w.n_blocks - 1:
        msg += "  |C-Co|/|C| %.2e" % (np.abs(F.Cs[i] - Fo.Cs[i]).max() / np.abs(Fo.Cs[i]).max())
    print(msg)
