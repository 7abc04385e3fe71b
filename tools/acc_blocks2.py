import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package(); lib = pkg._cabi.load()
w = pkg.workloads.make("burgers512x64"); bs = w.block_size; A = w.Q.tocsr()
F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks); Fo = O.tridiagonal_cholesky(w.Q, w.n_blocks)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
L0g, X0g, C0g, L1g = np.tril(F.chos[0]), np.tril(F.inverses[0]), F.Cs[0], np.tril(F.chos[1])
B1 = A[bs:2*bs, :bs].toarray(); D1 = A[bs:2*bs, bs:2*bs].toarray()
C0_from_gpuX = B1 @ X0g.T
print("C0: gpu vs numpy(B @ X0g^T) %.2e | gpu vs oracle %.2e | numpy(B X0g^T) vs oracle %.2e" % (relm(C0g, C0_from_gpuX), relm(C0g, Fo.Cs[0]), relm(C0_from_gpuX, Fo.Cs[0])))
S1_np = D1 - C0g @ C0g.T
L1_np = np.linalg.cholesky(S1_np)
print("L1: gpu vs chol(D1 - C0g C0g^T in numpy) %.2e | that numpy L1 vs oracle %.2e | gpu vs oracle %.2e" % (relm(L1g, L1_np), relm(L1_np, Fo.chos[1]), relm(L1g, Fo.chos[1])))
S = np.tril(S1_np).copy(); Li = np.zeros((bs, bs)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Li), C.byref(info)))
print("potrf_block hook on the numpy S1: vs numpy chol %.2e ; cond(S1) %.2e" % (relm(np.tril(S), L1_np), np.linalg.cond(S1_np)))
# GEMM hook: S = D1 - C0g C0g^T
out = D1.copy(); Cc = np.ascontiguousarray(C0g)
pkg._cabi.check(lib.gmrf_test_gemm(0, bs, bs, bs, 0, 1, 0, 0, -1.0, pkg._cabi.ptr(Cc), bs, pkg._cabi.ptr(Cc), bs, 1.0, pkg._cabi.ptr(out), bs))
S1_ld = (D1.astype(np.longdouble) - C0g.astype(np.longdouble) @ C0g.astype(np.longdouble).T).astype(float)
print("S1: gpu gemm vs longdouble %.2e | numpy vs longdouble %.2e ; max|D1| %.2e max|S1| %.2e  min diag S1 %.2e" % (relm(out, S1_ld), relm(S1_np, S1_ld), np.abs(D1).max(), np.abs(S1_ld).max(), np.diag(S1_ld).min()))
e_g = np.abs(out - S1_ld); e_n = np.abs(S1_np - S1_ld)
print("   elementwise abs err: gpu max %.3e mean %.3e | numpy max %.3e mean %.3e" % (e_g.max(), e_g.mean(), e_n.max(), e_n.mean()))
