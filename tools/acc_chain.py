"""Error of the factor blocks against a long-double block factorisation (first NB blocks):
oracle (LAPACK), NumPy model of the device algorithm, and the HIP path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "burgers512x64"; NB = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = pkg.workloads.make(name); bs = w.block_size; A = w.Q.tocsr(); N = w.n_blocks
ld = np.longdouble
def chol_ld(M):
    n = M.shape[0]; L = np.zeros_like(M)
    for j in range(n):
        L[j, j] = np.sqrt(M[j, j] - L[j, :j] @ L[j, :j])
        L[j+1:, j] = (M[j+1:, j] - L[j+1:, :j] @ L[j, :j]) / L[j, j]
    return L
def fwd_ld(L, B):
    n = L.shape[0]; X = np.zeros_like(B)
    for i in range(n): X[i] = (B[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X
def mm_ld(P, Q):   # long double matmul without BLAS
    return np.einsum("ik,kj->ij", P, Q, optimize=False) if False else (P[:, :, None] * Q[None, :, :]).sum(1) if P.shape[0] <= 64 else np.stack([ (P[i][:, None] * Q).sum(0) for i in range(P.shape[0]) ])
Lt = []; Ct = []
for i in range(NB):
    D = A[i*bs:(i+1)*bs, i*bs:(i+1)*bs].toarray().astype(ld)
    if i > 0:
        B = A[i*bs:(i+1)*bs, (i-1)*bs:i*bs].toarray().astype(ld)
        C = fwd_ld(Lt[-1], B.T).T; Ct.append(C)
        D = D - mm_ld(C, C.T)
    Lt.append(chol_ld(D))
Fo = O.tridiagonal_cholesky(w.Q, N)
# NumPy model of the device algorithm (tile potrf with tile inverses, doubling, C = B X^T)
def potrf_tiles(S):
    n = S.shape[0]; S = S.copy(); L = np.zeros_like(S); Xd = {}
    for j in range(0, n, 64):
        Ljj = np.linalg.cholesky(S[j:j+64, j:j+64]); Xjj = sla.solve_triangular(Ljj, np.eye(64), lower=True)
        L[j:j+64, j:j+64] = Ljj; Xd[j] = Xjj
        if j + 64 < n:
            P = S[j+64:, j:j+64] @ Xjj.T; L[j+64:, j:j+64] = P; S[j+64:, j+64:] -= P @ P.T
    return L, Xd
def inv_doubling(L, Xd):
    n = L.shape[0]; X = np.zeros_like(L)
    for j, Xjj in Xd.items(): X[j:j+64, j:j+64] = Xjj
    h = 64
    while h < n:
        for o in range(0, n, 2*h):
            T = L[o+h:o+2*h, o:o+h] @ X[o:o+h, o:o+h]; X[o+h:o+2*h, o:o+h] = -X[o+h:o+2*h, o+h:o+2*h] @ T
        h *= 2
    return X
Lm = []; Xm = None
for i in range(NB):
    D = A[i*bs:(i+1)*bs, i*bs:(i+1)*bs].toarray()
    if i > 0:
        B = A[i*bs:(i+1)*bs, (i-1)*bs:i*bs].toarray(); C = B @ Xm.T; D = D - C @ C.T
    L, Xd = potrf_tiles(np.tril(D) + np.tril(D, -1).T); Xm = inv_doubling(L, Xd); Lm.append(L)
F = pkg.tridiagonal_cholesky(w.Q, N)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for i in range(NB):
    T = Lt[i].astype(float)
    print("blk %d: L err vs long double:  oracle %.2e | numpy model %.2e | HIP %.2e" % (i, relm(Fo.chos[i], T), relm(Lm[i], T), relm(np.tril(F.chos[i]), T)))
