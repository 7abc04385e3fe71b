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
from oracle import bt_oracle as O
_w = pkg.workloads.make("burgers512x64"); _bs = _w.block_size; _A = _w.Q.tocsr(); _Fo = O.tridiagonal_cholesky(_w.Q, _w.n_blocks)
_i = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_D = _A[_i*_bs:(_i+1)*_bs, _i*_bs:(_i+1)*_bs].toarray(); _C = _Fo.Cs[_i-1]
S1 = _D - _C @ _C.T; S1 = np.tril(S1) + np.tril(S1, -1).T; bs = S1.shape[0]
Lt = chol_ld(S1.astype(ld)).astype(float)
S = np.tril(S1).copy(); Li = np.zeros((bs, bs)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Li), C.byref(info)))
Lg = np.tril(S); Xg = np.tril(Li)
relm = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
def model(get_tile):
    Sx = S1.copy(); L = np.zeros_like(Sx)
    for j in range(0, bs, 64):
        Ljj, Xjj = get_tile(Sx[j:j+64, j:j+64], j)
        L[j:j+64, j:j+64] = Ljj
        if j + 64 < bs:
            P = Sx[j+64:, j:j+64] @ Xjj.T; L[j+64:, j:j+64] = P; Sx[j+64:, j+64:] -= P @ P.T
    return L
def np_tile(T, j):
    Ljj = np.linalg.cholesky(T); return Ljj, sla.solve_triangular(Ljj, np.eye(64), lower=True)
def gpu_tile(T, j):
    t = np.ascontiguousarray(np.tril(T) + np.tril(T, -1).T); inv = np.zeros((64, 64)); inf = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(inf)))
    return np.tril(t), np.tril(inv)
def gpu_tile_L_np_inv(T, j):
    Lj, _ = gpu_tile(T, j); return Lj, sla.solve_triangular(Lj, np.eye(64), lower=True)
def np_tile_gpu_inv(T, j):
    Lj, Xj = gpu_tile(T, j); return np.linalg.cholesky(T), Xj
print("err vs long double: HIP potrf_block %.2e | numpy cholesky %.2e" % (relm(Lg, Lt), relm(np.linalg.cholesky(S1), Lt)))
for nm, f in [("numpy tiles + numpy inverse", np_tile), ("GPU tiles + GPU inverse (numpy products)", gpu_tile), ("GPU tile L + numpy inverse", gpu_tile_L_np_inv), ("numpy tile L + GPU inverse", np_tile_gpu_inv)]:
    Lm = model(f); print("  model with %-42s: %.2e   (vs HIP potrf_block %.2e)" % (nm, relm(Lm, Lt), relm(Lm, Lg)))
Lm = model(gpu_tile)
for j in range(0, bs, 64):
    dd = relm(Lg[j:j+64, j:j+64], Lm[j:j+64, j:j+64]); pp = relm(Lg[j+64:, j:j+64], Lm[j+64:, j:j+64]) if j + 64 < bs else 0.0
    print("  HIP vs (GPU tiles, numpy products) model: tile col %3d diag %.2e panel %.2e ; panel rows detail:" % (j, dd, pp),
          " ".join("%.1e" % relm(Lg[r:r+64, j:j+64], Lm[r:r+64, j:j+64]) for r in range(j + 64, bs, 64)))
print("cond of diagonal 64-tiles of S:", " ".join("%.1e" % np.linalg.cond(S1[j:j+64, j:j+64]) for j in range(0, bs, 64)))
Ltt = Lt
for j in range(0, bs, 64):
    Tj = Ltt[j:j+64, j:j+64] @ Ltt[j:j+64, j:j+64].T      # exact Schur tile at step j
    Lgj, Xgj = gpu_tile(Tj, j); Lnj = np.linalg.cholesky(Tj); Ltj = chol_ld(Tj.astype(ld)).astype(float)
    print("  schur tile %3d cond %.1e: tile L err numpy %.2e gpu %.2e" % (j, np.linalg.cond(Tj), relm(Lnj, Ltj), relm(Lgj, Ltj)))
T0 = S1[:64, :64]; L0, X0 = gpu_tile(T0, 0); Ln = np.linalg.cholesky(T0); Xn = sla.solve_triangular(Ln, np.eye(64), lower=True)
I = np.eye(64)
print("tile 0: gpu X: ||XL-I|| %.2e ||LX-I|| %.2e | numpy X: %.2e %.2e | |Xg-Xn|/|X| %.2e" % (np.abs(X0 @ L0 - I).max(), np.abs(L0 @ X0 - I).max(), np.abs(Xn @ Ln - I).max(), np.abs(Ln @ Xn - I).max(), relm(X0, Xn)))
