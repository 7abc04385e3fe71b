import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "burgers512x64"
w = pkg.workloads.make(name); N = w.n_blocks; bs = w.block_size
Fo = O.tridiagonal_cholesky(w.Q, N); x_o = O.ldiv(Fo, w.rhs)
Ql = w.Q.tocsr().astype(np.longdouble); bl = w.rhs.astype(np.longdouble); x = x_o.astype(np.longdouble)
for it in range(6):
    r = bl - Ql @ x; x = x + O.ldiv(Fo, np.asarray(r, dtype=np.float64)).astype(np.longdouble)
xt = np.asarray(x, dtype=np.float64)
rel = lambda a: float(np.linalg.norm(a - xt) / np.linalg.norm(xt))
F = pkg.tridiagonal_cholesky(w.Q, N)
x_g = pkg.ldiv(F, w.rhs)
Lg = [np.tril(F.chos[i]) for i in range(N)]; Cg = [F.Cs[i] for i in range(N - 1)]; Xg = [np.tril(F.inverses[i]) for i in range(N)]
Fg = O.TridiagonalCholeskyFactor(w.n, Lg, Cg)
x_mix = O.ldiv(Fg, w.rhs)                       # GPU factor, LAPACK substitution sweeps
def xsweeps(Xs, Cs, b):                         # explicit-inverse sweeps in NumPy
    y = [Xs[0] @ b[:bs]]
    for i in range(1, N): y.append(Xs[i] @ (b[i*bs:(i+1)*bs] - Cs[i-1] @ y[-1]))
    xx = [None] * N; xx[N-1] = Xs[N-1].T @ y[N-1]
    for i in range(N - 2, -1, -1): xx[i] = Xs[i].T @ (y[i] - Cs[i].T @ xx[i+1])
    return np.concatenate(xx)
x_gx = xsweeps(Xg, Cg, w.rhs)                    # GPU L, C, X ; NumPy products
Xo = [np.tril(sla.solve_triangular(L, np.eye(bs), lower=True)) for L in Fo.chos]
x_ox = xsweeps(Xo, Fo.Cs, w.rhs)                 # oracle factor, explicit inverse sweeps
print(f"{name}: fwd err  oracle {rel(x_o):.2e} | HIP {rel(x_g):.2e} | HIP factor + LAPACK sweeps {rel(x_mix):.2e} | HIP L,C,X + numpy X-sweeps {rel(x_gx):.2e} | oracle factor + X-sweeps {rel(x_ox):.2e}")
