import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package()
w = pkg.workloads.make("burgers512x64"); bs = w.block_size; N = w.n_blocks
F = pkg.tridiagonal_cholesky(w.Q, N)
L0 = [np.tril(F.chos[i]) for i in range(N)]
Q2 = w.Q.tolil(copy=True)
Q2[bs:2*bs, bs:2*bs] = w.Q[bs:2*bs, bs:2*bs].toarray() * (1 + 1e-10)
Q2 = Q2.tocsc()
F2 = pkg.tridiagonal_cholesky(Q2, N)
d = [np.abs(np.tril(F2.chos[i]) - L0[i]).max() / np.abs(L0[i]).max() for i in range(N)]
print("HIP, scale perturbation of D_1 by 1e-10:", " ".join("%.1e" % d[i] for i in [0, 1, 2, 3, 4, 8, 16, 32, 48, 63]))
Fo = O.tridiagonal_cholesky(w.Q, N)
e = [np.abs(L0[i] - Fo.chos[i]).max() / np.abs(Fo.chos[i]).max() for i in range(N)]
print("HIP vs oracle per block:", " ".join("%.1e" % e[i] for i in range(0, N, 4)))
