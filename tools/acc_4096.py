import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse.linalg as spla, scipy.linalg as sla
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package()
w = pkg.workloads.burgers(4096, 3)
F = pkg.tridiagonal_cholesky(w.Q, 3)
mu = pkg.ldiv(F, w.rhs)
qn = abs(w.Q).sum(axis=1).max()
be = lambda v: np.linalg.norm(w.Q @ v - w.rhs) / (qn * np.linalg.norm(v) + np.linalg.norm(w.rhs))
Fo = O.tridiagonal_cholesky(w.Q, 3); mo = O.ldiv(Fo, w.rhs)
ms = spla.splu(w.Q.tocsc()).solve(w.rhs)
print("backward err: HIP %.2e oracle %.2e splu %.2e | HIP vs oracle %.2e, oracle vs splu %.2e" % (be(mu), be(mo), be(ms), np.linalg.norm(mu-mo)/np.linalg.norm(mo), np.linalg.norm(mo-ms)/np.linalg.norm(ms)))
for i in range(3):
    L = np.tril(F.chos[i]); X = np.tril(F.inverses[i]); I = np.eye(4096)
    print("blk %d cond(L) %.1e: |L-Lo|/|L| %.2e ; ||L X - I|| %.2e ||X L - I|| %.2e" % (i, np.linalg.cond(Fo.chos[i]), np.abs(L-Fo.chos[i]).max()/np.abs(Fo.chos[i]).max(), np.abs(L@X-I).max(), np.abs(X@L-I).max()))
