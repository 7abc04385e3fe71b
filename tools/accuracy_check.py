"""Forward error of the HIP path and of the oracle against an extended-precision reference
(iterative refinement with long-double residuals) -- who is closer to the truth?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
from oracle import bt_oracle as O
pkg = g.load_package()
for name in sys.argv[1:] or ["burgers512x64", "darcy64"]:
    w = pkg.workloads.make(name)
    Fo = O.tridiagonal_cholesky(w.Q, w.n_blocks)
    x_o = O.ldiv(Fo, w.rhs)
    Ql = w.Q.tocsr().astype(np.longdouble); bl = w.rhs.astype(np.longdouble)
    x = x_o.astype(np.longdouble)
    for it in range(6):
        r = bl - Ql @ x
        x = x + O.ldiv(Fo, np.asarray(r, dtype=np.float64)).astype(np.longdouble)
    x_true = np.asarray(x, dtype=np.float64)
    rel = lambda a: float(np.linalg.norm(a - x_true) / np.linalg.norm(x_true))
    line = f"{name}: oracle fwd err {rel(x_o):.2e}"
    try:
        F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
        x_g = pkg.ldiv(F, w.rhs)
        line += f" | HIP fwd err {rel(x_g):.2e} | HIP vs oracle {float(np.linalg.norm(x_g - x_o)/np.linalg.norm(x_o)):.2e}"
        res = lambda v: float(np.linalg.norm(w.Q @ v - w.rhs) / np.linalg.norm(w.rhs))
        line += f" | residuals oracle {res(x_o):.2e} HIP {res(x_g):.2e}"
    except Exception as e:
        line += f" | (no GPU: {type(e).__name__})"
    print(line, flush=True)
