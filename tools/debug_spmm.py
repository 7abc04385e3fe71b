"""Debug aid: column-major SpMM (lane-group kernel) against SciPy at growing sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for nt in (8, 128, 512):
    w = pkg.workloads.burgers(4096, nt)
    Qr = w.Q.tocsr()
    rng = np.random.default_rng(1)
    X = rng.standard_normal((w.n, 8))
    ref = Qr @ X
    for f32 in (False, True):
        S = pkg.CsrMatrix(Qr, values_f32=f32)
        tol_ref = ref
        if f32:
            Q32 = Qr.copy(); Q32.data = Q32.data.astype(np.float32).astype(np.float64); tol_ref = Q32 @ X
        for k in (8, 64):
            reps = k // 8
            Xh = np.tile(X, (1, reps))
            Xd = torch.from_numpy(np.ascontiguousarray(Xh.T)).cuda().t()     # column-major on the device
            Yd = (S @ Xd).cpu().numpy()
            Yh = S @ np.asfortranarray(Xh)                                   # column-major on the host
            Yr = S @ np.ascontiguousarray(Xh)                                # node-major
            print(f"nt={nt} n={w.n} f32={f32} k={k}: dev-cols {rel(Yd[:, :8], tol_ref):.2e} (last 8: {rel(Yd[:, -8:], tol_ref):.2e})  "
                  f"host-cols {rel(Yh[:, :8], tol_ref):.2e}  rows {rel(Yr[:, :8], tol_ref):.2e} (last 8: {rel(Yr[:, -8:], tol_ref):.2e})", flush=True)
        del S
