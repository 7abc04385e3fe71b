"""CSR SpMV / SpMM (K6): time and algorithmic bandwidth, fp64 and fp32 values (fp64 accumulation)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
for name in ("darcy256", "burgers4096x512"):
    w = pkg.workloads.make(name)
    for f32 in (False, True):
        S = pkg.CsrMatrix(w.Q, values_f32=f32)
        for k in (1, 4, 64):
            X = torch.randn(k, w.n, dtype=torch.float64, device="cuda").t() if k > 1 else torch.randn(w.n, dtype=torch.float64, device="cuda")
            for _ in range(3): Y = S @ X
            torch.cuda.synchronize(); reps = 20; t = time.perf_counter()
            for _ in range(reps): Y = S @ X
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
            vb = 4 if f32 else 8
            bytes_alg = w.Q.nnz * (vb + 4) + 8 * (w.n + 1) + 16 * w.n * k
            print(f"{name:16s} values {'fp32' if f32 else 'fp64'} k={k:3d}: {dt*1e6:8.1f} us  {bytes_alg/dt/1e9:7.0f} GB/s algorithmic  {2*w.Q.nnz*k/dt/1e9:7.1f} GF/s", flush=True)
        del S
    if name == "darcy256":
        Y = (pkg.CsrMatrix(w.Q, values_f32=True) @ np.ones(w.n))
        print("fp32-value rel diff vs fp64:", np.linalg.norm(Y - w.Q @ np.ones(w.n)) / np.linalg.norm(w.Q @ np.ones(w.n)))
