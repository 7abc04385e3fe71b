"""CSR SpMV / SpMM (K6): time and algorithmic bandwidth, fp64 and fp32 values (fp64 accumulation);
column-major right-hand sides (lane-group kernel) and node-major ones (LDS-tiled kernel).
bytes = nnz (vbytes + 4) + 8 (n + 1) + 16 n k   (SURVEY 8d)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ("darcy256", "burgers4096x512")
out = []
for name in names:
    w = pkg.workloads.make(name)
    for f32 in (False, True):
        S = pkg.CsrMatrix(w.Q, values_f32=f32)
        for k, layout in ((1, "vector"), (4, "cols"), (64, "cols"), (16, "rows"), (50, "rows"), (64, "rows")):
            if layout == "vector":
                X = torch.randn(w.n, dtype=torch.float64, device="cuda")
            elif layout == "cols":
                X = torch.randn(k, w.n, dtype=torch.float64, device="cuda").t()
            else:
                X = torch.randn(w.n, k, dtype=torch.float64, device="cuda")
            for _ in range(3): Y = S @ X
            torch.cuda.synchronize(); reps = 20; t = time.perf_counter()
            for _ in range(reps): Y = S @ X
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
            vb = 4 if f32 else 8
            bytes_alg = w.Q.nnz * (vb + 4) + 8 * (w.n + 1) + 16 * w.n * k
            rec = {"matrix": name, "values": "fp32" if f32 else "fp64", "k": k, "layout": layout, "us": dt * 1e6,
                   "alg_GBps": bytes_alg / dt / 1e9, "GFps": 2 * w.Q.nnz * k / dt / 1e9}
            out.append(rec)
            print(f"{name:16s} {rec['values']} k={k:3d} {layout:6s}: {dt*1e6:8.1f} us  {rec['alg_GBps']:7.0f} GB/s algorithmic  {rec['GFps']:7.1f} GF/s", flush=True)
        del S
print("JSON " + json.dumps(out))
