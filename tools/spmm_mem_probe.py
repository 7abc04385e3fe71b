"""Does the k = 64 SpMM time depend on what the process did to device memory before?  Fresh process vs after
allocating and freeing 100 GB through the library (the state bench.py's K6 leg runs in)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
w = pkg.workloads.make("burgers4096x512")
b = w.Q.nnz * 12 + 8 * (w.n + 1) + 16 * w.n * 64

def leg(tag):
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    S = pkg.CsrMatrix(w.Q, stream=st.cuda_stream)
    X = torch.randn(w.n, 64, dtype=torch.float64, device="cuda"); Y = torch.empty_like(X)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10): S.matmul_into(X, Y)
        torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30): S.matmul_into(X, Y)
        e1.record(st); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{tag:40s} {us:7.1f} us  {b / us / 1e3:6.0f} GB/s   X at {X.data_ptr():#x}", flush=True)
    del S, X, Y
    torch.cuda.synchronize(); torch.cuda.set_stream(torch.cuda.default_stream())

leg("fresh process")
wd = pkg.workloads.make("darcy256")
Fs = []
for i in range(4):
    F = pkg.TridiagonalCholeskyFactor(batch=32); F.set_keep_l(False)
    F.factor(wd.Q, wd.n_blocks, values=np.tile(wd.Q.data, (32, 1))); Fs.append(F)
torch.cuda.synchronize()
print("allocated", round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1), "GB")
leg("with 4 x 32 factors resident")
for F in Fs: F.close()
torch.cuda.empty_cache()
leg("after freeing them")
