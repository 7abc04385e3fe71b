"""Is the k = 64 node-major SpMM sensitive to what ran before it?  Times 10 launches (HIP events) when the GPU was
idle, right after a burst of fp64 MFMA work (a batch-32 factorisation job), and after pauses."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
w = pkg.workloads.make("burgers4096x512")
st = torch.cuda.current_stream()
S = pkg.CsrMatrix(w.Q, stream=st.cuda_stream)
X = torch.randn(w.n, 64, dtype=torch.float64, device="cuda")
x1 = torch.randn(w.n, dtype=torch.float64, device="cuda")
b = w.Q.nnz * 12 + 8 * (w.n + 1) + 16 * w.n * 64

def timed(tag, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        S @ X
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{tag:44s} {us:7.1f} us  {b / us / 1e3:6.0f} GB/s", flush=True)

for _ in range(3): S @ X
torch.cuda.synchronize()
timed("cold (3 warm-up launches)")
timed("again"); timed("again, 50 reps", 50); timed("again")
wd = pkg.workloads.make("darcy256")
vals = np.tile(wd.Q.data, (32, 1))
F = pkg.TridiagonalCholeskyFactor(batch=32)
F.set_keep_l(False)
F.factor(wd.Q, wd.n_blocks, values=vals)
nz = torch.from_numpy(vals).cuda()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(12): F.refactor(nz)
    torch.cuda.synchronize()
    print(f"  burst: 12 batch-32 factorisations {time.perf_counter() - t0:.2f} s")
    timed("right after the burst")
    timed("again")
    time.sleep(0.5); timed("after 0.5 s idle")
    time.sleep(2.0); timed("after 2 s idle"); timed("again, 50 reps", 50)
