"""Quick GPU probe: times of the darcy job phases (not the bench; a development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "darcy256"
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 64
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # set_eager bits 1, 2 (split step, sweeps on sweep_mm)
t = time.time(); w = pkg.workloads.make(name); print(f"{name}: n={w.n} N={w.n_blocks} bs={w.block_size} nnz={w.Q.nnz} gen {time.time()-t:.2f}s", flush=True)
t = time.time(); F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks); print(f"first factor (analyse+capture) {time.time()-t:.3f}s", flush=True)
import torch
nz = torch.from_numpy(w.Q.data).cuda(); rhs = torch.from_numpy(w.rhs).cuda()
for mode in ("graph", "eager"):
    F.set_eager(int(mode == "eager") | flags)
    for it in range(3):
        t = time.time(); F.refactor(nz); tf = time.time() - t
        t = time.time(); mu = pkg.ldiv(F, rhs); ts = time.time() - t
        t = time.time(); X = F.sample(ks, mean=mu, seed=1, like=rhs); tx = time.time() - t
        st = F.stats()
        print(f"[{mode}] factor wall {tf*1e3:.2f} ms (dev {st['factor_ms']:.2f}) | solve wall {ts*1e3:.2f} ms (dev {st['solve_ms']:.3f}) | sample{ks} wall {tx*1e3:.2f} ms (dev {st['sample_ms']:.3f})", flush=True)
F.set_eager(1 | flags); F.set_profiling(1)
F.refactor(nz); mu = pkg.ldiv(F, rhs); X = F.sample(ks, mean=mu, seed=1, like=rhs)
st = F.stats()
print("profile:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
if False:
    print(f"gemm kernel: {st['gemm_flops']/st['gemm_ms']/1e9:.2f} TFLOP/s over {st['gemm_launches']} launches; tile kernel avg {1e3*st['tile_ms']/max(1,st['tile_launches']):.2f} us")
    print(f"sweep kernels: {st['sweep_kernel_bytes']/st['sweep_kernel_ms']/1e6:.1f} GB/s over {st['sweep_launches']} launches")
F.set_profiling(0)
r = w.Q @ mu.cpu().numpy() - w.rhs
print("residual rel", np.linalg.norm(r) / np.linalg.norm(w.rhs))
