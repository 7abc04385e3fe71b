"""Device posterior assembly: time and algorithmic bandwidth of the numeric phase."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
for ns, nt in ((512, 64), (4096, 64)):
    gn = pkg.workloads.burgers_gauss_newton(ns, nt)
    x = gn["x_prior"]; J = gn["jacobian"](x); r = gn["residual"](x)
    t = time.perf_counter(); asm = pkg.PosteriorAssembler(gn["Q"], J); t_sym = time.perf_counter() - t
    qd = torch.from_numpy(gn["Q"].data).cuda(); jd = torch.from_numpy(J.data).cuda()
    qx = torch.from_numpy(gn["Qx_prior"]).cuda(); xd = torch.from_numpy(x).cuda(); od = torch.from_numpy(-r).cuda()
    for _ in range(3): a = asm.precision(qd, jd, gn["noise"]); b = asm.rhs(qx, jd, xd, od, gn["noise"])
    torch.cuda.synchronize(); reps = 50
    t = time.perf_counter()
    for _ in range(reps): a = asm.precision(qd, jd, gn["noise"])
    torch.cuda.synchronize(); tp = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps): b = asm.rhs(qx, jd, xd, od, gn["noise"])
    torch.cuda.synchronize(); tr = (time.perf_counter() - t) / reps
    # algorithmic bytes: per product 2 indices (4 B) + 2 gathered values (8 B); per entry pptr, qmap, q, out
    bytes_p = asm.n_products * 24 + asm.nnz_out * 32
    t = time.perf_counter(); A = (gn["Q"] + gn["noise"] * (J.T @ J)).tocsc(); t_cpu = time.perf_counter() - t
    print(f"burgers {ns}x{nt}: n={gn['n']} nnz(Q)={gn['Q'].nnz} nnz(J)={J.nnz} nnz(A)={asm.nnz_out} products={asm.n_products} | "
          f"symbolic {t_sym*1e3:.0f} ms once | precision {tp*1e6:.1f} us ({bytes_p/tp/1e9:.0f} GB/s algorithmic) | "
          f"rhs {tr*1e6:.1f} us | SciPy Q + noise*J'J on the host {t_cpu*1e3:.1f} ms", flush=True)
