"""Development aid (round 5): one problem's sweeps as ONE persistent launch each (sweep_persist) against a launch per product
(set_eager bit 16) -- bitwise comparison of mean / samples / forward-only / backward-only solves, and the times of both."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
F = None
for name in (sys.argv[1:] or ["burgers512x64", "darcy256"]):
    w = pkg.workloads.make(name)
    F = None                   # (the next handle can claim the chip only when this one has let go of it)
    import gc; gc.collect()
    F = pkg.tridiagonal_cholesky(w.Q, w.n_blocks)
    rhs = torch.from_numpy(w.rhs).cuda()
    res = {}
    for label, bits in (("persist", 0), ("per_product", 65536)):
        F.set_eager(bits)
        times = []
        for it in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter(); mu = pkg.ldiv(F, rhs); ts = time.perf_counter() - t
            s1 = F.stats()
            t = time.perf_counter(); X = F.sample(64, mean=mu, seed=3, like=rhs); tx = time.perf_counter() - t
            s2 = F.stats()
            times.append((ts * 1e3, s1["solve_ms"], tx * 1e3, s2["sample_ms"]))
        X16 = F.sample(16, mean=mu, seed=5, like=rhs)
        tf, tb = [], []
        for it in range(3):
            yf = pkg.forward_solve(F, rhs); tf.append(F.stats()["solve_ms"])
            xb = pkg.backward_solve(F, rhs); tb.append(F.stats()["solve_ms"])
        print(f"{name} [{label}] forward k=1 {min(tf):.3f} ms, backward k=1 {min(tb):.3f} ms", flush=True)
        res[label] = (mu.cpu().numpy(), X.cpu().numpy(), np.concatenate([X16.cpu().numpy().ravel(), yf.cpu().numpy().ravel(), xb.cpu().numpy().ravel()]), s1["sweep_persist"], s2["sweep_persist"], s2["persist_aborts"], s2["persist_cus"])
        ts = np.array(times[1:]).min(axis=0)
        print(f"{name} [{label}] solve wall {ts[0]:.3f} ms (dev {ts[1]:.3f}) | sample64 wall {ts[2]:.3f} ms (dev {ts[3]:.3f}) | stats sweep_persist {s1['sweep_persist']}/{s2['sweep_persist']} launches {s2['sweep_persist_launches']} aborts {s2['persist_aborts']} cus {s2['persist_cus']}", flush=True)
    a, b = res["persist"], res["per_product"]
    print(f"{name}: mean equal {np.array_equal(a[0], b[0])}, samples64 equal {np.array_equal(a[1], b[1])}, samples16 / forward / backward equal {np.array_equal(a[2], b[2])}; "
          f"residual {np.linalg.norm(w.Q @ a[0] - w.rhs) / np.linalg.norm(w.rhs):.2e}", flush=True)
