"""Selected inversion (gmrf_bt_marginal_var "exact") of a darcy256 batch: wall time per problem and the GEMM launches by shape
(profiles/r04_*: the by-shape table of one var_exact call)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = pkg.workloads.make("darcy256")
vals = np.tile(w.Q.data, (batch, 1))
F = pkg.TridiagonalCholeskyFactor(batch=batch); F.set_keep_l(False)
F.factor(w.Q, w.n_blocks, values=vals)
nz = torch.from_numpy(vals).cuda()
out = torch.empty((batch, w.n), dtype=torch.float64, device="cuda")
import ctypes as C
lib = pkg._cabi.load()
def var():
    pkg._cabi.check(lib.gmrf_bt_marginal_var(F._h, pkg._cabi.VAR_EXACT, 0, 0, None, pkg._cabi.ptr(out)))
F.refactor(nz); var()
ts = []
for _ in range(3):
    F.refactor(nz)                      # (back to the split representation: the conversion is part of what is timed, as in bench.py)
    torch.cuda.synchronize(); t0 = time.perf_counter(); var(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t_conv = sorted(ts)[1]
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); var(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t_full = sorted(ts)[1]
print(f"batch {batch}: selected inversion {1e3 * t_conv / batch:.3f} ms per problem from the split factor, {1e3 * t_full / batch:.3f} from the full one")
F.refactor(nz)
F.set_profiling(1)
var()
st = F.stats(); sh = F.gemm_shapes()
F.set_profiling(0)
tot = sum(s["ms"] for s in sh)
print(f"GEMM launches of one call: {tot:.1f} ms = {tot / batch:.3f} ms per problem, {sum(s['flops'] for s in sh) / tot / 1e9:.1f} TF/s time-weighted (booked flops)")
for s in sorted(sh, key=lambda s: -s["ms"]):
    print(f"  class {s['class']:2d} {s['M']:5d} x {s['N']:5d} x {s['K']:5d} tri {s['tri']:2d} lower {s['lower_only']} bounds {s['k_bounds']} x{s['problems']:3d}: "
          f"{s['launches']:4d} launches {s['ms']:8.2f} ms  {s['flops'] / max(s['ms'], 1e-9) / 1e9:6.1f} TF/s")
