"""Child process of test_persistent_launch_abort_falls_back (GMRF_PERSIST_SPIN_MS=0 in its environment: the first wait inside a
persistent launch that actually has to wait gives up at once and raises the abort word).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
lib = pkg._cabi.load()


def aborts(F):
    n = C.c_int32(-1)
    pkg._cabi.check(lib.gmrf_test_persist_aborts(F._h, C.byref(n)))
    return int(n.value)


out = {}
# one problem: one persistent launch per block (darcy64: 16 blocks of 256)
w = pkg.workloads.make("darcy64")
F = pkg.TridiagonalCholeskyFactor()
F.factor(w.Q, w.n_blocks)
out["single_aborts_after_first"] = aborts(F)
mu = pkg.ldiv(F, w.rhs)
F.refactor(w.Q.data)                       # the handle keeps the launch-per-step form: no second abort
out["single_aborts_after_second"] = aborts(F)
mu2 = pkg.ldiv(F, w.rhs)
Fs = pkg.TridiagonalCholeskyFactor()
Fs.set_eager(8192)                         # the launch-per-step form up front
Fs.factor(w.Q, w.n_blocks)
out["single_aborts_of_the_step_form"] = aborts(Fs)
out["single_equal"] = bool(np.array_equal(mu, pkg.ldiv(Fs, w.rhs)) and np.array_equal(mu, mu2)
                           and np.array_equal(F.chos[7], Fs.chos[7]))
out["single_logdet_equal"] = bool(F.logdet() == Fs.logdet())
keys = ("persist_aborts", "persist_route", "persist_cus", "persist_refused")
out["single_stats"] = {k: F.stats()[k] for k in keys}
F.set_eager(0); F.refactor(w.Q.data)       # (bit 13 clear: a handle that gave a persistent launch up does not get the form back)
out["single_after_set_eager_0"] = {k: F.stats()[k] for k in keys}
# stepwise (the shared-factor job): every range is packed right after its step, as HipEngine.share_range does
import torch
Fp = pkg.TridiagonalCholeskyFactor()
Fp.factor_begin(w.Q, w.n_blocks)
Fq = pkg.TridiagonalCholeskyFactor(); Fq.set_eager(8192)
Fq.factor_begin(w.Q, w.n_blocks)
same = True
for i0 in range(0, w.n_blocks, 4):
    imgs = []
    for G in (Fp, Fq):
        G.factor_step_async(i0, i0 + 4)
        buf = torch.zeros((1, G.packed_size(i0, i0 + 4)), dtype=torch.float64, device="cuda")
        G.pack_blocks_async(i0, i0 + 4, buf)
        G.synchronize()
        imgs.append(buf.cpu().numpy())
    same = same and bool(np.array_equal(imgs[0], imgs[1]))
Fp.factor_end(); Fq.factor_end()
out["stepwise_aborts"] = aborts(Fp)
out["stepwise_images_equal"] = same
out["stepwise_solve_equal"] = bool(np.array_equal(pkg.ldiv(Fp, w.rhs), pkg.ldiv(Fq, w.rhs)))
# a small batch: one persistent launch per 256-column panel (burgers512x64 as a batch of two)
w = pkg.workloads.make("burgers512x64")
vals = np.stack([w.Q.data, w.Q.data * 1.25])
rhs = np.stack([w.rhs, w.rhs * 2.0])
Fb = pkg.TridiagonalCholeskyFactor(batch=2).factor(w.Q, w.n_blocks, values=vals)
out["batch_aborts"] = aborts(Fb)
xb = Fb.solve_batch(rhs[:, None, :])
Fd = pkg.TridiagonalCholeskyFactor(batch=2)
Fd.set_eager(8192)                         # persistent launches off: potrf_diag128 + GEMM
Fd.factor(w.Q, w.n_blocks, values=vals)
out["batch_equal"] = bool(np.array_equal(xb, Fd.solve_batch(rhs[:, None, :])))
Fb.select_problem(1); Fd.select_problem(1)
out["batch_block_equal"] = bool(np.array_equal(Fb.chos[40], Fd.chos[40]))
print(json.dumps(out))
