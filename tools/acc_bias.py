import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
rng = np.random.default_rng(0)
M = N = 256
for K, label in [(64, "K=64"), (512, "K=512"), (1024, "K=1024")]:
    A = rng.uniform(0.5, 1.5, (M, K)); B = rng.uniform(0.5, 1.5, (N, K))
    exact = (A.astype(np.longdouble) @ B.astype(np.longdouble).T)
    out = np.zeros((M, N))
    pkg._cabi.check(lib.gmrf_test_gemm(0, M, N, K, 0, 1, 0, 0, 1.0, pkg._cabi.ptr(A), K, pkg._cabi.ptr(B), K, 0.0, pkg._cabi.ptr(out), N))
    npr = A @ B.T
    eg = ((out - exact) / exact).astype(float); en = ((npr - exact) / exact).astype(float)
    print(f"{label}: relative error of positive dot products: GPU mean {eg.mean():+.3e} rms {np.sqrt((eg**2).mean()):.3e} | numpy mean {en.mean():+.3e} rms {np.sqrt((en**2).mean()):.3e}")
