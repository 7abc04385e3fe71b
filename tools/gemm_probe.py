import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
M = N = K = 1024
rng = np.random.default_rng(0)
out = np.zeros(18)
for label, A, B in [("random", rng.standard_normal((M, K)), rng.standard_normal((N, K))), ("ones", np.ones((M, K)), np.ones((N, K))), ("zeros", np.zeros((M, K)), np.zeros((N, K)))]:
    Cm = np.zeros((M, N))
    for tri in [0, 256 + 512 + 1024]:
        for rep in range(3):
            pkg._cabi.check(lib.gmrf_test_gemm(0, M, N, K, 0, 1, tri, 0, 1.0, pkg._cabi.ptr(A), K, pkg._cabi.ptr(B), K, 0.0, pkg._cabi.ptr(Cm), N))
        pkg._cabi.check(lib.gmrf_test_tile_timing(pkg._cabi.ptr(out), 18))
        # g_tile_stamps[0], [1] relative to [15] (=0 here unless tile test ran): out[1], out[2]
        cyc, ticks = out[1] - 0, out[2] - 0
        print(f"{label:7s} tri={tri:5d}: block0 {cyc:.0f} shader cycles, {ticks/100:.2f} us, clock {cyc/ticks*0.1:.2f} GHz, {cyc/1024:.1f} cycles/MFMA")
