import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
rng = np.random.default_rng(1); G = rng.standard_normal((64, 64)); A = G @ G.T + 64 * np.eye(64)
t = A.copy(); inv = np.zeros((64, 64)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
L = np.linalg.cholesky(A)
print("err L", np.abs(t - L).max() / np.abs(L).max(), "err inv", np.abs(inv @ L - np.eye(64)).max())
out = np.zeros(24); pkg._cabi.check(lib.gmrf_test_tile_timing(pkg._cabi.ptr(out), 24))
print('pf0 detail: loads done at %d, loop done at %d, stores done at %d (cycles)' % (out[18], out[19], out[20]))
print(f"tile kernel: {out[0]:.2f} us per launch (back-to-back)")
names = ["start", "pf0 begin", "pf0 end", "after B1(0)", "pf1 begin", "pf1 end", "after B1(1)", "pf2 begin", "pf2 end", "after B1(2)",
         "pf3 begin", "pf3 end", "after B1(3)", "assembly begin", "assembly end", "kernel begin", "kernel end"]
st = out[1:18]
for i in [15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16]:
    print(f"  {names[i]:16s} {st[i]:9.0f} cycles")
