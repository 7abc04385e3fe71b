"""The 64 x 64 tile Cholesky + inverse (tile_potrf_inv) alone: error against NumPy, time per launch, s_memtime stamps of its phases.
Round 5: four waves, each the owner of a 16-column panel (csrc/potrf_step.hpp)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
rng = np.random.default_rng(1); G = rng.standard_normal((64, 64)); A = G @ G.T + 64 * np.eye(64)
t = A.copy(); inv = np.zeros((64, 64)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
L = np.linalg.cholesky(A)
print("info", info.value, "err L", np.abs(t - L).max() / np.abs(L).max(), "err inv", np.abs(inv @ L - np.eye(64)).max(),
      "upper L", np.abs(np.triu(t, 1)).max(), "upper X", np.abs(np.triu(inv, 1)).max())
out = np.zeros(62); pkg._cabi.check(lib.gmrf_test_tile_timing(pkg._cabi.ptr(out), 62))
for p in range(4):
    if out[30 + 3 * p] >= 0: print('panel %d: followed %6d, chain begins %6d, done %6d (%5d), columns stored %6d' % (p, out[54 + p], out[30 + 3 * p], out[31 + 3 * p], out[31 + 3 * p] - out[30 + 3 * p], out[32 + 3 * p]))
print(f"tile kernel: {out[0]:.2f} us per launch (back-to-back)")
names = {15: "kernel begin", 0: "routine begin", 1: "panel 0 led + stored", 2: "panel 1 led + stored", 3: "panel 2 led + stored", 4: "panel 3 led + stored",
         7: "X22 there", 8: "X33 there", 13: "wave 0 at the last barrier", 14: "routine end", 16: "kernel end"}
st = out[1:18]
for i in [15, 0, 1, 2, 3, 4, 7, 8, 13, 14, 16]:
    print(f"  {names[i]:28s} {st[i]:9.0f} cycles")
