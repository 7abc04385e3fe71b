"""Phase stamps of the fused panel step (workgroup 1 of step 0, bs = 1024): where do the 13 us after the tile go?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
bs = 1024
rng = np.random.default_rng(0); G = rng.standard_normal((bs, bs)); A = G @ G.T + bs * np.eye(bs)
S = np.tril(A).copy(); Linv = np.zeros((bs, bs)); info = C.c_int32(0)
pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Linv), C.byref(info)))
out = np.zeros(30); pkg._cabi.check(lib.gmrf_test_tile_timing(pkg._cabi.ptr(out), 30))
names = ["tile phase begin", "tile factor + inverse done", "panel products done", "panel tiles in LDS", "update products done", "stores issued"]
for n, v in zip(names, out[24:30]):
    print(f"{n:28s} {v:9.0f} cycles")
