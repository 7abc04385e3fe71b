"""Chain workgroup of the persistent in-block Cholesky (csrc/potrf_persist.hpp): s_memtime stamps per step of one
1024-block (gmrf_test_potrf_block), in cycles.  A development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import __graft_entry__ as g

pkg = g.load_package(); lib = pkg._cabi.load()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
A = rng.standard_normal((bs, bs)); A = A @ A.T / bs + 2.0 * np.eye(bs)
for rep in range(2):
    S = np.tril(A).copy(); Linv = np.zeros((bs, bs)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Linv), C.byref(info)))
L = np.linalg.cholesky(A)
print("info", info.value, "L err", np.max(np.abs(np.tril(S) - L)) / np.max(np.abs(L)), "inv err", np.max(np.abs(np.tril(Linv) @ L - np.eye(bs))))
nt = bs // 64
out = np.zeros(8 * (nt - 1) + 1); pkg._cabi.check(lib.gmrf_test_persist_stamps(pkg._cabi.ptr(out), len(out)))
out_base = 0
print("tile 0 done at 0; per step (cycles): wait for flags | operands -> LDS | L[j+1,j] = S X^T | -> LDS | store issued | S - L L^T (wave 3) | tile in LDS | potrf + inverse + publish")
prev = 0.0
for j in range(nt - 1):
    s = out[8 * j + 1: 8 * j + 9].copy()
    seen = s[4]                        # slot 5: the panel in which the NEXT step's operands were seen ready (-1 / 0: not prefetched), relative
    s[4] = s[3]
    d = np.diff(np.concatenate([[prev], s]))
    print(f"step {j:2d}: " + " ".join(f"{x:7.0f}" for x in d) + f"   total {s[7]-prev:7.0f}   next operands seen ready in panel {int(seen)}")
    prev = s[7]
print(f"chain total {prev:.0f} cycles for {nt - 1} steps = {prev / max(nt - 1, 1):.0f} per step")
