"""Device-resident GEMM timing on the factor's launch shapes (darcy256, batch 32): register-staged 64 x 64 kernel against the
LDS-DMA kernel (gemm_f64_dma.hpp) with 64 x 64 / 128 x 64 / 64 x 128 tiles and the production choice."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = [   # M, N, K, transB (BLAS flag: 1 = B stored [n][k]), tri, lower, batch, label
    (768, 768, 768, 1, 0, 1, NB, "G2 S=-C*C^T lower (no staircase)"),
    (768, 768, 256, 1, 0, 1, NB, "trailing 768 K=256 lower"),
    (512, 512, 256, 1, 0, 1, NB, "trailing 512 K=256 lower"),
    (256, 256, 256, 1, 0, 1, NB, "trailing 256 K=256 lower"),
    (512, 512, 512, 0, 4, 0, NB, "doubling 512 tri_b_lower"),
    (512, 512, 512, 0, 1, 0, NB, "doubling 512 tri_a_lower"),
    (256, 256, 256, 0, 4, 0, 2 * NB, "doubling 256 tri_b_lower"),
    (128, 128, 128, 0, 4, 0, 4 * NB, "doubling 128 tri_b_lower"),
    (64, 64, 64, 0, 4, 0, 8 * NB, "doubling 64 tri_b_lower"),
    (64, 1024, 1024, 0, 4, 0, NB, "sweep X^T k=64 (B [k][n] tri)"),
    (64, 1024, 1024, 1, 8, 0, NB, "sweep X k=64 (B [n][k] tri)"),
    (64, 768, 768, 0, 0, 0, NB, "sweep C^T k=64"),
    (1024, 1024, 1024, 1, 0, 0, NB // 2, "full 1024 NT"),
    (1024, 1024, 1024, 0, 0, 0, NB // 2, "full 1024 NN"),
]
names = {0: "regs64", 3: "dma64", 4: "dma128x64", 5: "dma64x128", 6: "auto"}
for M, N, K, tb, tri, lower, batch, label in cases:
    fl = 2.0 * M * N * K * batch * (0.5 * (1 + 64 / K) if tri else 1.0) * (0.5 * (1 + 64 / M) if lower else 1.0)
    line = f"{label:34s} b={batch:3d}"
    for big in (0, 3, 4, 5, 6):
        if (big == 4 and M % 128) or (big == 5 and (N % 128 or lower)):
            line += f" | {names[big]:9s}     -          "
            continue
        ms = C.c_double(0)
        pkg._cabi.check(lib.gmrf_test_gemm_rate(0, M, N, K, tb, tri, lower, batch, big, 20, C.byref(ms)))
        line += f" | {names[big]:9s} {ms.value*1e3:7.1f} us {fl/ms.value/1e9:5.1f}"
    print(line, flush=True)
