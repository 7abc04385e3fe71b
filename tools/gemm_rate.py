"""Device-resident GEMM timing: 64x64 kernel against the 128x128 kernel on the factor's shapes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
import sys
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cases = [
    (768, 768, 768, 1, 8, 0, NB, "G1 C=B*X^T tri_b_upper"),
    (768, 768, 768, 1, 0, 1, NB, "G2 S-=C*C^T lower"),
    (768, 768, 256, 1, 0, 1, NB, "trailing 768 K=256 lower"),
    (512, 512, 256, 1, 0, 1, NB, "trailing 512 K=256 lower"),
    (256, 256, 256, 1, 0, 1, NB, "trailing 256 K=256 lower"),
    (512, 512, 512, 0, 4, 0, NB, "doubling 512 tri_b_lower"),
    (512, 512, 512, 0, 1, 0, NB, "doubling 512 tri_a_lower"),
    (256, 256, 256, 0, 4, 0, 2 * NB, "doubling 256 tri_b_lower"),
    (128, 128, 128, 0, 4, 0, 4 * NB, "doubling 128 tri_b_lower"),
    (64, 1024, 1024, 0, 4, 0, NB, "sweep X^T k=64"),
    (64, 1024, 1024, 1, 0, 0, NB, "sweep C k=64"),
]
cases_old = [  # M, N, K, transB(BLAS flag: 1 = stored [n][k]), tri, lower, batch, label
    (768, 768, 768, 1, 8, 0, 16, "G1 C=B*X^T tri_b_upper"),
    (768, 768, 768, 1, 0, 1, 16, "G2 S-=C*C^T lower"),
    (512, 512, 512, 0, 4, 0, 32, "doubling 512 tri_b_lower"),
    (512, 512, 512, 0, 1, 0, 32, "doubling 512 tri_a_lower"),
    (256, 256, 256, 0, 4, 0, 64, "doubling 256"),
    (1024, 1024, 1024, 1, 0, 0, 16, "full 1024 NT"),
    (1024, 1024, 1024, 0, 0, 0, 16, "full 1024 NN"),
    (1024, 1024, 1024, 1, 0, 0, 1, "full 1024 NT single"),
]
for M, N, K, tb, tri, lower, batch, label in cases:
    fl = 2.0 * M * N * K * batch * (0.5 if tri else 1.0) * (0.5 * (1 + 128 / M) if lower else 1.0)
    line = f"{label:28s} b={batch:3d}"
    for big in (0, 1, 2):
        ms = C.c_double(0)
        pkg._cabi.check(lib.gmrf_test_gemm_rate(0, M, N, K, tb, tri, lower, batch, big, 20, C.byref(ms)))
        line += f" | {('small', 'big  ', 'model')[big]} {ms.value*1e3:8.1f} us {fl/ms.value/1e9:6.1f} TF/s"
    print(line, flush=True)
