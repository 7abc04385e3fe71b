"""64x64 / 128x128 GEMM kernels against K on full 1024 x 1024 outputs of a batch: the fixed cost per output tile
(prologue, epilogue) shows as the rate lost at short K."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for tb, lab in ((1, "NT"), (0, "NN")):
    for M in (1024, 256):
        for K in (64, 128, 256, 512, 1024):
            fl = 2.0 * M * M * K * NB
            line = f"{lab} M=N={M:4d} K={K:4d} b={NB}"
            for big in (0, 1):
                ms = C.c_double(0)
                pkg._cabi.check(lib.gmrf_test_gemm_rate(0, M, M, K, tb, 0, 0, NB, big, 20, C.byref(ms)))
                line += f" | {('small', 'big  ')[big]} {ms.value*1e3:8.1f} us {fl/ms.value/1e9:6.1f} TF/s"
            print(line, flush=True)
