"""Does the fp64 GEMM rate hold over a second of continuous work (clock / power)?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
for reps in (20, 200, 2000, 4000):
    ms = C.c_double(0)
    pkg._cabi.check(lib.gmrf_test_gemm_rate(0, 1024, 1024, 1024, 1, 0, 0, 16, 1, reps, C.byref(ms)))
    print(f"reps {reps:5d}: {ms.value*1e3:8.1f} us/launch  {2*1024**3*16/ms.value/1e9:6.1f} TF/s  ({ms.value*reps:.0f} ms total)", flush=True)
