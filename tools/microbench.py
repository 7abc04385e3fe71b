import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); lib = pkg._cabi.load()
out = np.zeros(16)
for rep in range(2):
    pkg._cabi.check(lib.gmrf_test_microbench(0, pkg._cabi.ptr(out), 16))
    print(f"rep {rep}")
    print(f" MFMA f64 1 wave/SIMD 4 acc : {out[0]:.1f} TF/s @ {out[1]:.2f} GHz")
    print(f" MFMA f64 2 waves/SIMD 4 acc: {out[2]:.1f} TF/s @ {out[3]:.2f} GHz")
    print(f" MFMA f64 1 wave/SIMD 1 acc : {out[4]:.1f} TF/s @ {out[5]:.2f} GHz")
    print(f" VALU f64 fma 2 waves/SIMD  : {out[6]:.1f} TF/s @ {out[7]:.2f} GHz")
    print(f" single WG MFMA loop        : {out[8]*1e3:.1f} GF/s @ {out[9]:.2f} GHz, {out[10]:.1f} cycles/MFMA")
    print(f" empty kernel cadence eager : {out[11]:.2f} us ; graph: {out[12]:.2f} us")
    print(f" light kernel after 200 ms idle: clock {out[13]:.2f} GHz, duration {out[14]:.1f} us")
