"""Shader clock beside every rate: a bounded probe kernel (gmrf_test_clock_probe_*) stamps s_memtime /
s_memrealtime on a stream of its own while the load under test runs on others; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
Loads: nothing; the fp64 MFMA-only loop of gmrf_test_mfma_f64_rate; the 1024^3 x 16 product on the 128 x 128 kernel and on the
LDS-DMA kernel for 20 ... 4000 back-to-back launches (short burst -> sustained); the factor's G2 / rank-256 shapes; one
handle's darcy256 factor step (batch from argv)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import __graft_entry__ as g

pkg = g.load_package(); lib = pkg._cabi.load()


def probed(fn, n=4000, sleeps=2):
    """Run fn() under the probe; returns (fn's result, clock samples GHz, their start times ms)."""
    h = C.c_void_p()
    pkg._cabi.check(lib.gmrf_test_clock_probe_start(0, n, sleeps, C.byref(h)))
    t0 = time.perf_counter()
    res = fn()
    wall = time.perf_counter() - t0
    ghz = np.zeros(n - 1); tms = np.zeros(n - 1)
    pkg._cabi.check(lib.gmrf_test_clock_probe_finish(h, pkg._cabi.ptr(ghz), pkg._cabi.ptr(tms)))
    return res, ghz, tms, wall


def during(ghz, tms, t_lo, t_hi):
    m = (tms >= t_lo) & (tms <= t_hi)
    v = ghz[m]
    return (float(np.median(v)), float(v.min()), float(v.max()), int(m.sum())) if m.any() else (float("nan"),) * 3 + (0,)


_, ghz, tms, _ = probed(lambda: time.sleep(0.02))
print(f"idle: clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}) over {tms[-1]:.1f} ms", flush=True)


def mfma_loop():
    tf = C.c_double(0)
    pkg._cabi.check(lib.gmrf_test_mfma_f64_rate(0, C.byref(tf)))
    return tf.value
tf, ghz, tms, wall = probed(mfma_loop, n=6000)
busy = ghz[ghz < np.median(ghz[:20]) * 0.995] if len(ghz) > 20 else ghz
print(f"fp64 MFMA-only loop (gmrf_test_mfma_f64_rate): {tf:.1f} TF/s; clock during the probe window: median {np.median(ghz):.3f}, "
      f"min {ghz.min():.3f}, lowest decile {np.percentile(ghz, 10):.3f} GHz; at the lowest-decile clock the 256 x 4 matrix pipes give "
      f"{256 * 4 * 32 * np.percentile(ghz, 10) / 1e3:.1f} TF/s", flush=True)

for label, (M, N, K, tb, tri, lower, batch, big) in {
        "1024^3 x 16, 128 x 128 kernel": (1024, 1024, 1024, 1, 0, 0, 16, 1),
        "1024^3 x 16, LDS-DMA 128 x 64": (1024, 1024, 1024, 1, 0, 0, 16, 4),
        "1024^3 x 16, LDS-DMA 64 x 64": (1024, 1024, 1024, 1, 0, 0, 16, 3),
        "rank-256 update 768^2 lower x 64 (production choice)": (768, 768, 256, 1, 0, 1, 64, 6),
        "G2 768^3 lower x 64 (production choice, no staircase)": (768, 768, 768, 1, 0, 1, 64, 6)}.items():
    fl = 2.0 * M * N * K * batch * (0.5 * (1 + 64 / M) if lower else 1.0)
    for reps in (20, 400, 4000):
        def run(reps=reps):
            ms = C.c_double(0)
            pkg._cabi.check(lib.gmrf_test_gemm_rate(0, M, N, K, tb, tri, lower, batch, big, reps, C.byref(ms)))
            return ms.value
        ms, ghz, tms, wall = probed(run, n=min(60000, max(2000, int(reps * 0.2 * 1e3 / 7) + 2000)))
        lo = np.percentile(ghz, 10)
        print(f"{label}: {reps:5d} launches, {ms * 1e3:7.1f} us each = {fl / ms / 1e9:5.1f} TF/s; clock lowest decile {lo:.3f} GHz "
              f"(median {np.median(ghz):.3f}) -> peak at that clock {256 * 4 * 32 * lo / 1e3:.1f} TF/s, fraction of it {fl / ms / 1e9 / (256 * 4 * 32 * lo / 1e3):.3f}", flush=True)

# one handle's factor + mean + samples step at the bench batch, under the probe
import torch
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = pkg.workloads.make("darcy256")
F = pkg.TridiagonalCholeskyFactor(batch=batch); F.set_keep_l(False)
vals = np.tile(w.Q.data, (batch, 1))
F.factor(w.Q, w.n_blocks, values=vals)
nz = torch.from_numpy(vals).cuda()
for _ in range(2):
    F.refactor(nz)
torch.cuda.synchronize()
def steps():
    t0 = time.perf_counter()
    for _ in range(6):
        F.refactor(nz)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 6
t, ghz, tms, wall = probed(steps, n=60000, sleeps=3)
m = tms < wall * 1e3
print(f"darcy256 factor, one handle x batch {batch}: {t * 1e3:.1f} ms per batch factor; clock during: median {np.median(ghz[m]):.3f}, lowest decile "
      f"{np.percentile(ghz[m], 10):.3f}, min {ghz[m].min():.3f} GHz", flush=True)
