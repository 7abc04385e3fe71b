"""Throughput of the darcy job as a function of the batch size (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
name = sys.argv[1] if len(sys.argv) > 1 else "darcy256"
w = pkg.workloads.make(name)
for B in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]:
    eng = post.HipEngine(pkg, w, batch=B)
    job = post.ShardedPosterior(eng, k_samples=64, replicate_factor=True)
    job.prepare()
    for s in range(2): job.step(s)
    torch.cuda.synchronize(); t = time.perf_counter()
    steps = 4
    for s in range(steps): job.step(2 + s)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / steps
    st = eng.F.stats()
    print(f"B={B}: {el*1e3:8.2f} ms/step  {B*65/el:9.1f} solves/s   (factor {st['factor_ms']:.2f} ms, solve {st['solve_ms']:.2f}, sample {st['sample_ms']:.2f})", flush=True)
    eng.F.close(); del eng, job
    torch.cuda.empty_cache()
