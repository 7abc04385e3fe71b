"""Throughput with several independent batched handles driven by host threads on separate streams."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
name = sys.argv[1] if len(sys.argv) > 1 else "darcy256"
w = pkg.workloads.make(name)
for spec in (sys.argv[2] if len(sys.argv) > 2 else "1x8,2x4,2x8,4x4").split(","):
    T, B = [int(x) for x in spec.split("x")]
    jobs = []
    for t in range(T):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            eng = post.HipEngine(pkg, w, batch=B, keep_l=os.environ.get('GMRF_KEEP_L', '1') != '0')
            job = post.ShardedPosterior(eng, k_samples=64, replicate_factor=True)
            if os.environ.get('GMRF_EAGER_FLAGS'): eng.F.set_eager(int(os.environ['GMRF_EAGER_FLAGS']))
            job.prepare()
            job.step(0)
        jobs.append((st, eng, job))
    torch.cuda.synchronize()
    steps = 4
    def run(st, job):
        with torch.cuda.stream(st):
            for s in range(steps): job.step(1 + s)
            st.synchronize()
    ths = [threading.Thread(target=run, args=(st, job)) for st, eng, job in jobs]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
    print(f"{T} streams x B={B}: {el*1e3:8.2f} ms/step  {T*B*65/el:9.1f} solves/s", flush=True)
    for st, eng, job in jobs: eng.F.close()
    del jobs; torch.cuda.empty_cache()
