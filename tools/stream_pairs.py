"""Which torch pool streams overlap?  Two batch-32 handles on stream pairs: overlapped pairs run faster than
2 x one-stream time; pairs that share a hardware queue serialise."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
w = pkg.workloads.make("darcy256")
B = 32
pool = [torch.cuda.Stream() for _ in range(12)]

def make(st):
    with torch.cuda.stream(st):
        eng = post.HipEngine(pkg, w, batch=B, keep_l=False)
        job = post.ShardedPosterior(eng, k_samples=64, replicate_factor=True)
        job.prepare(); job.step(0)
    return (st, eng, job)

def run(jobs, steps=3):
    def body(st, job):
        with torch.cuda.stream(st):
            for s in range(steps): job.step(1 + s)
            st.synchronize()
    ths = [threading.Thread(target=body, args=(st, job)) for st, eng, job in jobs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

# one engine per pool stream (12 x 27 GB does not fit: 4 at a time)
def group(idx):
    jobs = [make(pool[i]) for i in idx]
    one = run(jobs[:1])
    print(f"streams {idx}: one handle {one*1e3:7.1f} ms/step", flush=True)
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            t = run([jobs[a], jobs[b]])
            print(f"   pair ({idx[a]:2d},{idx[b]:2d}): {t*1e3:7.1f} ms/step  = {t/one:4.2f} x one", flush=True)
    t = run(jobs)
    print(f"   all four: {t*1e3:7.1f} ms/step  {len(idx)*B*65/t:8.0f} solves/s", flush=True)
    for st, eng, job in jobs: eng.F.close()
    torch.cuda.empty_cache()

group([0, 1, 2, 3])
group([4, 5, 6, 7])
group([0, 1, 4, 5])
