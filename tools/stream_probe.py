"""Why is the first 4-stream configuration of a process slower than later ones?  Same job (4 x 32 darcy256 posteriors),
streams and handles re-created or re-used in turn."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
w = pkg.workloads.make("darcy256")
T, B = 4, 32

def make_jobs(streams):
    jobs = []
    for st in streams:
        with torch.cuda.stream(st):
            eng = post.HipEngine(pkg, w, batch=B, keep_l=False)
            job = post.ShardedPosterior(eng, k_samples=64, replicate_factor=True)
            job.prepare(); job.step(0)
        jobs.append((st, eng, job))
    torch.cuda.synchronize()
    return jobs

def run(jobs, tag, steps=4):
    def body(st, job):
        with torch.cuda.stream(st):
            for s in range(steps): job.step(1 + s)
            st.synchronize()
    ths = [threading.Thread(target=body, args=(st, job)) for st, eng, job in jobs]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
    print(f"{tag:60s} {el*1e3:8.2f} ms/step {T*B*65/el:9.1f} solves/s   streams {[hex(s.cuda_stream)[-6:] for s, _, _ in jobs]}", flush=True)

def close(jobs):
    for st, eng, job in jobs: eng.F.close()
    torch.cuda.empty_cache()

S1 = [torch.cuda.Stream() for _ in range(T)]
j = make_jobs(S1); run(j, "1: first streams, first handles"); run(j, "1b: same again"); close(j)
j = make_jobs(S1); run(j, "2: first streams, new handles"); close(j)
S2 = [torch.cuda.Stream() for _ in range(T)]
j = make_jobs(S2); run(j, "3: new streams, new handles"); close(j)
j = make_jobs(S1); run(j, "4: first streams again, new handles"); close(j)
S3 = [torch.cuda.Stream() for _ in range(T)]
j = make_jobs(S3); run(j, "5: third set of streams"); close(j)
