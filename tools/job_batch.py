"""A few batched multi-stream steps for profiling runs: job_batch.py <config> <streams> <batch> <steps>"""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
from importlib import import_module
pkg = g.load_package(); post = import_module(g.PKG_NAME + ".posterior")
name, T, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = pkg.workloads.make(name)
jobs = []
for t in range(T):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng = post.HipEngine(pkg, w, batch=B)
        job = post.ShardedPosterior(eng, k_samples=64, replicate_factor=True)
        job.prepare()
    jobs.append((st, eng, job))
torch.cuda.synchronize()
def run(st, job):
    with torch.cuda.stream(st):
        for s in range(steps): job.step(1 + s)
        st.synchronize()
ths = [threading.Thread(target=run, args=(st, job)) for st, eng, job in jobs]
for th in ths: th.start()
for th in ths: th.join()
torch.cuda.synchronize()
print("done")
