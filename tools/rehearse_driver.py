"""Multi-rank rehearsal on a ONE-GPU box.  This process never touches the GPU: it starts, one after the
other, (a) two ranks on cuda:0 over gloo (torch.distributed.run -> tools/rehearse_rank.py) and (b) a
single-rank reference + RCCL world-of-one run (tools/rehearse_single.py), then compares their outputs
with NumPy and writes result.json into argv[1].  tests/conftest.py runs it at session start, before
the pytest process initialises the GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outdir = sys.argv[1]
os.makedirs(outdir, exist_ok=True)
res = {"ok": False, "checks": {}}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(cmd, log):
    with open(os.path.join(outdir, log), "w") as f:
        return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT).returncode


try:
    env_py = sys.executable
    rc = run([env_py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(free_port()), os.path.join(ROOT, "tools", "rehearse_rank.py"), outdir], "two_ranks.log")
    res["two_ranks_rc"] = rc
    rc1 = run([env_py, os.path.join(ROOT, "tools", "rehearse_single.py"), outdir], "single.log")
    res["single_rc"] = rc1
    if rc == 0 and rc1 == 0:
        r0, r1, s = (np.load(os.path.join(outdir, f)) for f in ("r0.npz", "r1.npz", "single.npz"))
        c = res["checks"]
        # both ranks hold the same factor: bitwise equal means, equal to the one-rank run
        c["mean_equal_across_ranks"] = bool(np.array_equal(r0["mu1"], r1["mu1"]) and np.array_equal(r0["mu0"], r1["mu0"]))
        c["mean_equals_single_rank"] = bool(np.array_equal(r0["mu1"], s["mu"]))
        # every rank drew its own sample ids; a single rank drawing the same ids gives the same samples
        c["samples_rank_invariant"] = bool(all(np.array_equal(r[f"X{st}"], s[f"X{st}_{k}"])
                                               for st in (0, 1) for k, r in ((0, r0), (1, r1))))
        c["allreduce_equal"] = bool(np.array_equal(r0["acc"], r1["acc"]))
        c["layout_travelled"] = bool(np.array_equal(r0["layout"], r1["layout"]) and r0["layout"].size > 3)
        c["solves_per_step"] = int(r0["solves"]) == 2 * (1 + 6 * 2)
        # the packed transport image (lower-triangular tiles of Linv + C windows + log-determinant parts) is what moved
        c["logdet_travelled"] = bool(np.array_equal(r0["logdet"], r1["logdet"]) and np.all(np.isfinite(r1["logdet"])))
        c["packed_bytes_below_raw"] = bool(0 < float(r0["bytes"][0]) < 2 * 2 * 8.0 * (16 * 256 * 256 + 15 * 256 * 256))
        # round 4, the all-gather form: each rank factored one of the two problems; same means / samples / log-determinants as the
        # broadcast form (the factor of a problem does not depend on who made it), samples rank-invariant
        # (to rounding, not bitwise: a share of ONE problem takes the one-problem kernels, the broadcast form's root factored the
        #  batch of two; both ranks of the all-gather form hold bitwise the same gathered factors)
        close = lambda a, b: bool(np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b))
        c["allgather_equal_across_ranks"] = bool(np.array_equal(r0["gmu1"], r1["gmu1"]) and np.array_equal(r0["glogdet"], r1["glogdet"]))
        c["allgather_mean_equals_broadcast"] = close(r0["gmu1"], r0["mu1"])
        c["allgather_samples_rank_invariant"] = bool(all(close(r[f"gX{st}"], s[f"X{st}_{k}"])
                                                         for st in (0, 1) for k, r in ((0, r0), (1, r1))))
        c["allgather_logdet"] = bool(np.allclose(r0["glogdet"], r0["logdet"], rtol=1e-12, atol=0))
        c["allgather_moved_half"] = bool(0 < float(r0["gbytes"][0]) < float(r0["bytes"][0]))
        c["allgather_solves_per_step"] = int(r0["gsolves"]) == 2 * (1 + 6 * 2)
        # RCCL through the C ABI, world of one: same job, rank 0 of a world of one at step 1 draws ids (1*1+0)*6*2 = 12.. = step 0 / rank 1 above
        c["cabi_mean"] = bool(np.array_equal(s["mu_cabi"], s["mu"]))
        c["cabi_samples"] = bool(np.array_equal(s["X_cabi"], s["X0_1"]))
        c["cabi_allreduce"] = bool(np.all(s["acc_cabi"] == 1.0))
        res["ok"] = all(c.values())
except Exception as e:        # noqa: BLE001 -- the test reports whatever went wrong
    res["error"] = repr(e)
for name in ("two_ranks.log", "single.log"):
    pth = os.path.join(outdir, name)
    if os.path.exists(pth):
        res[name] = open(pth).read()[-3000:]
json.dump(res, open(os.path.join(outdir, "result.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.endswith(".log")}))
sys.exit(0 if res["ok"] else 1)
