#!/usr/bin/env python3
"""bench.py -- GMRF posterior solves/sec (mean + samples) on the 256x256 Darcy posterior.

One step = one pass of the hot path over one batch of synthetic input: B independent 256^2
Darcy posteriors per GPU (same mesh and sparsity pattern, B coefficient fields -- the per-problem
loop of scripts/darcy/solve_darcy_gmrf-fem.jl:176-198), each one: numeric factorisation of
Q_post (values already in HBM; the pattern is analysed once before the timed region, like the
reference re-uses its permutation, :166-174), the posterior mean (forward + backward sweep) and
64 posterior samples (backward sweep of 64 right-hand sides).  The B problems advance in lock
step (problem = one more grid dimension of every kernel): the factorisation is a chain of
latency-bound launches that fills <= 121 of the 256 CUs, a batch fills the rest.
value = n_gpus * B * (1 + 64) * steps / time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1, default (--mode problems): every rank handles its own batch of problems, no data-path
collective (weak scaling).  --mode shared-factor: ONE problem; rank 0 factors and broadcasts the
factor block-range by block-range over RCCL while factoring the next range; every rank draws
its own 64 samples of the shared posterior.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X datasheet fp64 matrix peak; 77.7 measured (tools/mb2.hip)
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3-6.5 TB/s achievable)

KERNEL_CLASSES = {      # one class per kernel symbol (include/gmrf_hip.h, gmrf_stats)
    0: ("gemm_f64_mfma<false,false>", "mfma"),   # 64 x 64 tile GEMM, B stored [n][k]
    11: ("gemm_f64_mfma<false,true>", "mfma"),   # 64 x 64 tile GEMM, B stored [k][n]
    13: ("gemm_f64_ll", "mfma"),                 # 32 x 32 tile GEMM of launches with <= 128 tiles of 64 x 64
    6: ("gemm_f64_big<false>", "mfma"),     # 128 x 128 tile GEMM, B stored [n][k]
    7: ("gemm_f64_big<true>", "mfma"),      # 128 x 128 tile GEMM, B stored [k][n]
    1: ("potrf_step<false>", "mfma"),       # tile Cholesky + inverse (latency-bound, B workgroups)
    8: ("potrf_panel", "mfma"),
    9: ("potrf_update", "mfma"),
    2: ("sweep_mm", "mfma"),
    3: ("sweep_gemv", "hbm"),
    10: ("spmm_bxt", "hbm"),
    4: ("csr_spmm", "hbm"),
}


def cpu_sparse_direct(w, k_samples: int):
    """Secondary CPU comparator: a general sparse direct solver on the same posterior
    precision -- what the reference's scripts really call (CHOLMOD there; SuperLU via SciPy here,
    CHOLMOD bindings are not installed).  1 factorisation + (1 + k) solves, single thread."""
    import numpy as np
    import scipy.sparse.linalg as spla
    Z = np.random.default_rng(0).standard_normal((w.n, k_samples))
    t0 = time.perf_counter()
    lu = spla.splu(w.Q.tocsc(), permc_spec="MMD_AT_PLUS_A", options={"SymmetricMode": True})
    t1 = time.perf_counter()
    lu.solve(w.rhs)
    lu.solve(Z)
    t2 = time.perf_counter()
    return {"value": (1 + k_samples) / (t2 - t0), "unit": "solves/s", "cores": 1,
            "kind": "scipy.sparse.linalg.splu (SuperLU, MMD_AT_PLUS_A, symmetric mode)",
            "sample": f"1 factorisation {t1 - t0:.2f} s + {1 + k_samples} solves {t2 - t1:.2f} s"}


def cpu_baseline(w, k_samples: int):
    """The oracle (LAPACK-backed NumPy/SciPy restatement of the reference algorithm) timed on
    this box's host cores on ONE full job of the same workload: factor + mean + k samples."""
    import numpy as np
    from oracle import bt_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    Z = np.random.default_rng(0).standard_normal((w.n, k_samples))
    t0 = time.perf_counter()
    F = O.tridiagonal_cholesky(w.Q, w.n_blocks)
    t1 = time.perf_counter()
    mu = O.ldiv(F, w.rhs)
    X = O.sample(F, mu, Z)
    t2 = time.perf_counter()
    total = t2 - t0
    return {"value": (1 + k_samples) / total, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": f"1 full job of {w.name} (n={w.n}, {w.n_blocks} blocks of {w.block_size}): factor "
                      f"{t1 - t0:.2f} s + mean and {k_samples} samples {t2 - t1:.2f} s, SciPy/OpenBLAS",
            "factor_s": t1 - t0, "sweeps_s": t2 - t1}, (mu, X, Z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="darcy256")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32, help="independent problems per handle and step")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batched handles per GPU, each on its own HIP stream and host thread")
    ap.add_argument("--mode", choices=["problems", "shared-factor"], default="problems")
    ap.add_argument("--group", type=int, default=8, help="blocks per broadcast range (shared-factor)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-problem", action="store_true",
                    help="skip the batch-1 latency probe (profile runs: keeps one launch shape per kernel)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # Rehearsal of the N > 1 control flow on a one-GPU box: GMRF_BENCH_BACKEND=gloo with
    # GMRF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 (gloo moves CUDA tensors through the host).
    backend = os.environ.get("GMRF_BENCH_BACKEND", "nccl")
    if os.environ.get("GMRF_BENCH_ONE_DEVICE") == "1" and backend == "gloo":
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)

    import numpy as np
    w = pkg.workloads.make(args.config)
    from importlib import import_module
    post = import_module(g.PKG_NAME + ".posterior")
    import threading
    shared = (args.mode == "shared-factor" and world > 1)
    batch = 1 if shared else args.batch
    n_streams = 1 if shared else max(1, args.streams)
    # coefficient fields on the same mesh: same pattern, different values
    total_problems = batch * n_streams
    vals, rhss = [w.Q.data], [w.rhs]
    for p in range(1, min(total_problems, 8)):
        if args.config.startswith("darcy"):
            wp = pkg.workloads.darcy(int(args.config[5:]), seed=523802340 + 1000 * rank + p)
            same = wp.Q.nnz == w.Q.nnz and np.array_equal(wp.Q.indices, w.Q.indices)
        else:
            same = False
        if same:
            vals.append(wp.Q.data); rhss.append(wp.rhs)
        else:
            vals.append(w.Q.data * (1.0 + 0.01 * p)); rhss.append(w.rhs)
    jobs = []
    for t in range(n_streams):
        st_t = torch.cuda.current_stream() if n_streams == 1 else torch.cuda.Stream()
        idx = [(t * batch + p) % len(vals) for p in range(batch)]
        with torch.cuda.stream(st_t):
            e_t = post.HipEngine(pkg, w, device_index=local, batch=batch, values=np.stack([vals[i] for i in idx]),
                                 rhs=np.stack([rhss[i] for i in idx]))
            j_t = post.ShardedPosterior(e_t, dist=dist if shared else None, rank=rank, world=world if shared else 1,
                                        k_samples=args.samples, group=args.group, replicate_factor=not shared)
            j_t.prepare()
        jobs.append((st_t, e_t, j_t))
    eng, job = jobs[0][1], jobs[0][2]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(first, count):
        """`count` steps on every handle; with several handles each one is driven by its own host
        thread on its own stream (the C ABI releases the GIL) so that their launch chains overlap."""
        def worker(st_t, j_t):
            with torch.cuda.stream(st_t):
                for s in range(count):
                    j_t.step(first + s)
                st_t.synchronize()
        if len(jobs) == 1:
            worker(jobs[0][0], jobs[0][2])
            return
        ths = [threading.Thread(target=worker, args=(st_t, j_t)) for st_t, _, j_t in jobs]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    run_steps(0, args.warmup)
    sync()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        per_step = job.solves_per_step() if shared else world * n_streams * batch * (1 + args.samples)
        solves = per_step * args.steps
        st = eng.F.stats()
        out = {
            "metric": "GMRF posterior solves/sec (mean+samples), 256^2 Darcy",
            "value": solves / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{w.name}: n={w.n}, {w.n_blocks} blocks x {w.block_size}, nnz={w.Q.nnz}; "
                                   f"{n_streams * batch} independent posterior(s) per GPU and step ({n_streams} stream(s) x "
                                   f"batch {batch}), each factor + mean + {args.samples} samples",
                       "problems_per_gpu_per_step": n_streams * batch, "streams": n_streams, "batch": batch,
                       "samples_per_problem": args.samples,
                       "sharding": "independent problems per rank, no data-path collective" if job.replicate
                       else f"one shared factor, rccl broadcast in {args.group}-block ranges, samples sharded"},
        }
    if rank == 0:
        free_b, total_b = torch.cuda.mem_get_info(local)
        out["hbm_used_gb"] = round((total_b - free_b) / 1e9, 1)
    # ---- per-kernel roofline + parity + CPU baseline: rank 0, outside the timed region
    if rank == 0 and world == 1:
        import numpy as np
        eng.F.set_profiling(1)
        with torch.cuda.stream(jobs[0][0]):
            job.step(10_000)
        torch.cuda.synchronize()
        st = eng.F.stats()
        eng.F.set_profiling(0)
        ms, work, cnt = st["kernel_ms"], st["kernel_work"], st["kernel_launches"]
        # the kernel with the largest time; classes within 15 % of it count as tied and the first in
        # KERNEL_CLASSES order is named, so that the roofline object does not flip between the two
        # operand layouts of the same GEMM kernel from run to run
        top = max(ms[c] for c in KERNEL_CLASSES)
        dom = next(c for c in KERNEL_CLASSES if ms[c] >= 0.85 * top)
        name, bound = KERNEL_CLASSES[dom]
        if bound == "mfma":
            achieved = work[dom] / (ms[dom] * 1e-3) / 1e12
            peak, unit = PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
        else:
            achieved = work[dom] / (ms[dom] * 1e-3) / 1e9
            peak, unit = PEAK_HBM_GBPS, "GB/s"
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of this workload
        # (profiles/README.md); null when the profile has no row for it
        traffic, traffic_src = None, None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            key = next((k for k in prof["kernels"] if k.replace(" ", "").startswith(name.replace(" ", "").rstrip(">"))), None)
            if key and w.name == "darcy256" and batch == 32:
                traffic = prof["kernels"][key]["read_bytes_per_launch"] + prof["kernels"][key]["write_bytes_per_launch"]
                traffic_src = "profiles/r01_hbm_traffic.json: " + key
        except Exception:
            pass
        out["roofline"] = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                           "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src, "kernel": name,
                           "launches_per_step": int(cnt[dom]), "avg_launch_us": 1e3 * ms[dom] / max(cnt[dom], 1)}
        # the HBM-bound leg of the path (north_star: sweep HBM GB/s against the 8 TB/s roofline): the k = 1
        # GEMV sweep kernels of the same instrumented step; bytes = the reference's dense L / C blocks
        if ms[3] > 0:
            g3 = work[3] / (ms[3] * 1e-3) / 1e9
            out["roofline_sweep"] = {"bound": "hbm", "achieved": g3, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                     "frac": g3 / PEAK_HBM_GBPS, "kernel": "sweep_gemv_n / sweep_gemv_t",
                                     "launches_per_step": int(cnt[3]), "avg_launch_us": 1e3 * ms[3] / max(cnt[3], 1),
                                     "note": "algorithmic bytes of the dense blocks (SURVEY 8d); the kernels read only the non-zero part of C"}
        out["kernels"] = {KERNEL_CLASSES[c][0]: {"ms_per_step": ms[c], "launches": int(cnt[c]),
                                                  ("tflops" if KERNEL_CLASSES[c][1] == "mfma" else "gbps"):
                                                  (work[c] / (ms[c] * 1e-3) / (1e12 if KERNEL_CLASSES[c][1] == "mfma" else 1e9)) if ms[c] > 0 else 0.0}
                          for c in KERNEL_CLASSES}
        # phase times of the un-instrumented path (whole batch)
        eng.F.refactor(eng.nz)
        mu = eng.mean()
        s1 = eng.F.stats()
        Xs = eng.sample(args.samples, mu, 0x5EED, 0)
        s2 = eng.F.stats()
        out["phases_ms"] = {"factor": s1["factor_ms"], "mean_2_sweeps": s1["solve_ms"], "samples_1_sweep": s2["sample_ms"]}
        out["factor_tflops"] = st["factor_flops"] / (s1["factor_ms"] * 1e-3) / 1e12
        out["sweep_k1_gbps"] = s1["sweep_bytes"] / (0.5 * s1["solve_ms"] * 1e-3) / 1e9
        if not args.no_single_problem:
            # latency of ONE problem (batch 1) on the same GPU, for reference
            F1 = pkg.TridiagonalCholeskyFactor(device=local, stream=eng.stream.cuda_stream).factor(w.Q, w.n_blocks)
            nz1 = eng.nz[0].contiguous(); rhs1 = eng.rhs[0, 0].contiguous()
            for _ in range(2):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                F1.refactor(nz1); mu1 = pkg.ldiv(F1, rhs1); X1 = F1.sample(args.samples, mean=mu1, seed=1, like=rhs1)
                torch.cuda.synchronize(); lat = time.perf_counter() - t1
            out["single_problem"] = {"latency_ms": 1e3 * lat, "solves_per_s": (1 + args.samples) / lat}
            if not args.no_cpu_baseline:
                base, (mu_o, X_o, Z) = cpu_baseline(w, args.samples)
                out["cpu_baseline"] = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
                out["speedup_vs_cpu"] = out["value"] / base["value"]
                out["cpu_sparse_direct"] = cpu_sparse_direct(w, args.samples)
                mu_h = mu1.cpu().numpy()
                Xh = F1.sample(args.samples, mean=mu_o, z=Z)
                cond_eps = 3.4e9 * 2.2e-16 if w.name == "darcy256" else None
                out["parity"] = {"mean_rel_l2": float(np.linalg.norm(mu_h - mu_o) / np.linalg.norm(mu_o)),
                                 "samples_rel_l2": float(np.linalg.norm(Xh - X_o) / np.linalg.norm(X_o)),
                                 "bound_0.1_cond_eps": 0.1 * cond_eps if cond_eps else None}
            F1.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
