"""Oracle comparison of the code path bench.py TIMES (test infrastructure: imports the oracle).

bench.py's timed region is `ShardedPosterior.step` of a `HipEngine` with a batch of independent posteriors
(distinct coefficient fields on one sparsity pattern), `keep_l = 0`, on a stream of a `StreamSet`, replayed
from captured HIP graphs: `potrf_diag128` diagonal blocks + GEMM panels and rank-256 updates (`gemm_f64_dma`),
doubling assembly of the block inverses, `spmm_bxt_tiles`, GEMM-route k = 64 sweeps, Philox `sample_batch`.
`run` drives exactly that (second step = graph replay) and compares chosen problems of the batch with the
oracle (/root/repo/oracle/bt_oracle.py, the restatement of /root/reference/src/tridiagonal_cholesky.jl:24-82
and of `mean` / `rand` / `std`, scripts/darcy/solve_darcy_gmrf-fem.jl:190-192): posterior mean, the k samples
(the device's Philox draws fetched with gmrf_bt_normals and fed to the oracle), log-determinant, exact and
RBMC(50) marginal variances of the last blocks.  Used by tests/test_gpu_parity.py (full darcy256) and by
bench.py's `parity` field (the leading blocks of the timed workload, same batch route).
"""
from __future__ import annotations

import numpy as np


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def true_inverse_diagonal(O, Q, Fo, idx, sweeps: int = 5):
    """diag(Q^-1) at `idx` in extended precision: columns Q^-1 e_i by iterative refinement of the oracle's solve
    with long-double residuals (the refinement converges to ~1e-13 relative on the BASELINE posteriors)."""
    n = Q.shape[0]
    idx = np.asarray(idx)
    E = np.zeros((n, idx.size))
    E[idx, np.arange(idx.size)] = 1.0
    Ql = Q.tocsr().astype(np.longdouble)
    El = E.astype(np.longdouble)
    x = O.ldiv(Fo, E).astype(np.longdouble)
    for _ in range(sweeps):
        x = x + O.ldiv(Fo, np.asarray(El - Ql @ x, dtype=np.float64)).astype(np.longdouble)
    return np.array([float(x[i, j]) for j, i in enumerate(idx)])


def run(pkg, post, O, Q_pattern, n_blocks, values, rhs, k_samples=64, check=(1, -1), last_blocks=8, rbmc_k=50,
        stream_index=1, device=0, true_var_samples=0, seed=0x5EED):
    """values (B, nnz) / rhs (B, n): B problems on the pattern of Q_pattern (B a multiple of 8 puts the launches
    on the XCD-grouped route of the timed job).  Returns {problem index: {metric: value}} for the problems in
    `check` plus "route" (the kernel classes the instrumented step of the same engine launched)."""
    import torch
    import scipy.sparse as sp
    B = values.shape[0]
    n = Q_pattern.shape[0]
    bs = n // n_blocks

    class W:      # what HipEngine reads of a workload
        pass
    w = W()
    w.Q, w.n, w.n_blocks, w.rhs = sp.csc_matrix(Q_pattern), n, n_blocks, rhs[0]
    ss = pkg.StreamSet(stream_index + 1, device=device)
    st = torch.cuda.ExternalStream(ss.pointers[stream_index], device=torch.device("cuda", device))
    out = {}
    with torch.cuda.stream(st):
        eng = post.HipEngine(pkg, w, device_index=device, batch=B, values=values, rhs=rhs, keep_l=False)
        job = post.ShardedPosterior(eng, k_samples=k_samples, seed=seed, replicate_factor=True)
        job.prepare()
        job.step(3)                                   # captures the factor / sweep / sample graphs
        step = 5
        mu_t, X_t = job.step(step)                    # graph replay: what the timed region runs
        st.synchronize()
        mu, X = mu_t.cpu().numpy(), X_t.cpu().numpy()
        first = step * k_samples * B                  # ShardedPosterior.step: (step * world + rank) * k * batch
        Z = eng.F.normals_batch(k_samples, seed=seed, first_id=first)            # (B, k, n)
        v_exact = eng.F.marginal_var("exact")                                    # (B, n)
        Qc = pkg.CsrMatrix(w.Q, device=device, stream=ss.pointers[stream_index])
        v_rbmc = eng.F.marginal_var("rbmc", k=rbmc_k, seed=seed + 1, Q=Qc, q_values=values)
        Zr = eng.F.normals_batch(rbmc_k, seed=seed + 1, first_id=0)
        logdets = []
        for p in range(B):
            eng.F.select_problem(p)
            logdets.append(eng.F.logdet())
        # which kernels did this route launch?  (one instrumented step of the same engine)
        eng.F.set_profiling(1)
        job.step(7)
        st.synchronize()
        stt = eng.F.stats()
        eng.F.set_profiling(0)
        out["route"] = {int(c): int(v) for c, v in enumerate(stt["kernel_launches"]) if v}
        del Qc
        eng.F.close()
    ss.close()
    tail = last_blocks * bs
    for p in check:
        p = p % B
        Qp = sp.csc_matrix(Q_pattern).copy()
        Qp.data = np.array(values[p], dtype=np.float64)
        Fo = O.tridiagonal_cholesky(Qp, n_blocks)
        mu_o = O.ldiv(Fo, rhs[p])
        r = {"mean_rel_l2": rel(mu[p], mu_o),
             "samples_rel_l2": rel(X[p].T, O.sample(Fo, mu_o, Z[p].T)),
             "logdet_rel": abs(logdets[p] - O.logdet(Fo)) / abs(O.logdet(Fo))}
        vo = O.marginal_variances_exact(Fo, last_blocks=last_blocks)
        r["var_exact_max_rel"] = float(np.max(np.abs(v_exact[p][-vo.size:] - vo) / vo))
        Xr = O.backward_solve(Fo, np.asfortranarray(Zr[p].T))
        vr_o = O.marginal_variances_rbmc(Qp, Xr)
        r["var_rbmc_max_rel"] = float(np.max(np.abs(v_rbmc[p][-tail:] - vr_o[-tail:]) / vr_o[-tail:]))
        r["var_rbmc_vs_exact_median_rel"] = float(np.median(np.abs(v_rbmc[p][-tail:] - vo[-tail:]) / vo[-tail:]))
        if true_var_samples > 0:
            # a few interior nodes of the last blocks (Dirichlet nodes have variance 1 / Q_ii to the last bit in both)
            rng = np.random.default_rng(17)
            cand = np.flatnonzero(vo > 50.0 * vo.min()) + (n - vo.size)
            idx = np.sort(rng.choice(cand, size=min(true_var_samples, cand.size), replace=False))
            vt = true_inverse_diagonal(O, Qp, Fo, idx)
            r["var_err_hip_vs_true"] = float(np.max(np.abs(v_exact[p][idx] - vt) / vt))
            r["var_err_oracle_vs_true"] = float(np.max(np.abs(vo[idx - (n - vo.size)] - vt) / vt))
        out[p] = r
    return out
