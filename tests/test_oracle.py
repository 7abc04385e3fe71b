"""CPU tests pinning the oracle (the reference has no golden vectors for this path):
identities, independent solvers, closed-form cases and the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import bt_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("name", ["darcy32", "burgers64x8", "elliptic32"])
def test_identities_and_independent_solvers(pkg, name):
    w = pkg.workloads.make(name)
    assert pkg.workloads.block_bandwidth_ok(w.Q, w.n_blocks)
    F = O.tridiagonal_cholesky(w.Q, w.n_blocks)
    A = w.Q.toarray()
    assert np.linalg.norm(O.reconstruct(F) - A) / np.linalg.norm(A) < 1e-13          # L L^T = A
    mu = O.ldiv(F, w.rhs)
    qn = np.abs(A).sum(axis=1).max()
    assert np.linalg.norm(A @ mu - w.rhs) / (qn * np.linalg.norm(mu) + np.linalg.norm(w.rhs)) < 1e-14
    cond = np.linalg.cond(A)
    assert rel(mu, spla.splu(w.Q.tocsc()).solve(w.rhs)) < 1e-9 * max(1.0, cond / 1e7)
    assert rel(O.backward_solve(F, O.forward_solve(F, w.rhs)), mu) == 0.0
    assert abs(O.logdet(F) - np.linalg.slogdet(A)[1]) < 1e-9 * abs(O.logdet(F))
    # factor blocks are the tri-band blocks of the dense Cholesky factor
    Lfull = np.linalg.cholesky(A)
    bs = w.block_size
    assert np.max(np.abs(F.chos[1] - Lfull[bs:2 * bs, bs:2 * bs])) < 1e-11 * np.abs(Lfull).max()
    assert np.max(np.abs(F.Cs[0] - Lfull[bs:2 * bs, :bs])) < 1e-11 * np.abs(Lfull).max()
    # matrix right-hand sides and sampling
    B = np.random.default_rng(0).standard_normal((w.n, 3))
    assert rel(O.ldiv(F, B), np.linalg.solve(A, B)) < 1e-9 * max(1.0, cond / 1e7)
    v = O.marginal_variances_exact(F)
    assert np.max(np.abs(v - np.diag(np.linalg.inv(A))) / v) < 1e-7


def test_closed_form_laplace_grid(pkg):
    nx, ny, k2 = 12, 9, 0.5
    w = pkg.workloads.laplace_kappa_grid(nx, ny, k2)
    F = O.tridiagonal_cholesky(w.Q, ny)
    v = O.marginal_variances_exact(F)
    assert np.max(np.abs(v - pkg.workloads.laplace_kappa_grid_variances(nx, ny, k2))) < 1e-13
    assert rel(O.ldiv(F, w.rhs), np.linalg.solve(w.Q.toarray(), w.rhs)) < 1e-13


def test_closed_form_ar1_chain(pkg):
    # AR(1) chain (x) I has L_i = I (last: sqrt(1 - phi^2) I) and C_i = -phi I
    phi, N, bs = 0.6, 7, 5
    w = pkg.workloads.ar1_chain_kron_identity(N, bs, phi)
    F = O.tridiagonal_cholesky(w.Q, N)
    for i in range(N):
        want = np.eye(bs) * (np.sqrt(1.0 - phi * phi) if i == N - 1 else 1.0)
        assert np.max(np.abs(F.chos[i] - want)) < 1e-15
    for c in F.Cs:
        assert np.max(np.abs(c + phi * np.eye(bs))) < 1e-15
    assert abs(O.logdet(F) - bs * np.log(1.0 - phi * phi)) < 1e-13
    # stationary variance of the last state 1/(1 - phi^2), of the first ... the same chain reversed
    v = O.marginal_variances_exact(F)
    assert np.allclose(v, np.diag(np.linalg.inv(w.Q.toarray())), rtol=1e-13)


def test_degenerate_block_sizes(pkg):
    # bs = 1: scalar tridiagonal (Thomas); N = 1: plain dense Cholesky
    n = 40
    T = sp.diags([np.full(n - 1, -1.0), np.full(n, 2.5), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsc()
    b = np.arange(n, dtype=float)
    assert rel(O.ldiv(O.tridiagonal_cholesky(T, n), b), np.linalg.solve(T.toarray(), b)) < 1e-13
    assert rel(O.ldiv(O.tridiagonal_cholesky(T, 1), b), np.linalg.solve(T.toarray(), b)) < 1e-13
    with pytest.raises(ValueError):
        O.tridiagonal_cholesky(T, 7)


def test_non_spd_reports_block(pkg):
    w = pkg.workloads.random_block_tridiagonal(4, 16, seed=1)
    ns = w.Q.tolil()
    ns[40, 40] = -50.0
    with pytest.raises(O.NotPositiveDefinite) as e:
        O.tridiagonal_cholesky(ns.tocsc(), 4)
    assert e.value.block == 3


def test_make_chunks_and_extract_blocks(pkg):
    x = np.arange(11)
    ch = O.make_chunks(x, 3)
    assert [len(c) for c in ch] == [3, 3, 5] and ch[2][-1] == 10            # remainder to the last chunk
    w = pkg.workloads.random_block_tridiagonal(4, 6, seed=2)
    coo = w.Q.tocoo()
    d, o = O.extract_blocks(coo.row + 1, coo.col + 1, coo.data, 6)
    A = w.Q.toarray()
    assert len(d) == 4 and len(o) == 3
    for i in range(4):
        assert np.array_equal(d[i].toarray(), A[6 * i:6 * i + 6, 6 * i:6 * i + 6])
        if i:
            assert np.array_equal(o[i - 1].toarray(), A[6 * i:6 * i + 6, 6 * i - 6:6 * i])
    # entries outside the band are dropped silently, like the reference
    far = sp.coo_matrix(([7.0], ([20], [1])), shape=A.shape)
    c2 = (w.Q + far).tocoo()
    d2, o2 = O.extract_blocks(c2.row + 1, c2.col + 1, c2.data, 6)
    assert all((a != b).nnz == 0 for a, b in zip(d + o, d2 + o2))
    # the vectorised host version agrees with the line-by-line restatement
    d3, o3 = pkg.extract_blocks(c2.row + 1, c2.col + 1, c2.data, 6)
    assert all((a != b).nnz == 0 for a, b in zip(d + o, d3 + o3))


def test_rbmc_estimator_is_consistent(pkg):
    w = pkg.workloads.make("darcy16")
    F = O.tridiagonal_cholesky(w.Q, w.n_blocks)
    Z = np.random.default_rng(3).standard_normal((w.n, 4000))
    X = O.backward_solve(F, Z)
    exact = O.marginal_variances_exact(F)
    assert np.max(np.abs(O.marginal_variances_rbmc(w.Q, X) - exact) / exact) < 0.05
    assert np.max(np.abs(O.marginal_variances_mc(X) - exact) / exact) < 0.15
    # sample covariance of L^-T z approaches A^-1
    cov = X @ X.T / Z.shape[1]
    Ainv = np.linalg.inv(w.Q.toarray())
    assert np.linalg.norm(cov - Ainv) / np.linalg.norm(Ainv) < 0.1


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_against_golden_fixtures(path):
    gdat = np.load(path)
    n, N = int(gdat["n"]), int(gdat["n_blocks"])
    Q = sp.csc_matrix((gdat["nzval"], gdat["rowval"], gdat["colptr"]), shape=(n, n))
    F = O.tridiagonal_cholesky(Q, N)
    scale = np.abs(gdat["chos"]).max()
    assert np.max(np.abs(np.stack(F.chos) - gdat["chos"])) < 1e-11 * scale
    if N > 1:
        assert np.max(np.abs(np.stack(F.Cs) - gdat["Cs"])) < 1e-11 * scale
    assert rel(O.ldiv(F, gdat["rhs"]), gdat["mean"]) < 1e-9
    assert rel(O.forward_solve(F, gdat["Z"]), gdat["forward"]) < 1e-10
    assert rel(O.backward_solve(F, gdat["Z"]), gdat["backward"]) < 1e-10
    assert np.max(np.abs(O.marginal_variances_exact(F) - gdat["var"]) / gdat["var"]) < 1e-8
    assert abs(O.logdet(F) - float(gdat["logdet"])) < 1e-10 * abs(float(gdat["logdet"]))


def test_metrics_match_reference_definitions():
    p, s = np.array([1.0, 2.0, 4.0]), np.array([1.0, 1.0, 2.0])
    assert O.rmse(p, s) == pytest.approx(np.sqrt(5.0 / 3.0))
    assert O.max_err(p, s) == 2.0
    assert O.rel_err(p, s) == pytest.approx(np.sqrt(5.0) / np.sqrt(6.0))


def test_c_restatement_agrees_with_numpy_oracle(pkg):
    """oracle/bt_oracle.c (dependency-free) against the LAPACK-backed oracle on a small case."""
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "libbt_oracle.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libbt_oracle.so not built (run __graft_entry__.build())")
    lib = C.CDLL(path)
    w = pkg.workloads.random_block_tridiagonal(5, 12, seed=6)
    N, bs = 5, 12
    A = w.Q.toarray()
    D = np.ascontiguousarray(np.stack([A[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] for i in range(N)]))
    B = np.ascontiguousarray(np.stack([A[(i + 1) * bs:(i + 2) * bs, i * bs:(i + 1) * bs] for i in range(N - 1)]))
    Ld = np.zeros_like(D); Cs = np.zeros_like(B)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.bt_factor_dense.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bt_solve_dense.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    assert lib.bt_factor_dense(N, bs, p(D), p(B), p(Ld), p(Cs)) == 0
    F = O.tridiagonal_cholesky(w.Q, N)
    assert np.max(np.abs(Ld - np.stack(F.chos))) < 1e-12 and np.max(np.abs(Cs - np.stack(F.Cs))) < 1e-12
    for mode, ref in [(0, O.ldiv(F, w.rhs)), (1, O.forward_solve(F, w.rhs)), (2, O.backward_solve(F, w.rhs))]:
        y = np.zeros(N * bs)
        lib.bt_solve_dense(N, bs, p(Ld), p(Cs), p(np.ascontiguousarray(w.rhs)), p(y), mode)
        assert rel(y, ref) < 1e-12
    D[2, 3, 3] = -5.0
    assert lib.bt_factor_dense(N, bs, p(D), p(B), p(Ld), p(Cs)) == 3


def test_gauss_newton_step_restatement(pkg):
    # scripts/solve_burger.jl:143-149: A = Q + noise J'J, rhs = Qx_prior + noise J'(J x + obs_diff)
    gn = pkg.workloads.burgers_gauss_newton(32, 6)
    x = gn["x_prior"]
    J = gn["jacobian"](x)
    r = gn["residual"](x)
    xn = O.gn_step(gn["Q"], J, gn["Qx_prior"], x, -r, gn["noise"], gn["n_blocks"])
    A = O.assemble_posterior(gn["Q"], J, gn["noise"])
    rhs = O.gn_rhs(gn["Qx_prior"], J, x, -r, gn["noise"])
    assert np.linalg.norm(A @ xn - rhs) / np.linalg.norm(rhs) < 1e-12
    Ad = gn["Q"].toarray() + gn["noise"] * (J.toarray().T @ J.toarray())
    assert np.max(np.abs(A.toarray() - Ad)) / np.max(np.abs(Ad)) < 1e-15
    # the step minimises the linearised objective: its gradient vanishes at xn
    grad = gn["Q"] @ xn - gn["Qx_prior"] + gn["noise"] * (J.T @ (J @ (xn - x) + r))
    assert np.linalg.norm(grad) / np.linalg.norm(rhs) < 1e-12


def test_conditioning_restatement(pkg):
    # posterior precision Q + q A'A and mean Q_post^-1 (Q mu + q A' y), against dense algebra
    Q0, obs, N = pkg.workloads.darcy_conditioning(16)
    A, y = obs[0]
    mu0 = np.linspace(0.0, 1.0, Q0.shape[0])
    Qp, F, mu = O.condition_on_observations(Q0, mu0, A, 1e4, y, N)
    Ad, Qd = A.toarray(), Q0.toarray()
    Pd = Qd + 1e4 * Ad.T @ Ad
    assert np.max(np.abs(Qp.toarray() - Pd)) / np.max(np.abs(Pd)) < 1e-15
    want = np.linalg.solve(Pd, Qd @ mu0 + 1e4 * Ad.T @ y)
    assert rel(mu, want) < 1e-9
    # equivalent form mu + Q_post^-1 A' q (y - A mu)   
    alt = mu0 + np.linalg.solve(Pd, 1e4 * Ad.T @ (y - Ad @ mu0))
    assert rel(mu, alt) < 1e-9
