"""ORACLE (test infrastructure, never shipped, never on the measured GPU path).

CPU restatement, in NumPy/SciPy (LAPACK-backed like Julia's LinearAlgebra), of the
block-tridiagonal Cholesky path of timweiland/DiffEqGMRFs.jl:

    /root/reference/src/tridiagonal_cholesky.jl      (factor :65-82, sweeps :24-33, :43-52,
                                                      chunking :11-14, ldiv :54-63)
    /root/reference/scripts/solve_burger.jl:182-254  (extract_blocks)

PARITY UNPINNED: the reference cannot be executed here (no Julia toolchain, the GMRF
dependency is un-vendored and unpinned) and its test-suite holds no golden vector or
known-answer test for this path (test/runtests.jl:5-9 is Aqua only).  The oracle is
therefore pinned by (i) mathematical identities, (ii) independent solvers
(scipy dense Cholesky / SuperLU) and (iii) closed-form cases -- see tests/test_oracle.py.

The three defects of the reference's solve half (`L.u`, assigning
Vectors into a Vector{SubArray}, returning chunks instead of a flat vector) are corrected
to the evident intent  y = L^-T L^-1 b ; everything else follows the source line by line.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


@dataclass
class TridiagonalCholeskyFactor:
    """src/tridiagonal_cholesky.jl:5-9.  `N` is the TOTAL size n (not the block count);
    `chos[i]` is the lower Cholesky factor L_i of the i-th Schur complement (Julia stores
    U = L_i^T; same numbers), `Cs[i]` = L_{i+1,i} = B_{i+1} L_i^-T."""

    N: int
    chos: List[np.ndarray]
    Cs: List[np.ndarray]

    @property
    def n_blocks(self) -> int:
        return len(self.chos)

    @property
    def block_size(self) -> int:
        return self.chos[0].shape[0]


def make_chunks(X: np.ndarray, n: int):
    """src/tridiagonal_cholesky.jl:11-14: n contiguous views of length len//n, the last one
    absorbing the remainder."""
    c = X.shape[0] // n
    return [X[c * k:(X.shape[0] if k == n - 1 else c * k + c)] for k in range(n)]


def _mm(A: np.ndarray, B: np.ndarray, trans_a: bool = False) -> np.ndarray:
    """op(A) @ B through SciPy's BLAS (dgemv / dgemm).  NumPy and SciPy ship separate OpenBLAS
    thread pools; alternating between them makes the pools fight for the cores and slows the
    CPU baseline 3-4x, so every dense product of the oracle goes through the SciPy one."""
    if B.ndim == 1:
        return sla.blas.dgemv(1.0, A, B, trans=1 if trans_a else 0)
    return sla.blas.dgemm(1.0, A, B, trans_a=1 if trans_a else 0)


def _chol_forward(L: np.ndarray, b: np.ndarray) -> np.ndarray:
    """forward_solve(::Cholesky, b) = L.L \\ b  (:35-37)."""
    return sla.solve_triangular(L, b, lower=True, check_finite=False)


def _chol_backward(L: np.ndarray, b: np.ndarray) -> np.ndarray:
    """backward_solve(::Cholesky, b) = L.U \\ b  (:16-18, `L.u` defect corrected)."""
    return sla.solve_triangular(L, b, lower=True, trans="T", check_finite=False)


class NotPositiveDefinite(Exception):
    """Julia's PosDefException; `.block` is the 1-based block index that failed."""

    def __init__(self, block: int):
        super().__init__(f"block {block} is not positive definite")
        self.block = block


def tridiagonal_cholesky(A, N_blocks: int) -> TridiagonalCholeskyFactor:
    """src/tridiagonal_cholesky.jl:65-82.  Reads only the lower blocks (i,i) and (i,i-1)."""
    A = sp.csc_matrix(A)
    n = A.shape[0]
    bs = n // N_blocks                                   # :66
    if bs * N_blocks != n:
        raise ValueError("size(A,1) must be divisible by N_blocks")
    A = A.tocsr()

    def dense(r0, c0):
        return A[r0:r0 + bs, c0:c0 + bs].toarray()

    def chol(M, blk):
        try:
            return sla.cholesky(M, lower=True, check_finite=False)
        except sla.LinAlgError:
            raise NotPositiveDefinite(blk) from None

    chos = [chol(dense(0, 0), 1)]                        # :67
    Cs = []
    for i in range(1, N_blocks):                         # :70
        r0 = i * bs
        B = dense(r0, r0 - bs)                           # :73
        C = _chol_forward(chos[-1], B.T).T               # :74
        Cs.append(C)                                     # :75
        D = dense(r0, r0)                                # :76
        # D - C*C' : Julia lowers X*X' to dsyrk; only the lower triangle is formed and read
        S = sla.blas.dsyrk(-1.0, C, beta=1.0, c=D, lower=1, trans=0)
        chos.append(chol(S, i + 1))                      # :77
    return TridiagonalCholeskyFactor(n, chos, Cs)


def forward_solve(F: TridiagonalCholeskyFactor, b: np.ndarray) -> np.ndarray:
    """y = L^-1 b  (:43-52).  b is (n,) or (n,k); returns the flat result."""
    N = F.n_blocks
    bch = make_chunks(b, N)
    x = [None] * N
    x[0] = _chol_forward(F.chos[0], bch[0])                                     # :47
    for i in range(1, N):
        x[i] = _chol_forward(F.chos[i], bch[i] - _mm(F.Cs[i - 1], x[i - 1]))      # :49
    return np.concatenate(x, axis=0)


def backward_solve(F: TridiagonalCholeskyFactor, b: np.ndarray) -> np.ndarray:
    """x = L^-T b  (:24-33)."""
    N = F.n_blocks
    bch = make_chunks(b, N)
    x = [None] * N
    x[N - 1] = _chol_backward(F.chos[N - 1], bch[N - 1])                        # :28
    for i in range(N - 2, -1, -1):
        x[i] = _chol_backward(F.chos[i], bch[i] - _mm(F.Cs[i], x[i + 1], True))    # :30
    return np.concatenate(x, axis=0)


def ldiv(F: TridiagonalCholeskyFactor, b: np.ndarray) -> np.ndarray:
    """:60-63."""
    return backward_solve(F, forward_solve(F, b))


def ldiv_(y: np.ndarray, F: TridiagonalCholeskyFactor, b: np.ndarray) -> np.ndarray:
    """ldiv!(y, L, b)  (:54-58)."""
    y[...] = ldiv(F, b)
    return y


def logdet(F: TridiagonalCholeskyFactor) -> float:
    """2 sum log diag(L_i)  (the way the scripts get log-determinants from a Cholesky,
    scripts/burgers/solve_burgers_gmrf-collocation.jl:208-211)."""
    return 2.0 * float(sum(np.log(np.diag(L)).sum() for L in F.chos))


def extract_blocks(I, J, V, block_size: int):
    """scripts/solve_burger.jl:182-254 on 1-based COO triplets: stable sort by row, keep an
    entry when its column is in the same block range (diagonal block) or in the previous
    one (lower off-diagonal block); every other entry is dropped, as in the reference.
    Returns (diag_blocks, off_diag_blocks) as CSC matrices with duplicates summed."""
    I = np.asarray(I, dtype=np.int64)
    J = np.asarray(J, dtype=np.int64)
    V = np.asarray(V)
    p = np.argsort(I, kind="stable")                      # :183
    I, J, V = I[p], J[p], V[p]
    diag, off = [], []
    cur = ([], [], [])
    curo = ([], [], [])
    lo = 1                                                # cur_diag_block_range = lo:lo+bs-1
    for idx in range(I.shape[0]):
        i = int(I[idx])
        while i > lo + block_size - 1:                    # :205
            diag.append(cur)
            if lo - block_size > 0:                       # cur_off_diag_block_range[1] > 0
                off.append(curo)
            cur = ([], [], [])
            curo = ([], [], [])
            lo += block_size
        j = int(J[idx])
        if lo <= j <= lo + block_size - 1:                # :228
            cur[0].append(i - lo); cur[1].append(j - lo); cur[2].append(V[idx])
        if lo - block_size <= j <= lo - 1:                # :234
            curo[0].append(i - lo); curo[1].append(j - (lo - block_size)); curo[2].append(V[idx])
    diag.append(cur)
    if lo - block_size > 0:
        off.append(curo)

    def mk(t):
        return sp.coo_matrix((np.asarray(t[2], dtype=V.dtype), (t[0], t[1])),
                             shape=(block_size, block_size)).tocsc()

    return [mk(t) for t in diag], [mk(t) for t in off]


# ----------------------------------------------------------------------------- posterior use

def posterior_mean(F: TridiagonalCholeskyFactor, rhs: np.ndarray) -> np.ndarray:
    """mean(x_cond) = Q_post^-1 (information vector)  (solve_darcy_gmrf-fem.jl:190)."""
    return ldiv(F, rhs)


def sample(F: TridiagonalCholeskyFactor, mean: np.ndarray, Z: np.ndarray) -> np.ndarray:
    """rand(rng, x_cond) = mean + L^-T z  (solve_darcy_gmrf-fem.jl:191)."""
    X = backward_solve(F, Z)
    return X + (mean[:, None] if Z.ndim == 2 else mean)


def marginal_variances_exact(F: TridiagonalCholeskyFactor, last_blocks: int = 0) -> np.ndarray:
    """diag(A^-1) by block-tridiagonal selected inversion:
    S_NN = L_N^-T L_N^-1,  S_ii = L_i^-T (I + C_i^T S_{i+1,i+1} C_i) L_i^-1   (C_i = Cs[i]).
    The recursion runs from the last block upwards; `last_blocks` > 0 stops after that many blocks and
    returns their variances only (the tail of the vector) -- enough to pin a large case cheaply."""
    N, bs = F.n_blocks, F.block_size
    stop = 0 if last_blocks <= 0 else max(0, N - last_blocks)
    out = np.empty(F.N)
    eye = np.eye(bs)
    Li = _chol_forward(F.chos[N - 1], eye)
    S = _mm(Li, Li, True)
    out[(N - 1) * bs:] = np.diag(S)
    for i in range(N - 2, stop - 1, -1):
        C = F.Cs[i]
        Li = _chol_forward(F.chos[i], eye)
        S = _mm(Li, _mm(eye + _mm(C, _mm(S, C), True), Li), True)
        S = 0.5 * (S + S.T)
        out[i * bs:(i + 1) * bs] = np.diag(S)
    return out[stop * bs:]


def marginal_variances_rbmc(Q, X: np.ndarray) -> np.ndarray:
    """Rao-Blackwellised Monte-Carlo variances from centred samples X (n x S):
    var_i ~ 1/Q_ii + mean_s ( (1/Q_ii) sum_{j != i} Q_ij x_j^(s) )^2
    (RBMCStrategy of GaussianMarkovRandomFields.jl, solve_darcy_gmrf-fem.jl:100,192)."""
    Q = sp.csr_matrix(Q)
    d = Q.diagonal()
    QX = Q @ X
    off = (QX - d[:, None] * X) / d[:, None]
    return 1.0 / d + np.mean(off * off, axis=1)


def marginal_variances_mc(X: np.ndarray) -> np.ndarray:
    """Plain Monte-Carlo variances from centred samples."""
    return np.mean(X * X, axis=1)


# ----------------------------------------------------------------------------- helpers

def assemble_posterior(Q, J, noise: float):
    """`A = Symmetric(Q + noise * J_mat' * J_mat)`  -- /root/reference/scripts/solve_burger.jl:145
    (and `Q + noise_fem * J_final' * J_final`, scripts/burgers/solve_burgers_gmrf-fem.jl:186)."""
    import scipy.sparse as sp
    Q = sp.csc_matrix(Q)
    J = sp.csr_matrix(J)
    A = (Q + noise * (J.T @ J)).tocsc()
    A.sort_indices()
    return A


def gn_rhs(Qx_prior: np.ndarray, J, x: np.ndarray, obs_diff: np.ndarray, noise: float) -> np.ndarray:
    """`rhs = Qx_prior + noise * J_mat' * Array(J_mat * x + obs_diff)`  -- scripts/solve_burger.jl:146."""
    return Qx_prior + noise * (J.T @ (J @ x + obs_diff))


def gn_step(Q, J, Qx_prior: np.ndarray, x: np.ndarray, obs_diff: np.ndarray, noise: float, N_blocks: int) -> np.ndarray:
    """One `gn_step` of scripts/solve_burger.jl:143-149 with the block-tridiagonal factor in the place
    of `cholesky(A; perm = perm)`: assemble, factor, solve."""
    A = assemble_posterior(Q, J, noise)
    return ldiv(tridiagonal_cholesky(A, N_blocks), gn_rhs(Qx_prior, J, x, obs_diff, noise))


def condition_on_observations(Q, mu, A, q_eps: float, y, N_blocks: int):
    """Linear-Gaussian conditioning as the reference's scripts use it
    (`condition_on_observations(x, A, Q_eps, ys)`, scripts/darcy/solve_darcy_gmrf-fem.jl:188-189; the
    function itself lives in the absent GaussianMarkovRandomFields.jl, semantics per SURVEY 8b):
    posterior precision Q + q_eps A'A, mean = Q_post^-1 (Q mu + q_eps A' y).
    Returns (Q_post, factor, mean)."""
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    Qp = assemble_posterior(Q, A, q_eps)
    F = tridiagonal_cholesky(Qp, N_blocks)
    mu = np.zeros(Qp.shape[0]) if mu is None else mu
    return Qp, F, ldiv(F, Q @ mu + q_eps * (A.T @ y))


def get_xy_idcs(point_x, point_y, x_coords, y_coords):
    """src/datasets/darcy.jl:30-34: nearest grid point per coordinate, `argmin(abs.(coords .- p))` (the first
    minimum wins).  Vectorised over points."""
    xi = np.argmin(np.abs(np.asarray(x_coords)[None, :] - np.asarray(point_x)[:, None]), axis=1)
    yi = np.argmin(np.abs(np.asarray(y_coords)[None, :] - np.asarray(point_y)[:, None]), axis=1)
    return xi, yi


def assemble_darcy_diff_matrix(nx: int, ny: int, x_coords, y_coords, coeff_mat, beta: float = 1.0):
    """`assemble_darcy_diff_matrix`, /root/reference/src/problems/darcy.jl:5-63, on the structured P1 mesh that
    stands in for the reference's Gmsh P2 mesh: nx x ny nodes on the unit square, x fastest,
    every quad cut by the diagonal n00 - n11 into the cells (n00, n10, n11) [all of them first] and
    (n00, n11, n01); one quadrature point per cell (the centroid, weight |T|).  Per cell (:27-59):
        coeff_val = coeff_mat[get_xy_idcs(x_q, x_coords, y_coords)...]          :39
        fe[i]    += beta * phi_i(x_q) * dOmega                                  :47
        Ge[i, j] += (grad phi_i . coeff_val grad phi_j) * dOmega                :50-52
    then `assemble!` (:58) and `apply!(G, f, ch)` (:61) with homogeneous Dirichlet data on the boundary
    nodes: constrained rows and columns zeroed, their diagonal set to meandiag(G) = sum |G_ii| / n, f zeroed
    there (Ferrite semantics).  Returns (G as CSR with the full 7-point pattern, explicit zeros kept, f)."""
    xs, ys = np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny)
    qx, qy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    n00 = (qy * nx + qx).ravel(); n10 = n00 + 1; n01 = n00 + nx; n11 = n01 + 1
    cells = np.concatenate([np.stack([n00, n10, n11], axis=1), np.stack([n00, n11, n01], axis=1)], axis=0)
    X, Y = xs[cells % nx], ys[cells // nx]                                  # cell_coords, (cells, 3)
    # P1 shape functions on the cell: phi_i = (a_i + b_i x + c_i y) / (2 |T|)
    b = np.stack([Y[:, 1] - Y[:, 2], Y[:, 2] - Y[:, 0], Y[:, 0] - Y[:, 1]], axis=1)
    c = np.stack([X[:, 2] - X[:, 1], X[:, 0] - X[:, 2], X[:, 1] - X[:, 0]], axis=1)
    area = 0.5 * np.abs(X[:, 0] * b[:, 0] + X[:, 1] * b[:, 1] + X[:, 2] * b[:, 2])
    xq, yq = ((X[:, 0] + X[:, 1]) + X[:, 2]) / 3.0, ((Y[:, 0] + Y[:, 1]) + Y[:, 2]) / 3.0    # spatial_coordinate :35
    xi, yi = get_xy_idcs(xq, yq, x_coords, y_coords)
    coeff_val = np.asarray(coeff_mat)[xi, yi]                               # :39
    n = nx * ny
    G = sp.lil_matrix((n, n))
    rows, cols, vals = [], [], []
    f = np.zeros(n)
    for i in range(3):                                                      # test functions :44
        np.add.at(f, cells[:, i], beta * (1.0 / 3.0) * area)                # :47, phi_i(centroid) = 1/3
        for j in range(3):                                                  # trial functions :49
            rows.append(cells[:, i]); cols.append(cells[:, j])
            vals.append((b[:, i] * b[:, j] + c[:, i] * c[:, j]) / (4.0 * area) * coeff_val)   # :50-52
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    G = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()           # assemble! :58 (duplicates summed, zeros kept)
    G.sort_indices()
    # apply!(G, f, ch) :61
    ixn, iyn = np.arange(n) % nx, np.arange(n) // nx
    constrained = (ixn == 0) | (iyn == 0) | (ixn == nx - 1) | (iyn == ny - 1)
    m = np.abs(G.diagonal()).sum() / n
    G = G.tocoo()
    data = G.data.copy()
    hit = constrained[G.row] | constrained[G.col]
    data[hit] = 0.0
    data[hit & (G.row == G.col)] = m
    G = sp.csr_matrix((data, (G.row, G.col)), shape=(n, n))
    G.sort_indices()
    f = np.where(constrained, 0.0, f)
    return G, f


# Lagrange{RefTriangle,2} in Ferrite's reference coordinates: vertex 1 at xi = (1, 0), vertex 2 at (0, 1), vertex 3 at (0, 0),
# then the edge nodes (1-2), (2-3), (3-1).  gamma = 1 - xi_x - xi_y.
def _p2_triangle_shape(xi_x, xi_y):
    g = 1.0 - xi_x - xi_y
    return np.array([xi_x * (2.0 * xi_x - 1.0), xi_y * (2.0 * xi_y - 1.0), g * (2.0 * g - 1.0), 4.0 * xi_x * xi_y, 4.0 * xi_y * g, 4.0 * xi_x * g])


def _p2_triangle_ref_grad(xi_x, xi_y):
    g = 1.0 - xi_x - xi_y
    return np.array([[4.0 * xi_x - 1.0, 0.0], [0.0, 4.0 * xi_y - 1.0], [-(4.0 * g - 1.0), -(4.0 * g - 1.0)],
                     [4.0 * xi_y, 4.0 * xi_x], [-4.0 * xi_y, 4.0 * (g - xi_y)], [4.0 * (g - xi_x), -4.0 * xi_x]])


# QuadratureRule{RefTriangle}(3) (src/utils.jl:33: element_order + 1), taken as the 4-point Dunavant rule of degree 3
# (weights sum to 1/2, one of them negative); the order of the points fixes the summation order on both sides.
P2_TRI_QPOINTS = ((1.0 / 3.0, 1.0 / 3.0, -27.0 / 96.0), (0.2, 0.2, 25.0 / 96.0), (0.6, 0.2, 25.0 / 96.0), (0.2, 0.6, 25.0 / 96.0))


def p2_lattice_cells(nx: int, ny: int):
    """Cells of the structured quadratic mesh: the P1 triangulation of nx x ny vertices (all lower triangles
    (n00, n10, n11) first, then all upper (n00, n11, n01)), every cell with its three edge midpoints; dofs are the points
    of the (2 nx - 1) x (2 ny - 1) lattice, x fastest.  Returns (cells (nc, 6) dof numbers in Ferrite's local order --
    vertices 1, 2, 3, then the nodes of the edges (1-2), (2-3), (3-1) --, X, Y (nc, 3) vertex coordinates)."""
    W = 2 * nx - 1
    xs, ys = np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny)
    qx, qy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    qx, qy = qx.ravel(), qy.ravel()
    lat = lambda I, J: J * W + I
    I0, J0 = 2 * qx, 2 * qy
    v_lo = [(I0, J0), (I0 + 2, J0), (I0 + 2, J0 + 2)]
    v_up = [(I0, J0), (I0 + 2, J0 + 2), (I0, J0 + 2)]
    out, Xs, Ys = [], [], []
    for v in (v_lo, v_up):
        nodes = list(v) + [((v[0][0] + v[1][0]) // 2, (v[0][1] + v[1][1]) // 2), ((v[1][0] + v[2][0]) // 2, (v[1][1] + v[2][1]) // 2),
                           ((v[2][0] + v[0][0]) // 2, (v[2][1] + v[0][1]) // 2)]
        out.append(np.stack([lat(I, J) for I, J in nodes], axis=1))
        Xs.append(np.stack([xs[I // 2] for I, _ in v], axis=1)); Ys.append(np.stack([ys[J // 2] for _, J in v], axis=1))
    return np.concatenate(out, axis=0), np.concatenate(Xs, axis=0), np.concatenate(Ys, axis=0)


def assemble_darcy_diff_matrix_p2(nx: int, ny: int, x_coords, y_coords, coeff_mat, beta: float = 1.0, constrain: bool = True):
    """`assemble_darcy_diff_matrix` (/root/reference/src/problems/darcy.jl:5-63) with the reference's own element:
    `Lagrange{RefTriangle,2}` and `QuadratureRule{RefTriangle}(3)` (src/utils.jl:32-33) on the structured mesh of
    p2_lattice_cells.  Per cell and quadrature point (:27-59):
        x_q = spatial_coordinate (straight-sided cell: xi_x x_1 + xi_y x_2 + gamma x_3)              :35
        coeff_val = coeff_mat[get_xy_idcs(x_q, ...)]                                                 :39
        dOmega = w_q |det J|,  grad N_i = J^-T grad_xi N_i,  J = [x_1 - x_3, x_2 - x_3]               :42
        fe[i] += beta N_i dOmega;  Ge[i, j] += (grad N_i . coeff_val grad N_j) dOmega                :47-52
    `assemble!` and `apply!` with homogeneous Dirichlet data on the boundary lattice points as in the P1 form.
    Returns (G CSR over the lattice with the pattern of all cell couplings, zeros kept, f)."""
    cells, X, Y = p2_lattice_cells(nx, ny)
    nc = cells.shape[0]
    W, H = 2 * nx - 1, 2 * ny - 1
    n = W * H
    a, b = X[:, 0] - X[:, 2], X[:, 1] - X[:, 2]            # J = [[a, b], [c, d]]
    c, d = Y[:, 0] - Y[:, 2], Y[:, 1] - Y[:, 2]
    det = a * d - b * c
    Ge = np.zeros((nc, 6, 6)); fe = np.zeros((nc, 6))
    for (xi_x, xi_y, wq) in P2_TRI_QPOINTS:
        g = 1.0 - xi_x - xi_y
        xq = (xi_x * X[:, 0] + xi_y * X[:, 1]) + g * X[:, 2]
        yq = (xi_x * Y[:, 0] + xi_y * Y[:, 1]) + g * Y[:, 2]
        xi, yi = get_xy_idcs(xq, yq, x_coords, y_coords)
        coeff_val = np.asarray(coeff_mat)[xi, yi]
        dO = wq * np.abs(det)
        N = _p2_triangle_shape(xi_x, xi_y)
        dN = _p2_triangle_ref_grad(xi_x, xi_y)
        gx = (d[:, None] * dN[None, :, 0] - c[:, None] * dN[None, :, 1]) / det[:, None]        # J^-T grad_xi
        gy = (-b[:, None] * dN[None, :, 0] + a[:, None] * dN[None, :, 1]) / det[:, None]
        for i in range(6):
            fe[:, i] += beta * N[i] * dO
            for j in range(6):
                Ge[:, i, j] += coeff_val * (gx[:, i] * gx[:, j] + gy[:, i] * gy[:, j]) * dO
    rows = np.repeat(cells[:, :, None], 6, axis=2).ravel()
    cols = np.repeat(cells[:, None, :], 6, axis=1).ravel()
    G = sp.coo_matrix((Ge.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    G.sort_indices()
    f = np.zeros(n)
    np.add.at(f, cells.ravel(), fe.ravel())
    I, J = np.arange(n) % W, np.arange(n) // W
    if not constrain:                 # (the raw element sums, for the tests that pin this restatement by what G is)
        return G, f
    constrained = (I == 0) | (J == 0) | (I == W - 1) | (J == H - 1)
    G = _apply_constraints(G, constrained)
    f = np.where(constrained, 0.0, f)
    return G, f


def reconstruct(F: TridiagonalCholeskyFactor) -> np.ndarray:
    """Dense L L^T from the block factor (tests only, small n)."""
    N, bs = F.n_blocks, F.block_size
    L = np.zeros((F.N, F.N))
    for i in range(N):
        L[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] = F.chos[i]
        if i > 0:
            L[i * bs:(i + 1) * bs, (i - 1) * bs:i * bs] = F.Cs[i - 1]
    return L @ L.T


def rmse(pred, soln):
    """src/metrics.jl:3-5."""
    return float(np.sqrt(np.mean((pred - soln) ** 2)))


def max_err(pred, soln):
    """src/metrics.jl:7-9."""
    return float(np.max(np.abs(pred - soln)))


def rel_err(pred, soln):
    """src/metrics.jl:11-13."""
    return float(np.linalg.norm(pred - soln) / np.linalg.norm(soln))


# --------------------------------------------------------------------------- Burgers tangent
def _p1_line_cellvalues(h: float):
    """CellValues of one cell of the uniform P1 line, QuadratureRule{1,RefCube}(3): per quadrature point the
    weight detJ dV, the shape values and the physical shape gradients (constant for P1)."""
    xi = (-np.sqrt(3.0 / 5.0), 0.0, np.sqrt(3.0 / 5.0))
    wq = (5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0)
    jac = 0.5 * h
    dN = (-0.5 / jac, 0.5 / jac)
    return [(jac * wq[q], (0.5 * (1.0 - xi[q]), 0.5 * (1.0 + xi[q])), dN) for q in range(3)]


def _p2_line_cellvalues(h: float):
    """CellValues of one cell of the uniform QUADRATIC line (`generate_grid(QuadraticLine, ...)`, `Lagrange{RefLine,2}`,
    `QuadratureRule{RefLine}(3)`: /root/reference/src/utils.jl:42-49): local dofs (left, right, middle) with
    N = (xi (xi - 1) / 2, xi (xi + 1) / 2, 1 - xi^2) on xi in [-1, 1]."""
    xi = (-np.sqrt(3.0 / 5.0), 0.0, np.sqrt(3.0 / 5.0))
    wq = (5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0)
    jac = 0.5 * h
    out = []
    for q in range(3):
        x = xi[q]
        N = (0.5 * x * (x - 1.0), 0.5 * x * (x + 1.0), 1.0 - x * x)
        dN = ((x - 0.5) / jac, (x + 0.5) / jac, (-2.0 * x) / jac)
        out.append((jac * wq[q], N, dN))
    return out


def _line_cells(ns: int, order: int):
    """(cell values, list of the cells' dof tuples) of the periodic line with ns dofs on [0,1).  order 1: ns cells
    (e, e + 1 mod ns).  order 2: ns / 2 cells with the dofs (2 e, 2 e + 2 mod ns, 2 e + 1) in Ferrite's local order
    (left, right, middle); the dofs are numbered by position (x_i = i / ns), the periodic constraint of the
    reference's mesh (get_periodic_constraint, src/utils.jl:5-18) is the wrap-around."""
    if order == 1:
        return _p1_line_cellvalues(1.0 / ns), [(e, (e + 1) % ns) for e in range(ns)]
    if order != 2 or ns % 2:
        raise ValueError("order 1, or order 2 with an even number of dofs")
    nc = ns // 2
    return _p2_line_cellvalues(1.0 / nc), [(2 * e, (2 * e + 2) % ns, 2 * e + 1) for e in range(nc)]


def assemble_burgers_advection_matrix(ns: int, cur_weights, order: int = 1):
    """Parity oracle of /root/reference/src/problems/burgers.jl:5-59 (`assemble_burgers_advection_matrix`) on the
    periodic line with ns dofs on [0,1) (P1: cell e has the dofs (e, e + 1 mod ns); P2: see _line_cells) -- the
    periodic constraint of the reference's mesh condensed into the wrap-around, so there are no prescribed dofs
    left to zero (:53-57).  Returns (G as CSR, v).  Line-by-line: the cell loop :22-51, quadrature loop :30-50."""
    cv, cells = _line_cells(ns, order)
    nb = order + 1
    G = sp.lil_matrix((ns, ns))
    v = np.zeros(ns)
    for dofs in cells:                                         # CellIterator(dh)
        Ge = np.zeros((nb, nb)); ve = np.zeros(nb)
        w = [cur_weights[d] for d in dofs]                     # :28
        for dOm, N, dN in cv:                                  # :30
            cur_u = 0.0                                        # :34 function_value
            for k in range(nb):
                cur_u += N[k] * w[k]
            grad_u = 0.0                                       # :36-39
            for k in range(nb):
                grad_u += dN[k] * w[k]
            for i in range(nb):                                # :40
                for j in range(nb):                            # :43
                    Ge[i, j] += N[i] * (N[j] * grad_u + cur_u * dN[j]) * dOm        # :46
                ve[i] += N[i] * cur_u * grad_u * dOm           # :48
        for i in range(nb):                                    # :51 assemble!
            for j in range(nb):
                G[dofs[i], dofs[j]] += Ge[i, j]
            v[dofs[i]] += ve[i]
    return G.tocsr(), v


def assemble_burgers_mass_diffusion_matrices(ns: int, order: int = 1):
    """/root/reference/src/problems/burgers.jl:60-98 on the same mesh (consistent mass, lumping = false):
    Me[i][j] = sum_q N_i N_j dOmega, Ge[i][j] = sum_q grad N_i grad N_j dOmega per cell (:80-83), assembled."""
    cv, cells = _line_cells(ns, order)
    nb = order + 1
    M = sp.lil_matrix((ns, ns)); G = sp.lil_matrix((ns, ns))
    for dofs in cells:
        Me = np.zeros((nb, nb)); Ge = np.zeros((nb, nb))
        for dOm, N, dN in cv:
            for i in range(nb):
                for j in range(nb):
                    Me[i, j] += N[i] * N[j] * dOm
                    Ge[i, j] += dN[i] * dN[j] * dOm
        for i in range(nb):
            for j in range(nb):
                M[dofs[i], dofs[j]] += Me[i, j]
                G[dofs[i], dofs[j]] += Ge[i, j]
    return M.tocsr(), G.tocsr()


def burgers_f_and_J(ns: int, nt: int, dt: float, nu: float, w, order: int = 1):
    """`f_and_J(w)` of /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:118-149:
    J_static = M_{t+1} - M_t + dt nu G_{t+1} (:123-130), per slice t = 2 .. nt the advection tangent and residual
    (:132-144), f = J_static w + dt f_adv, J = J_static + dt J_adv (:146-149).  Returns (f, J as CSR with sorted
    indices, (nt - 1) ns x nt ns)."""
    M, G = assemble_burgers_mass_diffusion_matrices(ns, order)
    Z = sp.csr_matrix((ns, ns))

    def s2st(A, t):                                            # spatial_to_spatiotemporal(A, t, nt), t 1-based
        return sp.hstack([A if k == t - 1 else Z for k in range(nt)], format="csr")

    Mt = sp.vstack([s2st(M, t) for t in range(1, nt)], format="csr")
    Mt1 = sp.vstack([s2st(M, t) for t in range(2, nt + 1)], format="csr")
    Gt1 = sp.vstack([s2st(G, t) for t in range(2, nt + 1)], format="csr")
    J_static = (Mt1 - Mt + (dt * nu) * Gt1).tocsr()
    Js, vs = [], []
    w = np.asarray(w, dtype=np.float64)
    for t in range(2, nt + 1):
        Gt, vt = assemble_burgers_advection_matrix(ns, w[(t - 1) * ns:t * ns], order)
        Js.append(s2st(Gt, t)); vs.append(vt)
    J_adv = sp.vstack(Js, format="csr"); f_adv = np.concatenate(vs)
    f = J_static @ w + dt * f_adv
    J = (J_static + dt * J_adv).tocsr()
    J.sort_indices()
    return f, J


# ------------------------------------------------------------------------------------------------
# Linear shallow-water SPDE: element loops and per-step operators of /root/reference/src/spdes/shallow_water.jl
# on the structured P1 triangle mesh that stands in for the reference's Gmsh mesh.
# Ferrite is absent, so two conventions are this restatement's own (both stated to the device kernel too):
#   * dof numbering: node-major, dof = 3 * node + field with fields (h, u, v) = (0, 1, 2) -- Ferrite numbers dofs
#     in cell-visit order; the operators are the same up to that permutation;
#   * quadrature: the symmetric 3-point rule of QuadratureRule{2,RefTetrahedron}(2) (weights 1/6 on the reference
#     triangle, so dOmega = |T| / 3); point q carries the barycentric weight 2/3 on cell vertex 2 - q and 1/6 on
#     the other two.  The rule is symmetric: the assembled matrices depend on the order only through rounding.

def _p1_triangle_cells(nx: int, ny: int):
    """Cells of the structured mesh (all lower triangles (n00, n10, n11) first, then all upper (n00, n11, n01)),
    their P1 gradient coefficients and areas: grad phi_v = (b_v, c_v) / (2 |T|) * sign."""
    xs, ys = np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny)
    qx, qy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    n00 = (qy * nx + qx).ravel(); n10 = n00 + 1; n01 = n00 + nx; n11 = n01 + 1
    cells = np.concatenate([np.stack([n00, n10, n11], axis=1), np.stack([n00, n11, n01], axis=1)], axis=0)
    X, Y = xs[cells % nx], ys[cells // nx]
    b = np.stack([Y[:, 1] - Y[:, 2], Y[:, 2] - Y[:, 0], Y[:, 0] - Y[:, 1]], axis=1)
    c = np.stack([X[:, 2] - X[:, 1], X[:, 0] - X[:, 2], X[:, 1] - X[:, 0]], axis=1)
    # signed 2 |T| = det of the cell Jacobian (x1 - x0)(y2 - y0) - (x2 - x0)(y1 - y0), as reinit! forms it (the equivalent
    # sum_v x_v b_v cancels (n - 1)-fold on this mesh, and S ~ 1 / |T| would carry that rounding)
    area2 = c[:, 2] * b[:, 1] - c[:, 1] * b[:, 2]
    return cells, X, Y, b / area2[:, None], c / area2[:, None], 0.5 * np.abs(area2)


SWE_Q_BARY = np.array([[1 / 6, 1 / 6, 2 / 3], [1 / 6, 2 / 3, 1 / 6], [2 / 3, 1 / 6, 1 / 6]])   # phi_v at point q: [q][v]


def shallow_water_qpoints(nx: int, ny: int) -> np.ndarray:
    """spatial_coordinate(cvh, qp, cell_coords) (:52) of every cell: (cells, 3, 2)."""
    _, X, Y, _, _, _ = _p1_triangle_cells(nx, ny)
    xq = (SWE_Q_BARY[None, :, 0] * X[:, None, 0] + SWE_Q_BARY[None, :, 1] * X[:, None, 1]) + SWE_Q_BARY[None, :, 2] * X[:, None, 2]
    yq = (SWE_Q_BARY[None, :, 0] * Y[:, None, 0] + SWE_Q_BARY[None, :, 1] * Y[:, None, 1]) + SWE_Q_BARY[None, :, 2] * Y[:, None, 2]
    return np.stack([xq, yq], axis=2)


def _apply_constraints(A: sp.spmatrix, prescribed) -> sp.csr_matrix:
    """Ferrite `apply!(A, zeros(n), ch)` for homogeneous data (:119-121): constrained rows and columns zeroed, their
    diagonal set to meandiag(A) = sum |A_ii| / n."""
    A = sp.coo_matrix(A)
    n = A.shape[0]
    if prescribed is None or not np.any(prescribed):
        out = sp.csr_matrix(A); out.sort_indices(); return out
    pres = np.asarray(prescribed, dtype=bool)
    m = np.abs(sp.csr_matrix(A).diagonal()).sum() / n
    data = A.data.copy()
    hit = pres[A.row] | pres[A.col]
    data[hit] = 0.0
    data[hit & (A.row == A.col)] = m
    out = sp.csr_matrix((data, (A.row, A.col)), shape=A.shape)
    out.sort_indices()
    return out


def assemble_shallow_water_system(nx: int, ny: int, H_q, k: float, f: float, g: float, prescribed=None):
    """`assemble_system!`, /root/reference/src/spdes/shallow_water.jl:17-122.  H_q[cell, q] = H(x_q) (:53).
    Per cell and quadrature point (:51-111), with I = 3 * vertex + field:
        h-h  me += phi_i phi_j dO            se += grad phi_i . grad phi_j dO                      :68-71
        h-u  ke += -H dx(phi_i) phi_j dO     h-v  ke += -H dy(phi_i) phi_j dO                      :73-79
        u-h  ke += -g dx(phi_i) phi_j dO     v-h  ke += -g dy(phi_i) phi_j dO                      :82-84, :99-101
        u-u / v-v  me += phi_i phi_j dO, ke += k phi_i phi_j dO, se += grad . grad dO              :86-91, :107-112
        u-v  ke += -f phi_i phi_j dO         v-u  ke += f phi_i phi_j dO                           :93-95, :103-105
    then `assemble!` of ke, of the element-lumped me (`lump_matrix(me, ip)`, :116: row sums on the diagonal for
    the linear Lagrange interpolation) and of se (:114-118), and `apply!` of the constraint handler to all three
    (:119-121).  Returns (K: CSR 3 nn x 3 nn with the full field coupling over the 7-point node stencil, explicit
    zeros kept, like create_sparsity_pattern(dh, ch) :140; M: lumped mass as a vector (its off-diagonal pattern
    entries are zeros); S: CSR with the block-diagonal coupling of :141-150)."""
    cells, X, Y, gx, gy, area = _p1_triangle_cells(nx, ny)
    nc, nn = cells.shape[0], nx * ny
    H_q = np.asarray(H_q, dtype=np.float64).reshape(nc, 3)
    dO = area / 3.0
    ke = np.zeros((nc, 9, 9)); me = np.zeros((nc, 9, 9)); se = np.zeros((nc, 9, 9))
    rh, ru, rv = 0, 1, 2                                          # field of local dof I = 3 * vertex + field
    for qp in range(3):                                           # :51
        phi = SWE_Q_BARY[qp]                                      # shape_value: the same for the three fields (:56-66)
        Hv = H_q[:, qp]
        for i in range(3):
            for j in range(3):
                pp = phi[i] * phi[j] * dO
                gg = (gx[:, i] * gx[:, j] + gy[:, i] * gy[:, j]) * dO
                Ih, Iu, Iv = 3 * i + rh, 3 * i + ru, 3 * i + rv
                Jh, Ju, Jv = 3 * j + rh, 3 * j + ru, 3 * j + rv
                me[:, Ih, Jh] += pp; se[:, Ih, Jh] += gg                                  # h - h
                ke[:, Ih, Ju] += -Hv * gx[:, i] * phi[j] * dO                             # h - u
                ke[:, Ih, Jv] += -Hv * gy[:, i] * phi[j] * dO                             # h - v
                ke[:, Iu, Jh] += -g * gx[:, i] * phi[j] * dO                              # u - h
                me[:, Iu, Ju] += pp; ke[:, Iu, Ju] += k * pp; se[:, Iu, Ju] += gg         # u - u
                ke[:, Iu, Jv] += -f * pp                                                  # u - v
                ke[:, Iv, Jh] += -g * gy[:, i] * phi[j] * dO                              # v - h
                ke[:, Iv, Ju] += f * pp                                                   # v - u
                me[:, Iv, Jv] += pp; ke[:, Iv, Jv] += k * pp; se[:, Iv, Jv] += gg         # v - v
    dofs = (3 * cells[:, :, None] + np.arange(3)[None, None, :]).reshape(nc, 9)           # celldofs!: I = 3 * vertex + field
    rows = np.repeat(dofs[:, :, None], 9, axis=2).ravel()
    cols = np.repeat(dofs[:, None, :], 9, axis=1).ravel()
    n = 3 * nn
    K = sp.coo_matrix((ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()                   # duplicates summed in cell order
    K.sort_indices()
    # block-diagonal coupling of M and S (:141-150): entries between different fields are not part of the pattern
    same = (rows % 3) == (cols % 3)
    S = sp.coo_matrix((se.ravel()[same], (rows[same], cols[same])), shape=(n, n)).tocsr()
    S.sort_indices()
    ml = me.sum(axis=2)                                                                   # lump_matrix: row sums (:116)
    M = np.zeros(n)
    np.add.at(M, dofs.ravel(), ml.ravel())
    if prescribed is not None and np.any(prescribed):
        pres = np.asarray(prescribed, dtype=bool)
        K, S = _apply_constraints(K, pres), _apply_constraints(S, pres)
        m = np.abs(M).sum() / n                                                            # apply!(M, ...) on the diagonal matrix
        M = np.where(pres, m, M)
    return K, M, S


def shallow_water_operators(K, M, S, prescribed, kappa_matern: float, tau: float, dt: float):
    """The operators `discretize` builds from K, M, S (/root/reference/src/spdes/shallow_water.jl:170-217) for one time
    step dt: M~ (prescribed diagonal 1e-2, :170-174), K_matern = kappa^2 M~ + G with G = S, prescribed diagonal 1
    (:172,:178), the square root of the initial precision J = sqrt(ratio) M~^-1/2 K_matern (Q_matern = J'J, :187-189,
    ratio = Gamma(nu) / (Gamma(nu + 1) 4 pi kappa^(2 nu)), nu = 2, :180-184), the noise scalings beta(dt) = sqrt(dt) tau
    (1e-2 on prescribed dofs, :198-211) and the step matrix G(dt) = M~ + dt K with the constraints applied (:212-217)."""
    from math import gamma, pi, sqrt
    n = K.shape[0]
    pres = np.zeros(n, dtype=bool) if prescribed is None else np.asarray(prescribed, dtype=bool)
    Mt = np.where(pres, 1e-2, M)                                   # :172-174
    G = sp.lil_matrix(S)
    for d in np.flatnonzero(pres):
        G[d, d] = 1.0                                              # :173
    G = sp.csr_matrix(G)
    K_matern = (kappa_matern ** 2 * sp.diags(Mt) + G).tocsr()      # :178
    nu = 2
    ratio = gamma(nu) / (gamma(nu + 1) * (4 * pi) * kappa_matern ** (2 * nu))          # :180-184 (sigma^2_goal = 1)
    J = (sqrt(ratio) * sp.diags(np.sqrt(1.0 / Mt)) @ K_matern).tocsr()                # Q_matern_sqrt' (:188-189)
    J.sort_indices()
    noise = np.where(pres, 1e-2, tau)                              # :198, :204
    beta, beta_inv = sqrt(dt) * noise, (1.0 / sqrt(dt)) / noise    # :210-211
    Gdt = _apply_constraints(sp.diags(Mt) + dt * K, pres)          # :212-217
    return {"M_tilde": Mt, "K_matern": K_matern, "J": J, "ratio": ratio, "beta": beta, "beta_inv": beta_inv, "G_dt": Gdt}
