"""Synthetic FEM workloads for the block-tridiagonal GMRF hot path.

These generators build the *inputs* of the hot path (sparse SPD block-tridiagonal
posterior precisions and right-hand sides) for the BASELINE.json configs on
structured meshes.  The reference builds the same kind of matrices with
Ferrite/Gmsh P2 meshes and the FNO datasets, neither of which exists here
(SURVEY.md section 8d), so structured P1 meshes and a synthetic coefficient field
replace them.  Nothing in here is on the timed path.

Reference call sites the constructions follow:
  * Darcy stiffness / load:    /root/reference/src/problems/darcy.jl:27-62
  * nearest-grid-point lookup: /root/reference/src/datasets/darcy.jl:30-34
  * Matern prior hyper-params: /root/reference/scripts/darcy/solve_darcy_gmrf-fem.jl:92-98
  * observation noise Q_eps:   /root/reference/scripts/darcy/solve_darcy_gmrf-fem.jl:163
  * Burgers J_static:          /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:118-149
  * Burgers prior parameters:  /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:86-107
  * elliptic (Chen) setup:     /root/reference/_research/elliptic_chen24.jl:118-171,231-285
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


@dataclass
class Workload:
    """One posterior-solve problem: factor Q, solve Q mu = rhs, sample N(mu, Q^-1)."""

    name: str
    Q: sp.csc_matrix          # SPD, block tridiagonal with n_blocks blocks of size n // n_blocks
    rhs: np.ndarray           # information vector; posterior mean = Q^-1 rhs
    n_blocks: int
    meta: dict = field(default_factory=dict)

    @property
    def n(self) -> int:
        return self.Q.shape[0]

    @property
    def block_size(self) -> int:
        return self.n // self.n_blocks


# --------------------------------------------------------------------------- 2-D P1 FEM

def _grid_triangles(nx: int, ny: int):
    """Node ids of the 2*(nx-1)*(ny-1) triangles; every quad is cut by the same diagonal."""
    ix, iy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    n00 = (iy * nx + ix).ravel()
    n10 = n00 + 1
    n01 = n00 + nx
    n11 = n01 + 1
    lower = np.stack([n00, n10, n11], axis=1)
    upper = np.stack([n00, n11, n01], axis=1)
    return np.concatenate([lower, upper], axis=0)


def p1_unit_square(nx: int, ny: int, coeff=None):
    """Lumped mass (diag), stiffness G and coefficient-weighted stiffness D on the unit square.

    Nodes are lexicographic with x fastest, so a block of `w` consecutive node rows is a
    contiguous index range: the ordering the block-tridiagonal partition needs.
    `coeff(xc, yc)` is evaluated at the element centroids (one-point quadrature, the P1
    analogue of the quadrature-point lookup in src/problems/darcy.jl:37-39).
    """
    tri = _grid_triangles(nx, ny)
    xs = np.linspace(0.0, 1.0, nx)
    ys = np.linspace(0.0, 1.0, ny)
    X = np.tile(xs, ny)
    Y = np.repeat(ys, nx)
    x = X[tri]
    y = Y[tri]
    # P1 gradients: grad phi_i = (b_i, c_i) / (2 area)
    b = np.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], axis=1)
    c = np.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], axis=1)
    area2 = x[:, 0] * b[:, 0] + x[:, 1] * b[:, 1] + x[:, 2] * b[:, 2]
    area = 0.5 * np.abs(area2)
    Ke = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * area[:, None, None])
    n = nx * ny
    rows = np.repeat(tri, 3, axis=1).ravel()
    cols = np.tile(tri, (1, 3)).ravel()
    G = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    lumped = np.bincount(tri.ravel(), weights=np.repeat(area / 3.0, 3), minlength=n)
    D = None
    if coeff is not None:
        a = coeff(x.mean(axis=1), y.mean(axis=1))
        D = sp.coo_matrix(((Ke * a[:, None, None]).ravel(), (rows, cols)), shape=(n, n)).tocsr()
    return lumped, G, D, (X, Y)


def _dirichlet(D: sp.csr_matrix, f: np.ndarray, boundary: np.ndarray):
    """Zero the prescribed rows/columns and put a unit-scale value on their diagonal
    (Ferrite `apply!(G, f, ch)` semantics, src/problems/darcy.jl:61)."""
    keep = np.ones(D.shape[0])
    keep[boundary] = 0.0
    P = sp.diags(keep)
    scale = float(np.mean(D.diagonal()))
    Dd = (P @ D @ P + sp.diags((1.0 - keep) * scale)).tocsr()
    fd = f * keep
    return Dd, fd


def darcy_coefficient(seed: int = 523802340, n_grid: int = 241, n_modes: int = 16):
    """Piecewise-constant a(x) in {3, 12}: a smooth random Fourier field thresholded at 0,
    tabulated on the 241x241 grid of the FNO Darcy dataset (src/datasets/darcy.jl:12-13)
    and looked up nearest-neighbour like get_xy_idcs (src/datasets/darcy.jl:30-34).
    The seed is the one the reference script uses (solve_darcy_gmrf-fem.jl:55)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kx = rng.integers(1, 6, n_modes)
    ky = rng.integers(1, 6, n_modes)
    amp = rng.standard_normal(n_modes)
    ph = rng.uniform(0, 2 * np.pi, n_modes)
    g = np.linspace(0.0, 1.0, n_grid)
    gx, gy = np.meshgrid(g, g, indexing="ij")
    fld = np.zeros_like(gx)
    for m in range(n_modes):
        fld += amp[m] * np.cos(np.pi * (kx[m] * gx + ky[m] * gy) + ph[m])
    table = np.where(fld > 0.0, 12.0, 3.0)

    def coeff(xc, yc):
        i = np.clip(np.rint(xc * (n_grid - 1)).astype(np.int64), 0, n_grid - 1)
        j = np.clip(np.rint(yc * (n_grid - 1)).astype(np.int64), 0, n_grid - 1)
        return table[i, j]

    return coeff


def matern_precision_2d(lumped, G, kappa: float, alpha: int):
    """Q_alpha = tau^2 K (C^-1 K)^(alpha-1), K = kappa^2 C + G, scaled to unit marginal variance
    (Lindgren et al. SPDE construction; what MaternSPDE{2} + discretize produce in the
    reference, scripts/darcy/solve_darcy_gmrf-fem.jl:92-98)."""
    C = sp.diags(lumped)
    Ci = sp.diags(1.0 / lumped)
    K = (kappa ** 2) * C + G
    Q = K
    for _ in range(alpha - 1):
        Q = K @ Ci @ Q
    nu = alpha - 1.0  # d = 2
    tau2 = math.gamma(nu) / (math.gamma(alpha) * 4.0 * math.pi * kappa ** (2.0 * nu))
    Q = (tau2 * Q).tocsr()
    return ((Q + Q.T) * 0.5).tocsr()


def darcy(n_xy: int, rows_per_block: int = 4, q_eps: float = 1e8, beta: float = 1.0,
          seed: int = 523802340) -> Workload:
    """2-D Darcy posterior precision Q_post = Q_matern(alpha=3) + q_eps D^T D and the
    information vector rhs = q_eps D^T f (prior mean zero).

    Mirrors scripts/darcy/solve_darcy_gmrf-fem.jl:176-192: `condition_on_observations(x, A=D,
    Q_eps, y=f)` followed by mean / rand / std.
    """
    assert n_xy % rows_per_block == 0
    coeff = darcy_coefficient(seed)
    lumped, G, D, (X, Y) = p1_unit_square(n_xy, n_xy, coeff)
    rng_range = 1.0 / math.sqrt(n_xy)            # script :98
    kappa = math.sqrt(8.0 * 2.0) / rng_range     # smoothness 2
    Q0 = matern_precision_2d(lumped, G, kappa, alpha=3)
    f = beta * lumped.copy()                      # f_i = beta * int phi_i
    on_bnd = (X == 0.0) | (X == 1.0) | (Y == 0.0) | (Y == 1.0)
    Dd, fd = _dirichlet(D, f, np.flatnonzero(on_bnd))
    Q = (Q0 + q_eps * (Dd.T @ Dd)).tocsc()
    Q = ((Q + Q.T) * 0.5).tocsc()
    Q.sort_indices()
    rhs = q_eps * (Dd.T @ fd)
    return Workload(f"darcy{n_xy}", Q, np.asarray(rhs), n_xy // rows_per_block,
                    {"kappa": kappa, "q_eps": q_eps, "rows_per_block": rows_per_block,
                     "nnz": int(Q.nnz), "mesh": f"{n_xy}x{n_xy} P1"})


def darcy_conditioning(n_xy: int, rows_per_block: int = 4, q_eps: float = 1e8, beta: float = 1.0,
                       seeds=(523802340,)):
    """The ingredients of the reference's Darcy problem loop (scripts/darcy/solve_darcy_gmrf-fem.jl:
    176-192) on the mesh of `darcy`: the Matern prior precision Q0 (mean zero), and per seed the
    observation pair (A = Dirichlet-modified Darcy stiffness for that coefficient field, y = load) --
    all A share one sparsity pattern.  Returns (Q0, [(A, y), ...], n_blocks)."""
    obs = []
    pat = None
    for seed in seeds:
        coeff = darcy_coefficient(seed)
        lumped, G, D, (X, Y) = p1_unit_square(n_xy, n_xy, coeff)
        on_bnd = (X == 0.0) | (X == 1.0) | (Y == 0.0) | (Y == 1.0)
        Dd, fd = _dirichlet(D, beta * lumped.copy(), np.flatnonzero(on_bnd))
        if pat is None:
            pat = abs(p1_unit_square(n_xy, n_xy, lambda x, y: np.ones_like(x))[2]).tocsr() + sp.identity(n_xy * n_xy)
            pat.data[:] = 0.0
        Dd = (Dd + pat).tocsr()                   # explicit zeros: one pattern for every coefficient field
        Dd.sort_indices()
        obs.append((Dd, fd))
    kappa = math.sqrt(8.0 * 2.0) * math.sqrt(n_xy)
    lumped, G, _, _ = p1_unit_square(n_xy, n_xy)
    Q0 = matern_precision_2d(lumped, G, kappa, alpha=3).tocsc()
    Q0.sort_indices()
    return Q0, obs, n_xy // rows_per_block


def elliptic(n_xy: int, rows_per_block: int = 2, bnd_noise: float = 1e12,
             fem_noise: float = 3e13) -> Workload:
    """Nonlinear elliptic -Lap u + u^3 = f (Chen et al.) linearised at the true solution:
    Q_post = Q_matern(alpha=2, range 0.1) + bnd_noise A_b^T A_b + fem_noise J^T J with
    J = G + 3 u^2-weighted lumped mass (_research/elliptic_chen24.jl:118-171, 231-285)."""
    assert n_xy % rows_per_block == 0
    lumped, G, _, (X, Y) = p1_unit_square(n_xy, n_xy)
    kappa = math.sqrt(8.0 * 1.0) / 0.1
    Q0 = matern_precision_2d(lumped, G, kappa, alpha=2)
    u = np.sin(np.pi * X) * np.sin(np.pi * Y) + 4.0 * np.sin(4 * np.pi * X) * np.sin(4 * np.pi * Y)
    on_bnd = (X == 0.0) | (X == 1.0) | (Y == 0.0) | (Y == 1.0)
    interior = (~on_bnd).astype(np.float64)
    J = (sp.diags(interior) @ (G + sp.diags(3.0 * u * u * lumped))).tocsr()
    Ab = sp.diags(on_bnd.astype(np.float64)).tocsr()
    Q = (Q0 + bnd_noise * (Ab.T @ Ab) + fem_noise * (J.T @ J)).tocsc()
    Q = ((Q + Q.T) * 0.5).tocsc()
    Q.sort_indices()
    lap_u = 2 * np.pi ** 2 * np.sin(np.pi * X) * np.sin(np.pi * Y) \
        + 4.0 * 32 * np.pi ** 2 * np.sin(4 * np.pi * X) * np.sin(4 * np.pi * Y)
    fvals = lap_u + u ** 3
    resid = interior * lumped * fvals
    rhs = fem_noise * (J.T @ resid)
    return Workload(f"elliptic{n_xy}", Q, np.asarray(rhs), n_xy // rows_per_block,
                    {"kappa": kappa, "rows_per_block": rows_per_block, "nnz": int(Q.nnz)})


# --------------------------------------------------------------------------- 1-D space-time

def p1_periodic_line(ns: int):
    """Periodic P1 line on [0,1): consistent mass M, lumped mass, stiffness S, advection Adv."""
    h = 1.0 / ns
    i = np.arange(ns)
    ip = (i + 1) % ns
    im = (i - 1) % ns
    M = sp.coo_matrix((np.r_[np.full(ns, 4 * h / 6), np.full(ns, h / 6), np.full(ns, h / 6)],
                       (np.r_[i, i, i], np.r_[i, ip, im])), shape=(ns, ns)).tocsr()
    S = sp.coo_matrix((np.r_[np.full(ns, 2 / h), np.full(ns, -1 / h), np.full(ns, -1 / h)],
                       (np.r_[i, i, i], np.r_[i, ip, im])), shape=(ns, ns)).tocsr()
    Adv = sp.coo_matrix((np.r_[np.full(ns, 0.5), np.full(ns, -0.5)],
                         (np.r_[i, i], np.r_[ip, im])), shape=(ns, ns)).tocsr()
    return M, np.full(ns, h), S, Adv


def burgers(ns: int, nt: int, ic_noise: float = 1e8, fem_noise: float = 1e12) -> Workload:
    """1-D viscous Burgers space-time GMRF in the time-major ordering (t-1)*ns + s:
    Q = Q_prior + ic_noise A_ic^T A_ic + fem_noise J^T J, N = nt blocks of size ns.

    Q_prior: implicit-Euler state-space blocks  G x_{t+1} = M x_t + noise  with
    G = M + dt (nu S + gamma Adv) and a Matern(alpha=2) initial precision
    (scripts/burgers/solve_burgers_gmrf-fem.jl:86-107; block structure of joint_ssm,
    SURVEY.md appendix A).  J = J_static + dt J_adv(u) linearised at the initial
    condition (scripts/burgers/solve_burgers_gmrf-fem.jl:118-149)."""
    nu_b = 0.01 / math.pi
    dt = 1.0 / (nt - 1)
    M, lumped, S, Adv = p1_periodic_line(ns)
    xs = np.arange(ns) / ns
    ic = np.sin(2 * np.pi * xs) + 0.5 * np.sin(4 * np.pi * xs + 0.3)
    bulk = float(ic.mean())
    c = 1.0 / nu_b
    gamma = -c * bulk
    tau = 0.1 * math.sqrt(c)
    kappa = math.sqrt(8.0 * 1.5) / math.sqrt(1.0 / ns)
    Ml = sp.diags(lumped)
    K = (kappa ** 2) * Ml + S
    Q0 = (K @ sp.diags(1.0 / lumped) @ K).tocsr()
    Gm = (Ml + dt * (nu_b * c * S + gamma * Adv)).tocsr()
    W = sp.diags(np.full(ns, 1.0 / (dt * tau * tau)) / lumped)
    GWG = (Gm.T @ W @ Gm).tocsr()
    MWM = (Ml.T @ W @ Ml).tocsr()
    GWM = (Gm.T @ W @ Ml).tocsr()
    blocks = [[None] * nt for _ in range(nt)]
    for t in range(nt):
        d = GWG if t > 0 else Q0
        if t < nt - 1:
            d = d + MWM
        blocks[t][t] = d
        if t > 0:
            blocks[t][t - 1] = -GWM
            blocks[t - 1][t] = -GWM.T
    Qp = sp.bmat(blocks, format="csr")
    # Burgers residual tangent: rows couple slices t-1 and t only.
    u = ic
    Jadv = (sp.diags(u) @ Adv + sp.diags(Adv @ u)).tocsr()   # d/du of u u_x, lumped
    Jt = (M + dt * nu_b * S + dt * Jadv).tocsr()
    rowsJ = []
    for t in range(1, nt):
        row = [None] * nt
        row[t - 1] = -M
        row[t] = Jt
        for k in range(nt):
            if row[k] is None:
                row[k] = sp.csr_matrix((ns, ns))
        rowsJ.append(row)
    J = sp.bmat(rowsJ, format="csr")
    Aic = sp.hstack([sp.identity(ns, format="csr"), sp.csr_matrix((ns, ns * (nt - 1)))]).tocsr()
    Q = (Qp + ic_noise * (Aic.T @ Aic) + fem_noise * (J.T @ J)).tocsc()
    Q = ((Q + Q.T) * 0.5).tocsc()
    Q.sort_indices()
    mu0 = np.full(ns * nt, bulk)
    rhs = Qp @ mu0 + ic_noise * (Aic.T @ ic)
    return Workload(f"burgers{ns}x{nt}", Q, np.asarray(rhs), nt,
                    {"dt": dt, "nu": nu_b, "nnz": int(Q.nnz)})


def burgers_gauss_newton(ns: int, nt: int, ic_noise: float = 1e8, fem_noise: float = 1e12):
    """The pieces of the reference's Gauss-Newton loop for the Burgers space-time GMRF
    (scripts/solve_burger.jl:118-180; residual tangent scripts/burgers/solve_burgers_gmrf-fem.jl:118-149)
    on the mesh of `burgers`: returns a dict with
      Q          prior precision with the initial condition conditioned in (CSC, fixed),
      Qx_prior   Q * x_prior (information vector of the prior part),
      x_prior    starting point,
      residual   x -> r(x), the implicit-Euler Burgers residual of slices 1..nt-1 (m = ns (nt-1)),
      jacobian   x -> J(x) as CSR with ONE fixed sparsity pattern (explicit zeros kept),
      noise, n_blocks."""
    nu_b = 0.01 / math.pi
    dt = 1.0 / (nt - 1)
    M, lumped, S, Adv = p1_periodic_line(ns)
    w = burgers(ns, nt, ic_noise, 0.0)                 # prior + initial-condition part only
    xs = np.arange(ns) / ns
    ic = np.sin(2 * np.pi * xs) + 0.5 * np.sin(4 * np.pi * xs + 0.3)
    x_prior = np.full(ns * nt, float(ic.mean()))
    M = M.tocsr(); S = S.tocsr(); Adv = Adv.tocsr()
    Jstat = (M + dt * nu_b * S).tocsr()
    pat_t = (abs(Jstat) + abs(Adv) + sp.identity(ns)).tocsr()
    pat_t.data[:] = 0.0                                # explicit zeros: the union pattern of a diagonal block of J
    negM = (-M).tocsr()

    def residual(x):
        X = x.reshape(nt, ns)
        out = np.empty((nt - 1, ns))
        for t in range(1, nt):
            u = X[t]
            out[t - 1] = M @ (u - X[t - 1]) + dt * (nu_b * (S @ u) + u * (Adv @ u))
        return out.ravel()

    def jacobian(x):
        X = x.reshape(nt, ns)
        rows = []
        for t in range(1, nt):
            u = X[t]
            Jt = (Jstat + dt * (sp.diags(u) @ Adv + sp.diags(Adv @ u)) + pat_t).tocsr()
            row = [None] * nt
            row[t - 1] = negM
            row[t] = Jt
            for k in range(nt):
                if row[k] is None:
                    row[k] = sp.csr_matrix((ns, ns))
            rows.append(row)
        J = sp.bmat(rows, format="csr")
        J.sort_indices()
        return J

    return {"Q": w.Q, "Qx_prior": w.rhs, "x_prior": x_prior, "residual": residual, "jacobian": jacobian,
            "noise": fem_noise, "n_blocks": nt, "n": ns * nt, "m": ns * (nt - 1)}


# --------------------------------------------------------------------------- analytic / toy

def laplace_kappa_grid(nx: int, ny: int, kappa2: float = 0.5) -> Workload:
    """5-point kappa^2 I + Delta_h on an nx x ny Dirichlet grid in row-block form:
    D_i = tridiag(-1, kappa^2+4, -1), B_i = -I, bs = nx.  Closed-form inverse via the
    discrete sine transform (SURVEY.md section 8c known-answer case (i))."""
    T = sp.diags([np.full(nx - 1, -1.0), np.full(nx, kappa2 + 4.0), np.full(nx - 1, -1.0)], [-1, 0, 1])
    E = sp.diags([np.full(ny - 1, -1.0), np.full(ny - 1, -1.0)], [-1, 1])
    Q = (sp.kron(sp.identity(ny), T) + sp.kron(E, sp.identity(nx))).tocsc()
    Q.sort_indices()
    rng = np.random.Generator(np.random.PCG64(7))
    rhs = rng.standard_normal(nx * ny)
    return Workload(f"laplace{nx}x{ny}", Q, rhs, ny, {"kappa2": kappa2, "nx": nx, "ny": ny})


def laplace_kappa_grid_variances(nx: int, ny: int, kappa2: float) -> np.ndarray:
    """Exact diag(Q^-1) of `laplace_kappa_grid` from the sine-transform eigen-decomposition."""
    i = np.arange(1, nx + 1)
    j = np.arange(1, ny + 1)
    lx = 2.0 - 2.0 * np.cos(np.pi * i / (nx + 1))
    ly = 2.0 - 2.0 * np.cos(np.pi * j / (ny + 1))
    lam = kappa2 + lx[:, None] + ly[None, :]
    sx = np.sqrt(2.0 / (nx + 1)) * np.sin(np.pi * np.outer(np.arange(1, nx + 1), i) / (nx + 1))
    sy = np.sqrt(2.0 / (ny + 1)) * np.sin(np.pi * np.outer(np.arange(1, ny + 1), j) / (ny + 1))
    var = np.einsum("xi,yj,ij->yx", sx ** 2, sy ** 2, 1.0 / lam)
    return var.ravel()


def ar1_chain_kron_identity(n_blocks: int, bs: int, phi: float = 0.5) -> Workload:
    """AR(1) chain (x) I_bs: bs independent scalar chains x_t = phi x_{t-1} + eps with a unit-variance
    start, written in block form D_1..D_{N-1} = (1 + phi^2) I except D_1 = I, D_N = I, B_i = -phi I
    (SURVEY.md section 8c known-answer case (ii)).  Closed form: L_i = I for i < N,
    L_N = sqrt(1 - phi^2) I, C_i = -phi I, logdet = bs log(1 - phi^2)."""
    d = np.full(n_blocks, 1.0 + phi * phi)
    d[0] = 1.0
    d[-1] = 1.0
    T = sp.diags([np.full(n_blocks - 1, -phi), d, np.full(n_blocks - 1, -phi)], [-1, 0, 1])
    Q = sp.kron(T, sp.identity(bs)).tocsc()
    Q.sort_indices()
    rng = np.random.Generator(np.random.PCG64(11))
    return Workload(f"ar1_{n_blocks}x{bs}", Q, rng.standard_normal(n_blocks * bs), n_blocks, {"phi": phi})


def random_block_tridiagonal(n_blocks: int, bs: int, seed: int = 0, density: float = 0.2,
                             shift: float = 2.0) -> Workload:
    """Random sparse SPD block-tridiagonal matrix (strictly block diagonally dominant)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_blocks * bs
    blocks = [[None] * n_blocks for _ in range(n_blocks)]
    for i in range(n_blocks):
        R = sp.random(bs, bs, density=density, random_state=rng, data_rvs=rng.standard_normal)
        blocks[i][i] = (R + R.T) * 0.5
        if i > 0:
            B = sp.random(bs, bs, density=density, random_state=rng, data_rvs=rng.standard_normal)
            blocks[i][i - 1] = B
            blocks[i - 1][i] = B.T
    A = sp.bmat(blocks, format="csr")
    rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
    A = (A + sp.diags(rowsum + shift)).tocsc()
    A = ((A + A.T) * 0.5).tocsc()
    A.sort_indices()
    rhs = rng.standard_normal(n)
    return Workload(f"rand{n_blocks}x{bs}", A, rhs, n_blocks, {"seed": seed})


CONFIGS = {
    # BASELINE.json configs (SURVEY.md section 8a sizes)
    "burgers512x64": lambda: burgers(512, 64),
    "darcy64": lambda: darcy(64),
    "darcy256": lambda: darcy(256),
    "elliptic512": lambda: elliptic(512),
    "burgers4096x512": lambda: burgers(4096, 512),
    # reduced sizes for CPU-side tests
    "darcy32": lambda: darcy(32),
    "darcy16": lambda: darcy(16),
    "elliptic32": lambda: elliptic(32),
    "burgers64x8": lambda: burgers(64, 8),
}


def make(name: str) -> Workload:
    return CONFIGS[name]()


def block_bandwidth_ok(Q: sp.spmatrix, n_blocks: int) -> bool:
    """True when every stored entry lies in the block tri-band of the partition."""
    n = Q.shape[0]
    bs = n // n_blocks
    coo = Q.tocoo()
    return bool(np.all(np.abs(coo.row // bs - coo.col // bs) <= 1))
