"""Host-side mirror of the reference's operator surface for the block-tridiagonal path.

Same names, argument meaning and error behaviour as
/root/reference/src/tridiagonal_cholesky.jl (`tridiagonal_cholesky` :65-82,
`forward_solve` :43-52, `backward_solve` :24-33, `ldiv!`/`ldiv` :54-63, `make_chunks`
:11-14, struct fields `N`, `chos`, `Cs` :5-9) and scripts/solve_burger.jl:182-254
(`extract_blocks`), with every numeric step done by libgmrf_hip.so on the GPU.  Julia is not
available in this image, so this Python layer stands where the Julia shim
(julia/DiffEqGMRFsHIP.jl) stands in a Julia session; both are thin wrappers over the same
C ABI (include/gmrf_hip.h).

Vectors / matrices may be NumPy arrays (host) or torch CUDA tensors (device resident; no
PCIe traffic).  Matrices of right-hand sides are n x k in column-major order (Julia
layout): pass `np.asfortranarray(B)` or a torch tensor of shape (k, n) via `.T`.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import _cabi
from ._cabi import GmrfError, NotPositiveDefinite  # noqa: F401  (re-exported)


def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "device")


def _colmajor(b, n: int):
    """Return (array_or_tensor, k, ld, was_1d) with column-major n x k storage."""
    if _is_torch(b):
        import torch
        if b.dtype != torch.float64:
            raise TypeError("float64 required")
        if b.dim() == 1:
            if b.shape[0] != n:
                raise ValueError("dimension mismatch")
            return b.contiguous(), 1, n, True
        # a (n, k) tensor whose transpose is contiguous is column-major
        if b.shape[0] != n:
            raise ValueError("dimension mismatch")
        bt = b.t()
        if not bt.is_contiguous():
            bt = bt.contiguous()
        return bt, b.shape[1], n, False          # bt is (k, n) row-major == n x k column-major
    a = np.asarray(b, dtype=np.float64)
    if a.ndim == 1:
        if a.shape[0] != n:
            raise ValueError("dimension mismatch")
        return np.ascontiguousarray(a), 1, n, True
    if a.shape[0] != n:
        raise ValueError("dimension mismatch")
    return np.asfortranarray(a), a.shape[1], n, False


def _alloc_like(ref, n: int, k: int, one_d: bool):
    if _is_torch(ref):
        import torch
        out = torch.empty((k, n), dtype=torch.float64, device=ref.device)
        return out, (out[0] if one_d else out.t())
    out = np.empty((n, k), dtype=np.float64, order="F")
    return out, (out[:, 0] if one_d else out)


class CsrMatrix:
    """Device-resident sparse matrix for `Q * x` (K6).  For a symmetric matrix the CSC arrays
    of a SparseMatrixCSC are the CSR arrays of the same matrix."""

    def __init__(self, A, device: int = 0, values_f32: bool = False, stream: int = 0):
        A = sp.csr_matrix(A)
        A.sort_indices()
        self.shape = A.shape
        self.nnz = int(A.nnz)
        self._h = C.c_void_p()
        rp = A.indptr.astype(np.int64)
        ci = A.indices.astype(np.int64)
        v = A.data.astype(np.float64)
        lib = _cabi.load()
        _cabi.check(lib.gmrf_csr_create(device, C.c_void_p(stream), A.shape[0], A.shape[1], _cabi.ptr(rp),
                                        _cabi.ptr(ci), _cabi.ptr(v), 0, int(values_f32), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _cabi.load().gmrf_csr_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def __matmul__(self, X):
        return self.matmul(X)

    def matmul_into(self, X, Y):
        """Stream-ordered Y = S X for torch CUDA tensors, written into `Y`, without synchronising (gmrf_spmm_async /
        gmrf_spmm_rows_async): X (n,) or C-contiguous (n, k) = node-major; Y of the matching shape."""
        if not (_is_torch(X) and _is_torch(Y) and X.is_cuda and Y.is_cuda and X.is_contiguous() and Y.is_contiguous()):
            raise TypeError("matmul_into takes contiguous torch CUDA tensors")
        lib = _cabi.load()
        if X.ndim == 1:
            _cabi.check(lib.gmrf_spmm_async(self._h, _cabi.ptr(X), _cabi.ptr(Y), 1, self.shape[1], self.shape[0]))
        else:
            k = X.shape[1]
            _cabi.check(lib.gmrf_spmm_rows_async(self._h, _cabi.ptr(X), _cabi.ptr(Y), k, k, k))
        return Y

    def matmul_rows(self, X):
        """Y = S X for X stored node-major: an (n, k) C-contiguous NumPy array or torch CUDA tensor (the k
        values of a node side by side).  This is the layout of the LDS-tiled SpMM kernel."""
        n = self.shape[1]
        if X.ndim != 2 or X.shape[0] != n:
            raise ValueError("dimension mismatch")
        k = X.shape[1]
        if _is_torch(X):
            import torch
            if X.dtype != torch.float64:
                raise TypeError("float64 required")
            X = X.contiguous()
            Y = torch.empty((self.shape[0], k), dtype=torch.float64, device=X.device)
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            Y = np.empty((self.shape[0], k), dtype=np.float64)
        _cabi.check(_cabi.load().gmrf_spmm_rows(self._h, _cabi.ptr(X), _cabi.ptr(Y), k, k, k))
        return Y

    def matmul(self, X):
        n = self.shape[1]
        # a row-major (n, k) operand already is node-major: no copy, LDS-tiled kernel
        if getattr(X, "ndim", 0) == 2 and X.shape[1] > 1 and X.shape[0] == n:
            row_major = X.is_contiguous() if _is_torch(X) else (X.flags.c_contiguous and not X.flags.f_contiguous)
            if row_major and (not _is_torch(X) or X.dtype.is_floating_point):
                return self.matmul_rows(X)
        xa, k, ld, one_d = _colmajor(X, n)
        store, view = _alloc_like(xa, self.shape[0], k, one_d)
        _cabi.check(_cabi.load().gmrf_spmm(self._h, _cabi.ptr(xa), _cabi.ptr(store), k, ld, self.shape[0]))
        return view


class PosteriorAssembler:
    """Device assembly of the Gauss-Newton / conditioning system:

        A   = Q + noise * J' * J                        scripts/solve_burger.jl:145
        rhs = base + noise * J' * (J * x + obs_diff)    scripts/solve_burger.jl:146

    for fixed sparsity patterns of Q (n x n, symmetric, both triangles) and J (m x n).  The
    symbolic phase runs once here; `precision()` / `rhs()` take the current values as NumPy arrays
    or torch CUDA tensors and return the same kind.  `pattern` is the CSC matrix (values 1) whose
    `.data` order `precision()` fills -- factor it once with `TridiagonalCholeskyFactor.factor`,
    then `refactor(values)`.  device = -1: symbolic only (no GPU needed)."""

    def __init__(self, Q, J, device: int = 0, stream: int = 0):
        Q = sp.csc_matrix(Q); Q.sort_indices()
        J = sp.csr_matrix(J); J.sort_indices()
        if Q.shape[0] != Q.shape[1] or J.shape[1] != Q.shape[0]:
            raise ValueError("Q must be n x n and J m x n")
        self.n, self.m = Q.shape[0], J.shape[0]
        self.nnz_q, self.nnz_j = int(Q.nnz), int(J.nnz)
        self._h = C.c_void_p()
        lib = _cabi.load()
        qp, qi = Q.indptr.astype(np.int64), Q.indices.astype(np.int64)
        jp, ji = J.indptr.astype(np.int64), J.indices.astype(np.int64)
        _cabi.check(lib.gmrf_assemble_create(device, C.c_void_p(stream), self.n, _cabi.ptr(qp), _cabi.ptr(qi), self.m,
                                             _cabi.ptr(jp), _cabi.ptr(ji), 0, C.byref(self._h)))
        nnz, nprod = C.c_int64(0), C.c_int64(0)
        _cabi.check(lib.gmrf_assemble_pattern(self._h, C.byref(nnz), C.byref(nprod), None, None, 0))
        self.nnz_out, self.n_products = int(nnz.value), int(nprod.value)
        cp, rv = np.empty(self.n + 1, dtype=np.int64), np.empty(self.nnz_out, dtype=np.int64)
        _cabi.check(lib.gmrf_assemble_pattern(self._h, None, None, _cabi.ptr(cp), _cabi.ptr(rv), 0))
        self.pattern = sp.csc_matrix((np.ones(self.nnz_out), rv, cp), shape=(self.n, self.n))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _cabi.load().gmrf_assemble_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def _like(ref, count):
        if _is_torch(ref):
            import torch
            return torch.empty(count, dtype=torch.float64, device=ref.device)
        return np.empty(count, dtype=np.float64)

    @staticmethod
    def _vec(v, count, name):
        if v is None:
            return None
        if _is_torch(v):
            v = v.contiguous()
        else:
            v = np.ascontiguousarray(v, dtype=np.float64)
        if (v.numel() if _is_torch(v) else v.size) != count:
            raise ValueError(f"{name}: expected {count} values")
        return v

    def precision(self, q_values, j_values, noise: float):
        q = self._vec(q_values, self.nnz_q, "q_values")
        j = self._vec(j_values, self.nnz_j, "j_values")
        out = self._like(j, self.nnz_out)
        _cabi.check(_cabi.load().gmrf_assemble_precision(self._h, _cabi.ptr(q), _cabi.ptr(j), float(noise), _cabi.ptr(out)))
        return out

    def rhs(self, base, j_values, x, obs_diff, noise: float):
        j = self._vec(j_values, self.nnz_j, "j_values")
        xv = self._vec(x, self.n, "x")
        b = self._vec(base, self.n, "base")
        o = self._vec(obs_diff, self.m, "obs_diff")
        out = self._like(j, self.n)
        _cabi.check(_cabi.load().gmrf_assemble_rhs(self._h, _cabi.ptr(b) if b is not None else None, _cabi.ptr(j),
                                                  _cabi.ptr(xv), _cabi.ptr(o) if o is not None else None, float(noise),
                                                  _cabi.ptr(out)))
        return out


class DarcyP1Assembler:
    """Darcy stiffness matrix and load vector on the device (first piece):
    `assemble_darcy_diff_matrix` of /root/reference/src/problems/darcy.jl:5-63 on the structured P1 mesh
    (nx x ny nodes, x fastest, quads cut by the diagonal n00 - n11, one quadrature point per cell), the
    coefficient looked up by nearest grid point (src/datasets/darcy.jl:30-34), Dirichlet rows / columns
    applied.  `pattern` is the CSR matrix (values 1) whose `.data` order `assemble()` fills -- the `J`
    of `PosteriorAssembler`.  device = -1: pattern only (no GPU needed).
    order = 2: the reference's own element (src/utils.jl:32-33) -- Lagrange{RefTriangle,2} with the 4-point rule of degree 3
    on the same nx x ny vertex mesh; dofs = the (2 nx - 1) x (2 ny - 1) lattice of vertices and edge midpoints."""

    def __init__(self, nx: int, ny: int, device: int = 0, stream: int = 0, order: int = 1):
        if order not in (1, 2):
            raise ValueError("order must be 1 (P1) or 2 (quadratic triangles)")
        self.nx, self.ny, self.order = int(nx), int(ny), int(order)
        self.n = int(nx) * int(ny) if order == 1 else (2 * int(nx) - 1) * (2 * int(ny) - 1)
        self._h = C.c_void_p()
        lib = _cabi.load()
        create = lib.gmrf_darcy_p1_create if order == 1 else lib.gmrf_darcy_p2_create
        _cabi.check(create(device, C.c_void_p(stream), nx, ny, C.byref(self._h)))
        nnz = C.c_int64(0)
        _cabi.check(lib.gmrf_darcy_p1_pattern(self._h, C.byref(nnz), None, None, 0))
        self.nnz = int(nnz.value)
        rp, ci = np.empty(self.n + 1, dtype=np.int64), np.empty(self.nnz, dtype=np.int64)
        _cabi.check(lib.gmrf_darcy_p1_pattern(self._h, None, _cabi.ptr(rp), _cabi.ptr(ci), 0))
        self.pattern = sp.csr_matrix((np.ones(self.nnz), ci, rp), shape=(self.n, self.n))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _cabi.load().gmrf_darcy_p1_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def assemble(self, coeff_table, beta: float = 1.0):
        """coeff_table: (ng, ng) array a[x index, y index] on the grid linspace(0, 1, ng)^2 (NumPy array or
        torch CUDA tensor).  Returns (values in `pattern.data` order, load vector f), same kind as the input."""
        if _is_torch(coeff_table):
            import torch
            tab = coeff_table.contiguous()
            vals = torch.empty(self.nnz, dtype=torch.float64, device=tab.device)
            f = torch.empty(self.n, dtype=torch.float64, device=tab.device)
        else:
            tab = np.ascontiguousarray(coeff_table, dtype=np.float64)
            vals, f = np.empty(self.nnz), np.empty(self.n)
        if tab.ndim != 2 or tab.shape[0] != tab.shape[1]:
            raise ValueError("coeff_table must be square (ng x ng)")
        _cabi.check(_cabi.load().gmrf_darcy_p1_assemble(self._h, _cabi.ptr(tab), tab.shape[0], float(beta), _cabi.ptr(vals),
                                                        _cabi.ptr(f)))
        return vals, f


class BurgersP1Tangent:
    """Residual and tangent of the implicit-Euler Burgers space-time system on the device (second
    piece): `f_and_J` of /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:118-149 with
    `assemble_burgers_advection_matrix` (src/problems/burgers.jl:5-59) per time slice and the static part of
    `assemble_burgers_mass_diffusion_matrices` (:60-98), on the periodic P1 line (ns nodes on [0,1), nt slices,
    time-major index) or, with order = 2, the quadratic periodic line.  `pattern` is the CSR matrix (values 1, (nt-1) ns x nt ns, 6 entries per row) whose `.data`
    order `tangent()` fills -- the `J` of `PosteriorAssembler`.  device = -1: pattern only (no GPU needed)."""

    def __init__(self, ns: int, nt: int, dt: float, nu: float, device: int = 0, stream: int = 0, order: int = 1):
        """order = 2: the quadratic periodic line of the reference's scripts (src/utils.jl:42-49): ns = 2 N_x dofs numbered by
        position, vertex rows of J with 10 entries, midpoint rows with 6 (gmrf_burgers_p2_create)."""
        self.ns, self.nt, self.dt, self.nu, self.order = int(ns), int(nt), float(dt), float(nu), int(order)
        self.rows, self.n = (self.nt - 1) * self.ns, self.nt * self.ns
        self._h = C.c_void_p()
        lib = _cabi.load()
        create = lib.gmrf_burgers_p2_create if self.order == 2 else lib.gmrf_burgers_p1_create
        _cabi.check(create(device, C.c_void_p(stream), ns, nt, float(dt), float(nu), C.byref(self._h)))
        nnz = C.c_int64(0)
        _cabi.check(lib.gmrf_burgers_p1_pattern(self._h, C.byref(nnz), None, None, 0))
        self.nnz = int(nnz.value)
        rp, ci = np.empty(self.rows + 1, dtype=np.int64), np.empty(self.nnz, dtype=np.int64)
        _cabi.check(lib.gmrf_burgers_p1_pattern(self._h, None, _cabi.ptr(rp), _cabi.ptr(ci), 0))
        self.pattern = sp.csr_matrix((np.ones(self.nnz), ci, rp), shape=(self.rows, self.n))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _cabi.load().gmrf_burgers_p1_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def tangent(self, w):
        """w: (nt ns,) linearisation point (NumPy array or torch CUDA tensor).  Returns (J values in `pattern.data`
        order, residual f(w)), same kind as the input."""
        if _is_torch(w):
            import torch
            wv = w.contiguous()
            vals = torch.empty(self.nnz, dtype=torch.float64, device=wv.device)
            f = torch.empty(self.rows, dtype=torch.float64, device=wv.device)
        else:
            wv = np.ascontiguousarray(w, dtype=np.float64)
            vals, f = np.empty(self.nnz), np.empty(self.rows)
        if wv.ndim != 1 or wv.shape[0] != self.n:
            raise ValueError(f"w must have {self.n} entries")
        _cabi.check(_cabi.load().gmrf_burgers_p1_tangent(self._h, _cabi.ptr(wv), _cabi.ptr(vals), _cabi.ptr(f)))
        return vals, f


class ShallowWaterP1:
    """Element kernels of the linear shallow-water SPDE on the device (third piece):
    `assemble_system!` of /root/reference/src/spdes/shallow_water.jl:17-122 (coupling K, element-lumped mass M, stiffness S
    of the three fields h, u, v) and the per-step operators of `discretize` (:170-217), on the structured P1 triangle mesh
    (nx x ny nodes, x fastest, quads cut by the diagonal n00 - n11; dof = 3 * node + field; symmetric 3-point quadrature).
    `pattern_K` / `pattern_S` are CSR matrices (values 1) whose `.data` order `assemble()` / `operators()` fill;
    `qpoints` (cells, 3, 2) are the quadrature points at which the caller evaluates its H(x).  device = -1: patterns and
    quadrature points only (no GPU needed)."""

    def __init__(self, nx: int, ny: int, device: int = 0, stream: int = 0):
        self.nx, self.ny, self.nn, self.n = int(nx), int(ny), int(nx) * int(ny), 3 * int(nx) * int(ny)
        self.cells = 2 * (self.nx - 1) * (self.ny - 1)
        self._h = C.c_void_p()
        lib = _cabi.load()
        _cabi.check(lib.gmrf_shallow_water_p1_create(device, C.c_void_p(stream), nx, ny, C.byref(self._h)))
        pats = []
        for which in (0, 1):
            nnz = C.c_int64(0)
            _cabi.check(lib.gmrf_shallow_water_p1_pattern(self._h, which, C.byref(nnz), None, None, 0))
            rp, ci = np.empty(self.n + 1, dtype=np.int64), np.empty(nnz.value, dtype=np.int64)
            _cabi.check(lib.gmrf_shallow_water_p1_pattern(self._h, which, None, _cabi.ptr(rp), _cabi.ptr(ci), 0))
            pats.append(sp.csr_matrix((np.ones(nnz.value), ci, rp), shape=(self.n, self.n)))
        self.pattern_K, self.pattern_S = pats
        self.qpoints = np.empty((self.cells, 3, 2), dtype=np.float64)
        _cabi.check(lib.gmrf_shallow_water_p1_qpoints(self._h, _cabi.ptr(self.qpoints)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _cabi.load().gmrf_shallow_water_p1_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def _mask(prescribed, n):
        if prescribed is None:
            return None
        if _is_torch(prescribed):
            import torch
            m = prescribed.to(torch.uint8).contiguous()
            if m.numel() != n:
                raise ValueError(f"prescribed: expected {n} entries")
            return m
        m = np.ascontiguousarray(np.asarray(prescribed) != 0, dtype=np.uint8)
        if m.size != n:
            raise ValueError(f"prescribed: expected {n} entries")
        return m

    def _out(self, ref, count):
        if _is_torch(ref):
            import torch
            return torch.empty(count, dtype=torch.float64, device=ref.device)
        return np.empty(count, dtype=np.float64)

    def assemble(self, H_q, k: float = 0.0, f: float = 0.0, g: float = 9.81, prescribed=None):
        """H_q: (cells, 3) values of H at `qpoints` (NumPy array or torch CUDA tensor).  Returns (K values, lumped M, S values),
        same kind as H_q, with `apply!` of the prescribed dofs done."""
        hq = H_q.contiguous() if _is_torch(H_q) else np.ascontiguousarray(H_q, dtype=np.float64)
        if (hq.numel() if _is_torch(hq) else hq.size) != self.cells * 3:
            raise ValueError(f"H_q must have {self.cells} x 3 entries")
        kv, ml, sv = self._out(hq, self.pattern_K.nnz), self._out(hq, self.n), self._out(hq, self.pattern_S.nnz)
        m = self._mask(prescribed, self.n)
        _cabi.check(_cabi.load().gmrf_shallow_water_p1_assemble(self._h, _cabi.ptr(hq), float(k), float(f), float(g), _cabi.ptr(m),
                                                                _cabi.ptr(kv), _cabi.ptr(ml), _cabi.ptr(sv)))
        return kv, ml, sv

    def operators(self, K_vals, M_lumped, S_vals, prescribed=None, kappa_matern: float = 1.0, tau: float = 1.0, dt: float = 1.0):
        """The operators of one time step (`discretize`, :170-217): dict with G_dt (values in pattern_K: M~ + dt K, constraints
        applied), J (values in pattern_S: sqrt(ratio) M~^-1/2 (kappa^2 M~ + G); Q_matern = J'J), M_tilde, beta."""
        g_v, j_v = self._out(K_vals, self.pattern_K.nnz), self._out(K_vals, self.pattern_S.nnz)
        mt, be = self._out(K_vals, self.n), self._out(K_vals, self.n)
        m = self._mask(prescribed, self.n)
        _cabi.check(_cabi.load().gmrf_shallow_water_p1_operators(self._h, _cabi.ptr(K_vals), _cabi.ptr(M_lumped), _cabi.ptr(S_vals),
                                                                 _cabi.ptr(m), float(kappa_matern), float(tau), float(dt),
                                                                 _cabi.ptr(g_v), _cabi.ptr(j_v), _cabi.ptr(mt), _cabi.ptr(be)))
        return {"G_dt": g_v, "J": j_v, "M_tilde": mt, "beta": be}


def gn_step(F: "TridiagonalCholeskyFactor", asm: PosteriorAssembler, q_values, Qx_prior, j_values, x, obs_diff,
            noise: float):
    """One Gauss-Newton step of scripts/solve_burger.jl:143-149 with everything resident on the
    device: assemble A = Q + noise J'J and the right-hand side, re-factor on the analysed pattern
    (`F` must have been factored once on `asm.pattern`), solve."""
    a_vals = asm.precision(q_values, j_values, noise)
    rhs = asm.rhs(Qx_prior, j_values, x, obs_diff, noise)
    F.refactor(a_vals)
    return ldiv(F, rhs)


class ConditionedGMRF:
    """What `condition_on_observations(x, A, Q_eps, y; solver_blueprint)` returns in the reference's
    problem loop (scripts/darcy/solve_darcy_gmrf-fem.jl:176-192), served by the block-tridiagonal
    path: posterior precision Q + Q_eps A'A (assembled on the device), then `mean`, `rand`, `std`,
    `logdet`.  One object per sparsity pattern; `update(a_values, y)` re-conditions on a new
    observation matrix with the same pattern (the next problem of the data set) -- values only."""

    def __init__(self, Q, mu, A, q_eps: float, y, n_blocks: int, device: int = 0):
        Q = sp.csc_matrix(Q); Q.sort_indices()
        A = sp.csr_matrix(A); A.sort_indices()
        self.Q, self.q_eps, self.n_blocks = Q, float(q_eps), int(n_blocks)
        self.mu = np.zeros(Q.shape[0]) if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        self._qmu = Q @ self.mu
        self.asm = PosteriorAssembler(Q, A, device=device)
        self.F = TridiagonalCholeskyFactor(device=device)
        self._analysed = False
        self.update(A.data, y)

    def update(self, a_values, y):
        """Re-condition with new values of A (same pattern) and new observations y."""
        self._vals = np.asarray(self.asm.precision(self.Q.data, a_values, self.q_eps))
        if not self._analysed:
            self.F.factor(self.precision_matrix(), self.n_blocks)      # symbolic analysis + first factor
            self._analysed = True
        else:
            self.F.refactor(self._vals)
        # information vector Q mu + Q_eps A' y  (gmrf_assemble_rhs with x = 0, obs_diff = y)
        self._rhs = self.asm.rhs(self._qmu, a_values, np.zeros(self.asm.n), y, self.q_eps)
        self._mean = None
        self._csr = None
        return self

    def precision_matrix(self):
        """Posterior precision as a SciPy CSC matrix (`to_matrix(precision_map(x_cond))`)."""
        P = self.asm.pattern.copy()
        P.data = self._vals.copy()
        return P

    def mean(self):
        if self._mean is None:
            self._mean = ldiv(self.F, self._rhs)
        return self._mean

    def rand(self, k: int = 1, seed: int = 0x5EED, first_id: int = 0):
        """n x k samples  mean + L^-T z  with device Philox normals (`rand(rng, x_cond)`)."""
        return self.F.sample(k, mean=self.mean(), seed=seed, first_id=first_id)

    def var(self, method: str = "exact", k: int = 50, seed: int = 0x5EED):
        """Marginal variances: "exact" selected inversion, or "rbmc" = the reference's RBMCStrategy(k)."""
        if method == "exact":
            return self.F.marginal_var("exact")
        if self._csr is None:
            self._csr = CsrMatrix(self.precision_matrix())
        return self.F.marginal_var(method, k=k, seed=seed, Q=self._csr)

    def std(self, method: str = "exact", k: int = 50, seed: int = 0x5EED):
        return np.sqrt(self.var(method, k, seed))

    def logdet(self) -> float:
        return self.F.logdet()

    def sqmahal(self, z) -> float:
        """`sqmahal(x_cond, z)` = (z - mean)' Q_post (z - mean) (scripts/burgers/solve_burgers_gmrf-collocation.jl:262):
        one CSR SpMV of the posterior precision (K6) and a dot product."""
        if self._csr is None:
            self._csr = CsrMatrix(self.precision_matrix())
        d = np.ascontiguousarray(np.asarray(z, dtype=np.float64) - np.asarray(self.mean()))
        return float(d @ np.asarray(self._csr @ d))

    def nll(self, z) -> float:
        """Negative log-likelihood of z under the conditioned GMRF, `nll_soln` of the same script (:213-215):
        0.5 (n log 2 pi + sqmahal + logdet Sigma), logdet Sigma = -logdet Q_post = -2 sum log diag L (:208-211)."""
        n = self.asm.n
        return 0.5 * (n * np.log(2.0 * np.pi) + self.sqmahal(z) - self.logdet())


def condition_on_observations(Q, mu, A, q_eps: float, y, n_blocks: int, device: int = 0) -> ConditionedGMRF:
    """Python twin of the reference's `condition_on_observations(x, A, Q_eps, y)` for a GMRF
    N(mu, Q^-1) whose posterior precision is block tridiagonal with `n_blocks` blocks."""
    return ConditionedGMRF(Q, mu, A, q_eps, y, n_blocks, device)


class _LazyBlocks(Sequence):
    """`F.chos` / `F.Cs`: dense blocks copied from the device on access."""

    def __init__(self, owner: "TridiagonalCholeskyFactor", kind: int, count: int):
        self._o, self._kind, self._count = owner, kind, count

    def __len__(self):
        return self._count

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._count))]
        if i < 0:
            i += self._count
        if not 0 <= i < self._count:
            raise IndexError(i)
        return self._o.get_block(self._kind, i)


class TridiagonalCholeskyFactor:
    """Handle wrapper of the device-resident factor (struct at src/tridiagonal_cholesky.jl:5-9).

    `N` is the total size n (as in the reference, NOT the block count); `chos[i]` is the
    lower-triangular L_i, `Cs[i]` = L_{i+1,i}.
    """

    def __init__(self, device: int = 0, stream: int = 0, batch: int = 1):
        self._h = C.c_void_p()
        self._lib = _cabi.load()
        _cabi.check(self._lib.gmrf_bt_create(device, C.c_void_p(stream), C.byref(self._h)))
        self.batch = 1
        if batch != 1:
            self.set_batch(batch)
        self.N = 0
        self.n_blocks = 0
        self.block_size = 0
        self.device = device
        self._pattern = None

    # -- life cycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gmrf_bt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batch of independent problems on one sparsity pattern
    def set_batch(self, batch: int):
        _cabi.check(self._lib.gmrf_bt_set_batch(self._h, int(batch)))
        self.batch = int(batch)

    def select_problem(self, p: int):
        _cabi.check(self._lib.gmrf_bt_select_problem(self._h, int(p)))

    def solve_batch(self, b, mode: int = _cabi.SOLVE_FULL):
        """b: (B, k, n) C-contiguous NumPy array / torch CUDA tensor (each right-hand side
        contiguous) -> same shape.  Problem p is solved with factor p."""
        B, k, n = b.shape
        if B != self.batch or n != self.N:
            raise ValueError("expected shape (batch, k, n)")
        if _is_torch(b):
            import torch
            b = b.contiguous()
            out = torch.empty_like(b)
        else:
            b = np.ascontiguousarray(b, dtype=np.float64)
            out = np.empty_like(b)
        _cabi.check(self._lib.gmrf_bt_solve(self._h, _cabi.ptr(b), _cabi.ptr(out), k, n, n, mode))
        return out

    def sample_batch(self, k: int, mean=None, seed: int = 0x5EED, first_id: int = 0, like=None):
        """(B, k, n) samples: problem p, sample s = mean[p] + L_p^-T z(id = first_id + p*k + s)."""
        B, n = self.batch, self.N
        ref = like if like is not None else mean
        if _is_torch(ref):
            import torch
            out = torch.empty((B, k, n), dtype=torch.float64, device=ref.device)
            if mean is not None:
                mean = mean.contiguous()
        else:
            out = np.empty((B, k, n), dtype=np.float64)
            if mean is not None:
                mean = np.ascontiguousarray(mean, dtype=np.float64)
        _cabi.check(self._lib.gmrf_bt_sample(self._h, seed, first_id, k, _cabi.ptr(mean), None, _cabi.ptr(out), n))
        return out

    # -- factorisation
    def factor(self, A, N_blocks: int, values=None):
        """`values`: (batch, nnz) array of CSC values for a batch of problems on the pattern of A."""
        A = sp.csc_matrix(A)
        if A.shape[0] != A.shape[1]:
            raise ValueError("matrix must be square")
        if not A.has_sorted_indices:
            A = A.copy()
            A.sort_indices()
        n = A.shape[0]
        colptr = A.indptr.astype(np.int64)
        rowval = A.indices.astype(np.int64)
        nz = np.ascontiguousarray(A.data if values is None else values, dtype=np.float64)
        if nz.size != self.batch * A.nnz:
            raise ValueError("values must hold batch * nnz entries")
        info = C.c_int32(0)
        st = self._lib.gmrf_bt_factor_csc(self._h, n, int(N_blocks), _cabi.ptr(colptr), _cabi.ptr(rowval),
                                          _cabi.ptr(nz), 0, C.byref(info))
        _cabi.check(st, info.value)
        self._set_shape(n, int(N_blocks))
        self._pattern = (colptr, rowval)
        return self

    def factor_blocks(self, diag_blocks, off_diag_blocks):
        """Factor from the output of `extract_blocks` (lists of sparse bs x bs blocks)."""
        nb = len(diag_blocks)
        bs = diag_blocks[0].shape[0]
        keep = []

        def mk(blocks):
            arr = (_cabi.SparseBlock * max(len(blocks), 1))()
            for i, b in enumerate(blocks):
                b = sp.csc_matrix(b)
                b.sort_indices()
                p = b.indptr.astype(np.int64); ix = b.indices.astype(np.int64); v = b.data.astype(np.float64)
                keep.extend([p, ix, v])
                arr[i].nnz = b.nnz
                arr[i].ptr = p.ctypes.data; arr[i].idx = ix.ctypes.data; arr[i].val = v.ctypes.data
            return arr

        d = mk(diag_blocks)
        lo = mk(off_diag_blocks)
        info = C.c_int32(0)
        st = self._lib.gmrf_bt_factor_blocks(self._h, nb * bs, nb, C.cast(d, C.c_void_p), C.cast(lo, C.c_void_p),
                                             0, 1, C.byref(info))
        _cabi.check(st, info.value)
        self._set_shape(nb * bs, nb)
        return self

    def refactor(self, nzval):
        """New values on the sparsity pattern of the last `factor` (nzval in CSC order;
        NumPy array or torch CUDA tensor)."""
        info = C.c_int32(0)
        if not _is_torch(nzval):
            nzval = np.ascontiguousarray(nzval, dtype=np.float64)
        st = self._lib.gmrf_bt_refactor_values(self._h, _cabi.ptr(nzval), C.byref(info))
        _cabi.check(st, info.value)
        return self

    def _set_shape(self, n, nb):
        self.N = n
        self.n_blocks = nb
        self.block_size = n // nb
        self.chos = _LazyBlocks(self, _cabi.BLOCK_L, nb)
        self.Cs = _LazyBlocks(self, _cabi.BLOCK_C, nb - 1)
        self.inverses = _LazyBlocks(self, _cabi.BLOCK_LINV, nb)

    # -- solves
    def _solve(self, b, mode: int, out=None):
        xa, k, ld, one_d = _colmajor(b, self.N)
        if out is None:
            store, view = _alloc_like(xa, self.N, k, one_d)
        else:
            store, view = out, out
        _cabi.check(self._lib.gmrf_bt_solve(self._h, _cabi.ptr(xa), _cabi.ptr(store), k, ld, self.N, mode))
        return view

    def get_block(self, kind: int, i: int) -> np.ndarray:
        bs = self.block_size
        out = np.empty((bs, bs), dtype=np.float64, order="F")
        _cabi.check(self._lib.gmrf_bt_get_block(self._h, kind, i, _cabi.ptr(out), bs))
        return out

    def logdet(self) -> float:
        v = C.c_double(0.0)
        _cabi.check(self._lib.gmrf_bt_logdet(self._h, C.byref(v)))
        return v.value

    def normals(self, k: int, seed: int = 0x5EED, first_id: int = 0, like=None):
        if self.batch != 1:
            raise ValueError("a batched handle draws (batch, k, n) normals: normals_batch")
        store, view = _alloc_like(like if like is not None else np.empty(0), self.N, k, False)
        _cabi.check(self._lib.gmrf_bt_normals(self._h, seed, first_id, k, _cabi.ptr(store), self.N))
        return view

    def normals_batch(self, k: int, seed: int = 0x5EED, first_id: int = 0):
        """(B, k, n) host array of the N(0,1) draws `sample_batch(k, seed=, first_id=)` and the batched variance
        estimators use: problem p, draw s = Philox id first_id + p*k + s."""
        out = np.empty((self.batch, k, self.N), dtype=np.float64)
        _cabi.check(self._lib.gmrf_bt_normals(self._h, seed, first_id, k, _cabi.ptr(out), self.N))
        return out

    def sample(self, k: int, mean=None, z=None, seed: int = 0x5EED, first_id: int = 0, like=None):
        """k posterior samples mean + L^-T z as an n x k matrix (rand(rng, x_cond))."""
        ref = like if like is not None else (z if z is not None else (mean if mean is not None else np.empty(0)))
        za = None
        if z is not None:
            za, kz, _, _ = _colmajor(z, self.N)
            k = kz
        if mean is not None and not _is_torch(mean):
            mean = np.ascontiguousarray(mean, dtype=np.float64)
        store, view = _alloc_like(ref if _is_torch(ref) else np.empty(0), self.N, k, False)
        _cabi.check(self._lib.gmrf_bt_sample(self._h, seed, first_id, k, _cabi.ptr(mean), _cabi.ptr(za),
                                             _cabi.ptr(store), self.N))
        return view

    def posterior(self, b, k: int, seed: int = 0x5EED, first_id: int = 0):
        """(mean, samples) = (A^-1 b, mean + L^-T z) in ONE call (mean(x_cond) and rand(rng, x_cond) of the reference's script,
        scripts/darcy/solve_darcy_gmrf-fem.jl:190-191): bitwise `ldiv(F, b)` and `F.sample(k, mean=...)`; on device tensors the
        samples' sweep runs beside the mean's two where this handle's sweeps are persistent launches (gmrf_bt_posterior).
        One problem (batch == 1); b: n values; samples come back n x k like `sample`."""
        if self.batch != 1:
            raise ValueError("posterior(): one problem per handle (use solve_batch / sample_batch for batches)")
        xa, kb, ld, one_d = _colmajor(b, self.N)
        if kb != 1:
            raise ValueError("posterior(): one right-hand side")
        mstore, mview = _alloc_like(xa, self.N, 1, True)
        sstore, sview = _alloc_like(xa, self.N, k, False)
        _cabi.check(self._lib.gmrf_bt_posterior(self._h, _cabi.ptr(xa), seed, first_id, k, _cabi.ptr(mstore), _cabi.ptr(sstore),
                                                self.N))
        return mview, sview

    def marginal_var(self, method: str = "exact", k: int = 50, seed: int = 0x5EED, Q: Optional[CsrMatrix] = None,
                     q_values=None, out=None):
        """diag(Q^-1).  "exact" (selected inversion), "rbmc" (the reference's RBMCStrategy(k); needs Q)
        or "mc".  A batch returns (batch, n); its sampled estimators take Q for the pattern and
        `q_values` (batch, nnz): every problem's values in Q's CSR order (for a symmetric matrix the
        nzval arrays the factor was given).  `out`: a contiguous float64 array of that shape to fill -- NumPy, or a
        torch tensor on the handle's device (the variances then never leave the device); default: a new NumPy array."""
        m = {"exact": _cabi.VAR_EXACT, "rbmc": _cabi.VAR_RBMC, "mc": _cabi.VAR_MC}[method]
        shape = (self.N,) if self.batch == 1 else (self.batch, self.N)
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        else:
            ok = tuple(out.shape) == shape and (out.is_contiguous() and str(out.dtype) == "torch.float64" if _is_torch(out)
                                                else out.flags.c_contiguous and out.dtype == np.float64)
            if not ok:
                raise ValueError(f"out must be a contiguous float64 array of shape {shape}")
        if self.batch > 1 and m != _cabi.VAR_EXACT:
            qv = None
            if q_values is not None:
                qv = q_values.contiguous() if _is_torch(q_values) else np.ascontiguousarray(q_values, dtype=np.float64)
            _cabi.check(self._lib.gmrf_bt_marginal_var_batch(self._h, m, k, seed, Q._h if Q is not None else None,
                                                             _cabi.ptr(qv) if qv is not None else None, _cabi.ptr(out)))
            return out
        _cabi.check(self._lib.gmrf_bt_marginal_var(self._h, m, k, seed, Q._h if Q is not None else None,
                                                   _cabi.ptr(out)))
        return out

    def var_accumulate(self, acc, method: str, first_id: int, k: int, seed: int = 0x5EED,
                       Q: Optional[CsrMatrix] = None):
        m = {"rbmc": _cabi.VAR_RBMC, "mc": _cabi.VAR_MC}[method]
        _cabi.check(self._lib.gmrf_bt_var_accumulate(self._h, m, first_id, k, seed,
                                                     Q._h if Q is not None else None, _cabi.ptr(acc)))
        return acc

    def stats(self) -> dict:
        s = _cabi.Stats()
        _cabi.check(self._lib.gmrf_bt_stats(self._h, C.byref(s)))
        out = {}
        for f, _ in s._fields_:
            v = getattr(s, f)
            out[f] = list(v) if hasattr(v, "__len__") else v
        return out

    def gemm_shapes(self) -> list:
        """GEMM launches of the profiled steps by shape (test hook gmrf_test_gemm_shapes): list of dicts."""
        n = C.c_int64(0)
        _cabi.check(self._lib.gmrf_test_gemm_shapes(self._h, None, 0, C.byref(n)))
        buf = np.zeros((max(int(n.value), 1), 11))
        _cabi.check(self._lib.gmrf_test_gemm_shapes(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.shape[0], C.byref(n)))
        names = ("class", "M", "N", "K", "tri", "lower_only", "problems", "k_bounds")
        return [dict({k: int(r[i]) for i, k in enumerate(names)}, launches=int(r[8]), ms=float(r[9]), flops=float(r[10]))
                for r in buf[: int(n.value)]]

    def set_profiling(self, level: int):
        _cabi.check(self._lib.gmrf_bt_set_profiling(self._h, level))

    def set_eager(self, eager: bool):
        _cabi.check(self._lib.gmrf_bt_set_eager(self._h, int(eager)))

    def synchronize(self):
        _cabi.check(self._lib.gmrf_bt_synchronize(self._h))

    def factor_buffer(self, kind: int) -> Tuple[int, int]:
        p = C.c_void_p()
        nbytes = C.c_int64(0)
        _cabi.check(self._lib.gmrf_bt_factor_buffer(self._h, kind, C.byref(p), C.byref(nbytes)))
        return int(p.value or 0), int(nbytes.value)

    def export_factor(self) -> np.ndarray:
        """Flat host image (uint8) of the selected problem's factor: header + L, C, Linv blocks."""
        nbytes = C.c_int64(0)
        _cabi.check(self._lib.gmrf_bt_export_size(self._h, C.byref(nbytes)))
        buf = np.empty(nbytes.value, dtype=np.uint8)
        _cabi.check(self._lib.gmrf_bt_export_factor(self._h, _cabi.ptr(buf), nbytes.value))
        return buf

    def import_factor(self, image: np.ndarray):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        _cabi.check(self._lib.gmrf_bt_import_factor(self._h, _cabi.ptr(image), image.size))
        hdr = image[:64].view(np.int64)
        self._set_shape(int(hdr[2]), int(hdr[3]))
        return self

    def adopt_shape(self, n: int, n_blocks: int):
        _cabi.check(self._lib.gmrf_bt_adopt_shape(self._h, n, n_blocks))
        self._set_shape(n, n_blocks)

    def get_layout(self) -> np.ndarray:
        """Layout record of the stored factor: [cmin, rmax, n_row_tiles, kst..., split p of the block inverses (0: full)] (int64)."""
        cnt = C.c_int64(0)
        _cabi.check(self._lib.gmrf_bt_get_layout(self._h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, dtype=np.int64)
        _cabi.check(self._lib.gmrf_bt_get_layout(self._h, _cabi.ptr(out), out.size, C.byref(cnt)))
        return out

    def adopt_layout(self, n: int, n_blocks: int, layout=None):
        """A rank that receives the factor: shape + the root's layout record, storage without factoring."""
        if layout is None:
            _cabi.check(self._lib.gmrf_bt_adopt_layout(self._h, n, n_blocks, None, 0))
        else:
            lay = np.ascontiguousarray(layout, dtype=np.int64)
            _cabi.check(self._lib.gmrf_bt_adopt_layout(self._h, n, n_blocks, _cabi.ptr(lay), lay.size))
        self._set_shape(n, n_blocks)

    def adopt_commit(self, l_blocks_valid: bool = False):
        _cabi.check(self._lib.gmrf_bt_adopt_commit(self._h, int(l_blocks_valid)))

    def set_keep_l(self, keep: bool):
        """keep = False: the blocks chos[i].L are not retained (sweeps, samples, variances and logdet do
        not need them): 1.6 -> 0.84 GB per darcy256 posterior; `F.chos` then raises."""
        _cabi.check(self._lib.gmrf_bt_set_keep_l(self._h, int(keep)))

    def block_range(self, kind: int, i0: int, i1: int):
        """(first element, element count, problem stride) of blocks [i0, i1) inside a factor buffer."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _cabi.check(self._lib.gmrf_bt_block_range(self._h, kind, i0, i1, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def packed_size(self, i0: int, i1: int) -> int:
        """Doubles per problem of the packed transport image of blocks [i0, i1) (gmrf_bt_packed_size)."""
        v = C.c_int64(0)
        _cabi.check(self._lib.gmrf_bt_packed_size(self._h, i0, i1, C.byref(v)))
        return v.value

    def pack_blocks_async(self, i0: int, i1: int, dev_buf):
        """Blocks [i0, i1) of every problem -> dev_buf ((batch, packed_size) torch CUDA tensor), stream-ordered."""
        _cabi.check(self._lib.gmrf_bt_pack_blocks_async(self._h, i0, i1, _cabi.ptr(dev_buf)))

    def unpack_blocks_async(self, i0: int, i1: int, dev_buf):
        _cabi.check(self._lib.gmrf_bt_unpack_blocks_async(self._h, i0, i1, _cabi.ptr(dev_buf)))

    def factor_begin(self, A, N_blocks: int):
        A = sp.csc_matrix(A)
        n = A.shape[0]
        colptr = A.indptr.astype(np.int64); rowval = A.indices.astype(np.int64)
        nz = np.ascontiguousarray(A.data, dtype=np.float64)
        _cabi.check(self._lib.gmrf_bt_factor_begin_csc(self._h, n, int(N_blocks), _cabi.ptr(colptr),
                                                       _cabi.ptr(rowval), _cabi.ptr(nz), 0))
        self._set_shape(n, int(N_blocks))

    def factor_begin_values(self, nzval):
        """Start a pipelined re-factorisation on the analysed pattern (nzval host or device)."""
        if not _is_torch(nzval):
            nzval = np.ascontiguousarray(nzval, dtype=np.float64)
        _cabi.check(self._lib.gmrf_bt_factor_begin_csc(self._h, self.N, self.n_blocks, None, None,
                                                       _cabi.ptr(nzval), 0))

    def factor_step_async(self, i0: int, i1: int):
        _cabi.check(self._lib.gmrf_bt_factor_step_async(self._h, i0, i1))

    def factor_end(self):
        info = C.c_int32(0)
        st = self._lib.gmrf_bt_factor_end(self._h, C.byref(info))
        _cabi.check(st, info.value)


# ----------------------------------------------------------------------------- reference surface

def tridiagonal_cholesky(A, N_blocks: int, device: int = 0, stream: int = 0) -> TridiagonalCholeskyFactor:
    """tridiagonal_cholesky(A::SparseMatrixCSC, N_blocks)  (src/tridiagonal_cholesky.jl:65-82).

    Raises NotPositiveDefinite (PosDefException) with the failing block index, ValueError
    when size(A,1) is not a multiple of N_blocks (the reference would silently mis-factor),
    GmrfError(ERR_BAND) when entries lie outside the block tri-band (the reference silently
    ignores them)."""
    n = A.shape[0]
    if N_blocks <= 0 or n % N_blocks != 0:
        raise ValueError("size(A,1) must be a positive multiple of N_blocks")
    return TridiagonalCholeskyFactor(device, stream).factor(A, N_blocks)


def forward_solve(L: TridiagonalCholeskyFactor, b):
    """y = L^-1 b  (src/tridiagonal_cholesky.jl:43-52); returns the flat vector/matrix."""
    return L._solve(b, _cabi.SOLVE_FORWARD)


def backward_solve(L: TridiagonalCholeskyFactor, b):
    """x = L^-T b  (src/tridiagonal_cholesky.jl:24-33)."""
    return L._solve(b, _cabi.SOLVE_BACKWARD)


def ldiv(L: TridiagonalCholeskyFactor, b):
    """ldiv(L, b) = A^-1 b  (src/tridiagonal_cholesky.jl:60-63)."""
    return L._solve(b, _cabi.SOLVE_FULL)


def ldiv_(y, L: TridiagonalCholeskyFactor, b):
    """ldiv!(y, L, b)  (src/tridiagonal_cholesky.jl:54-58): result written into y, which is
    returned.  y may alias b."""
    res = L._solve(b, _cabi.SOLVE_FULL)
    if _is_torch(y):
        y.copy_(res)
    else:
        y[...] = res
    return y


def make_chunks(X, n: int):
    """src/tridiagonal_cholesky.jl:11-14: n contiguous views, the last takes the remainder."""
    c = X.shape[0] // n
    return [X[c * k:(X.shape[0] if k == n - 1 else c * k + c)] for k in range(n)]


def extract_blocks(I, J, V, block_size: int):
    """scripts/solve_burger.jl:182-254 on 1-based COO triplets: (diag_blocks, off_diag_blocks),
    the lower off-diagonal convention, entries outside the tri-band dropped as the reference
    does.  Vectorised host code (no per-entry Python loop)."""
    I = np.asarray(I, dtype=np.int64) - 1
    J = np.asarray(J, dtype=np.int64) - 1
    V = np.asarray(V)
    if I.size == 0:
        return [], []
    bi, bj = I // block_size, J // block_size
    nb = int(bi.max()) + 1
    diag, off = [], []
    is_d = bi == bj
    is_o = bi == bj + 1
    order_d = np.flatnonzero(is_d)
    order_o = np.flatnonzero(is_o)
    for b in range(nb):
        sel = order_d[bi[order_d] == b]
        diag.append(sp.coo_matrix((V[sel], (I[sel] - b * block_size, J[sel] - b * block_size)),
                                  shape=(block_size, block_size)).tocsc())
        if b > 0:
            sel = order_o[bi[order_o] == b]
            off.append(sp.coo_matrix((V[sel], (I[sel] - b * block_size, J[sel] - (b - 1) * block_size)),
                                     shape=(block_size, block_size)).tocsc())
    return diag, off


def logdet(L: TridiagonalCholeskyFactor) -> float:
    return L.logdet()


class StreamSet:
    """n HIP streams on hardware queues of their own (gmrf_streams_create): one per handle that is driven side by side
    with others.  `.pointers` are hipStream_t values (pass as `stream=` to TridiagonalCholeskyFactor / CsrMatrix, or wrap
    with torch.cuda.ExternalStream); `.n_distinct` of them were measured to overlap pairwise."""

    def __init__(self, n: int, device: int = 0):
        import ctypes as C
        lib = _cabi.load()
        arr = (C.c_void_p * n)()
        nd = C.c_int32(0)
        _cabi.check(lib.gmrf_streams_create(device, n, arr, C.byref(nd)))
        self.device, self._arr, self._n = device, arr, n
        self.pointers = [int(arr[i] or 0) for i in range(n)]
        self.n_distinct = int(nd.value)

    def close(self):
        if self._arr is not None:
            _cabi.load().gmrf_streams_destroy(self.device, self._n, self._arr)
            self._arr, self.pointers = None, []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator of this process (one per GPU) through the C ABI -- what a Julia host uses to
    share a factor over xGMI (include/gmrf_hip.h, "multi-GPU").  `unique_id()` on rank 0, ship the
    128 bytes to the other ranks by any means, then `Comm(device, rank, world, id)` everywhere."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        _cabi.check(_cabi.load().gmrf_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, device: int, rank: int, world: int, uid: bytes):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        self._lib = _cabi.load()
        buf = (C.c_char * 128).from_buffer_copy(uid)
        _cabi.check(self._lib.gmrf_comm_create(device, rank, world, C.cast(buf, C.c_void_p), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gmrf_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bcast_host(self, arr: np.ndarray, root: int = 0) -> np.ndarray:
        """In-place broadcast of a small contiguous NumPy array."""
        _cabi.check(self._lib.gmrf_comm_bcast_host(self._h, _cabi.ptr(arr), arr.nbytes, root))
        return arr

    def bcast_blocks_async(self, F: TridiagonalCholeskyFactor, i0: int, i1: int, root: int = 0, with_l: bool = False):
        _cabi.check(self._lib.gmrf_bt_bcast_blocks_async(F._h, self._h, root, i0, i1, int(with_l)))

    def allgather_blocks_async(self, F_own: TridiagonalCholeskyFactor, F_all: TridiagonalCholeskyFactor, i0: int, i1: int):
        """Blocks [i0, i1) of every rank's share `F_own` (batch b) -> `F_all` (batch world * b) on every rank
        (gmrf_bt_allgather_blocks_async: pack, ncclAllGather, unpack on the communicator's stream)."""
        _cabi.check(self._lib.gmrf_bt_allgather_blocks_async(F_own._h, F_all._h, self._h, i0, i1))

    def wait(self, F: TridiagonalCholeskyFactor):
        _cabi.check(self._lib.gmrf_comm_wait(F._h, self._h))

    def bytes_moved(self, reset: bool = False) -> float:
        """Factor bytes broadcast through this communicator so far."""
        v = C.c_double(0.0)
        _cabi.check(self._lib.gmrf_comm_bytes(self._h, int(reset), C.byref(v)))
        return v.value

    def allreduce_sum(self, dev_tensor, F: Optional[TridiagonalCholeskyFactor] = None):
        _cabi.check(self._lib.gmrf_comm_allreduce_sum(self._h, F._h if F is not None else None, _cabi.ptr(dev_tensor),
                                                      dev_tensor.numel()))
        return dev_tensor
