// FEM block assembly on the device, third piece: the linear shallow-water SPDE,
// /root/reference/src/spdes/shallow_water.jl -- the element loops of `assemble_system!` (:17-122: coupling matrix K,
// element-lumped mass M, stiffness S of the three-field system h, u, v) and the per-step operators `discretize`
// forms from them (:170-217: M~, the Matern square root sqrt(ratio) M~^-1/2 (kappa^2 M~ + G), beta(dt), G(dt) = M~ + dt K).
//
// Mesh and conventions (the reference's mesh comes from Gmsh through Ferrite, both absent; oracle/bt_oracle.py
// `assemble_shallow_water_system` restates the same loops line by line on this mesh and is the parity target):
//   * nx x ny nodes on the unit square, x fastest, every quad cut by the diagonal n00 - n11 into the P1 triangles
//     (n00, n10, n11) [cell qy (nx-1) + qx] and (n00, n11, n01) [cell (nx-1)(ny-1) + qy (nx-1) + qx];
//   * dof = 3 * node + field, fields (h, u, v) = (0, 1, 2);
//   * the symmetric 3-point rule (QuadratureRule{2,RefTetrahedron}(2)): dOmega = |T| / 3, point q has the barycentric
//     weight 2/3 on cell vertex 2 - q and 1/6 on the other two; H enters through its values at the quadrature
//     points, H_q[cell][q] (the caller evaluates its H(x) at gmrf_shallow_water_p1_qpoints).
//
// Gather instead of scatter, as in the Darcy and Burgers kernels: a thread owns the row of one dof (node i, field a),
// walks the node's (at most six) cells in ascending cell number -- all lower triangles before the upper ones, like the
// cell iterator -- forms, per cell, the element row sum_q (...) in quadrature order and adds it to the row's slots:
// fixed summation order, no atomics.  K rows have the full field coupling over the 7-point node stencil (21 entries
// for an interior node, explicit zeros kept as create_sparsity_pattern(dh, ch) keeps them, :140), S rows the
// block-diagonal coupling (7 entries, :141-150); M is element-lumped (`lump_matrix(me, ip)`, :116: row sums for the
// linear Lagrange interpolation) and therefore a vector.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fem_assemble.hpp"

namespace gmrf {

struct SweP1Args {
    int nx, ny;
    const double* Hq;               // [cells][3]
    double k, f, g;
    const int64_t* rowptr_k;        // [3 nn + 1]
    const int64_t* rowptr_s;        // [3 nn + 1]
    double* kv;                     // [nnz_k]
    double* sv;                     // [nnz_s]
    double* ml;                     // [3 nn]
    double* dk;                     // [3 nn] |K_ii|, |S_ii| before the constraints (for meandiag)
    double* ds;
};

__global__ __launch_bounds__(256) void swe_p1_rows(SweP1Args a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nn = (int64_t)a.nx * a.ny;
    if (row >= 3 * nn) return;
    const int64_t node = row / 3;
    const int fa = (int)(row % 3);
    const int ix = (int)(node % a.nx), iy = (int)(node / a.nx);
    double kslot[7][3], sslot[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) { sslot[s] = 0.0; kslot[s][0] = kslot[s][1] = kslot[s][2] = 0.0; }
    double mrow = 0.0;
    const int cq[6][4] = {{-1, -1, 0, 2}, {-1, 0, 0, 1}, {0, 0, 0, 0}, {-1, -1, 1, 1}, {0, -1, 1, 2}, {0, 0, 1, 0}};
    const int64_t nlow = (int64_t)(a.nx - 1) * (a.ny - 1);
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int qx = ix + cq[e][0], qy = iy + cq[e][1];
        if (qx < 0 || qy < 0 || qx >= a.nx - 1 || qy >= a.ny - 1) continue;
        const bool upper = cq[e][2] != 0;
        const int li = cq[e][3];
        const int nxs[3] = {qx, qx + 1, upper ? qx : qx + 1};
        const int nys[3] = {qy, upper ? qy + 1 : qy, qy + 1};
        double x[3], y[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) { x[v] = lin_coord(nxs[v], a.nx); y[v] = lin_coord(nys[v], a.ny); }
        const double b[3] = {y[1] - y[2], y[2] - y[0], y[0] - y[1]};
        const double c[3] = {x[2] - x[1], x[0] - x[2], x[1] - x[0]};
        // signed 2 |T| as the determinant of the cell Jacobian, (x1 - x0)(y2 - y0) - (x2 - x0)(y1 - y0) (what reinit! forms;
        // sum_v x_v b_v is the same number with (n - 1)-fold cancellation: S ~ 1 / |T| would carry it)
        const double area2 = c[2] * b[1] - c[1] * b[2];
        const double dO = 0.5 * fabs(area2) / 3.0;
        double gx[3], gy[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) { gx[v] = b[v] / area2; gy[v] = c[v] / area2; }
        const int64_t cell = (upper ? nlow : 0) + (int64_t)qy * (a.nx - 1) + qx;
        // element row of this dof: ke[v][b] (columns vertex v, field b), se[v], me[v] -- sums over the quadrature points
        double ke[3][3], se[3], me[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) { se[v] = 0.0; me[v] = 0.0; ke[v][0] = ke[v][1] = ke[v][2] = 0.0; }
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
            const double Hv = a.Hq[cell * 3 + qp];
            const double phi_i = (li == 2 - qp) ? (2.0 / 3.0) : (1.0 / 6.0);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double phi_j = (v == 2 - qp) ? (2.0 / 3.0) : (1.0 / 6.0);
                const double pp = phi_i * phi_j * dO;
                const double gg = (gx[li] * gx[v] + gy[li] * gy[v]) * dO;
                me[v] += pp;
                se[v] += gg;
                if (fa == 0) {               // row h: h-u, h-v                      (:73-79)
                    ke[v][1] += -Hv * gx[li] * phi_j * dO;
                    ke[v][2] += -Hv * gy[li] * phi_j * dO;
                } else if (fa == 1) {        // row u: u-h, u-u, u-v                 (:82-95)
                    ke[v][0] += -a.g * gx[li] * phi_j * dO;
                    ke[v][1] += a.k * pp;
                    ke[v][2] += -a.f * pp;
                } else {                     // row v: v-h, v-u, v-v                 (:99-112)
                    ke[v][0] += -a.g * gy[li] * phi_j * dO;
                    ke[v][1] += a.f * pp;
                    ke[v][2] += a.k * pp;
                }
            }
        }
        // assemble! (:114-118): the element row lands on the row's stencil slots, the lumped mass on the diagonal
        mrow += (me[0] + me[1]) + me[2];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int s = stencil_slot(nxs[v] - ix, nys[v] - iy);
            sslot[s] += se[v];
            kslot[s][0] += ke[v][0]; kslot[s][1] += ke[v][1]; kslot[s][2] += ke[v][2];
        }
    }
    const bool has[7] = {ix > 0 && iy > 0, iy > 0, ix > 0, true, ix < a.nx - 1, iy < a.ny - 1, ix < a.nx - 1 && iy < a.ny - 1};
    int64_t pk = a.rowptr_k[row], ps = a.rowptr_s[row];
#pragma unroll
    for (int s = 0; s < 7; ++s)
        if (has[s]) {
            a.kv[pk++] = kslot[s][0]; a.kv[pk++] = kslot[s][1]; a.kv[pk++] = kslot[s][2];
            a.sv[ps++] = sslot[s];
        }
    a.ml[row] = mrow;
    a.dk[row] = fabs(kslot[3][fa]);
    a.ds[row] = fabs(sslot[3]);
}

// Ferrite `apply!(A, zeros, ch)` on CSR values in place: entries in a prescribed row or column vanish, a prescribed
// diagonal entry becomes meandiag.  One thread per row; colidx: 32-bit columns of the pattern.
__global__ __launch_bounds__(256) void csr_apply_constraints(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                             const uint8_t* __restrict__ pres, int64_t n,
                                                             const double* __restrict__ meandiag, double* __restrict__ vals) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const bool pr = pres[row] != 0;
    for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
        const int64_t col = colidx[p];
        if (pr || pres[col]) vals[p] = (col == row) ? meandiag[0] : 0.0;
    }
}

// prescribed entries of a diagonal matrix: v[d] = value (a device scalar, e.g. meandiag) or a constant
__global__ __launch_bounds__(256) void vec_set_prescribed(const uint8_t* __restrict__ pres, int64_t n, const double* __restrict__ dev_value,
                                                          double constant, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && pres[i]) v[i] = dev_value ? dev_value[0] : constant;
}

// The per-step operators of `discretize` (:170-217), one thread per dof row:
//   Mt[row]   = M~: M with 1e-2 on prescribed dofs                                             (:172-174)
//   beta[row] = sqrt(dt) * (prescribed ? 1e-2 : tau)                                            (:198-211)
//   J[row][:] = sqrt(ratio) / sqrt(Mt[row]) * (kappa^2 Mt on the diagonal + G), G = S with 1 on prescribed diagonals,
//               in S's pattern: Q_matern = J'J, J' = Q_matern_sqrt                              (:173,:178,:187-189)
//   Gd[row][:] = Mt on the diagonal + dt K, in K's pattern (its |diagonal| goes to dg for the meandiag of the
//               apply! that follows, :212-217)
struct SweOpArgs {
    int64_t n;
    const int64_t* rowptr_k; const int32_t* col_k; const double* kv;
    const int64_t* rowptr_s; const int32_t* col_s; const double* sv;
    const double* ml;
    const uint8_t* pres;            // may be nullptr
    double kappa2, sqrt_ratio, tau, dt;
    double* Mt; double* beta; double* J; double* Gd; double* dg;
};

__global__ __launch_bounds__(256) void swe_p1_operators(SweOpArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.n) return;
    const bool pr = a.pres && a.pres[row];
    const double mt = pr ? 1e-2 : a.ml[row];
    a.Mt[row] = mt;
    a.beta[row] = sqrt(a.dt) * (pr ? 1e-2 : a.tau);
    const double sc = a.sqrt_ratio * sqrt(1.0 / mt);
    for (int64_t p = a.rowptr_s[row]; p < a.rowptr_s[row + 1]; ++p) {
        const bool dg = a.col_s[p] == row;
        const double gv = (dg && pr) ? 1.0 : a.sv[p];
        a.J[p] = sc * (dg ? a.kappa2 * mt + gv : gv);
    }
    double d = 0.0;
    for (int64_t p = a.rowptr_k[row]; p < a.rowptr_k[row + 1]; ++p) {
        const bool dg = a.col_k[p] == row;
        const double v = dg ? mt + a.dt * a.kv[p] : a.dt * a.kv[p];
        a.Gd[p] = v;
        if (dg) d = fabs(v);
    }
    a.dg[row] = d;
}

}  // namespace gmrf
