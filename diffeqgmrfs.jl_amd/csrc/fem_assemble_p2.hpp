// FEM block assembly on the device, quadratic triangles: the Darcy stiffness matrix and load of
// /root/reference/src/problems/darcy.jl:5-63 with the reference's own element -- `Lagrange{RefTriangle,2}` and
// `QuadratureRule{RefTriangle}(3)` (src/utils.jl:32-33: element_order = 2) -- on the structured mesh of the BASELINE Darcy
// configs.  oracle/bt_oracle.py `assemble_darcy_diff_matrix_p2` restates the same loops and is the parity target.
//
// Mesh and conventions:
//   * the P1 triangulation of nx x ny vertices on the unit square (quads cut by the diagonal n00 - n11; cells: all lower
//     triangles (n00, n10, n11), then all upper (n00, n11, n01)), every cell with its three edge midpoints;
//   * dofs = the points of the (2 nx - 1) x (2 ny - 1) lattice, x fastest: (even, even) vertices, (odd, even) midpoints of
//     horizontal edges, (even, odd) of vertical edges, (odd, odd) of the diagonals;
//   * local node order of a cell (Ferrite): vertices 1, 2, 3 -- at xi = (1, 0), (0, 1), (0, 0) of the reference triangle --,
//     then the nodes of the edges (1-2), (2-3), (3-1); N_1 = xi_x (2 xi_x - 1), N_2 = xi_y (2 xi_y - 1), N_3 = g (2 g - 1),
//     N_4 = 4 xi_x xi_y, N_5 = 4 xi_y g, N_6 = 4 xi_x g with g = 1 - xi_x - xi_y;
//   * the 4-point Dunavant rule of degree 3: (1/3, 1/3; -27/96), (1/5, 1/5), (3/5, 1/5), (1/5, 3/5) with 25/96 each;
//     dOmega = w_q |det J|, J = [x_1 - x_3, x_2 - x_3] (straight-sided cells: the geometry is affine);
//   * the coefficient is looked up at every quadrature point by nearest grid point (src/datasets/darcy.jl:30-34).
//
// Gather instead of scatter, as in the P1 kernels: a thread owns one lattice point, walks its cells (6 for a vertex, 2 for an
// edge midpoint) in ascending cell number, forms the element row sum_q (...) in quadrature order and adds it to a 5 x 5
// window of lattice offsets; the window's touched entries leave in ascending column order = the CSR order of the pattern.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fem_assemble.hpp"

namespace gmrf {

struct DarcyP2Args {
    int nx, ny, ng;                 // vertices per direction, coefficient table size (ng x ng)
    const double* table;            // table[x index][y index]
    const int64_t* rowptr;          // CSR of the lattice coupling pattern
    double beta;
    double* vals;                   // [nnz]
    double* f;                      // [n]
    double* diag;                   // [n] |G_ii| before the constraints (meandiag)
};

__device__ __forceinline__ void p2_tri_shape_grad(int i, double xx, double xy, double& N, double& gx, double& gy) {
    const double g = 1.0 - xx - xy;
    switch (i) {
        case 0: N = xx * (2.0 * xx - 1.0); gx = 4.0 * xx - 1.0; gy = 0.0; break;
        case 1: N = xy * (2.0 * xy - 1.0); gx = 0.0; gy = 4.0 * xy - 1.0; break;
        case 2: N = g * (2.0 * g - 1.0); gx = -(4.0 * g - 1.0); gy = -(4.0 * g - 1.0); break;
        case 3: N = 4.0 * xx * xy; gx = 4.0 * xy; gy = 4.0 * xx; break;
        case 4: N = 4.0 * xy * g; gx = -4.0 * xy; gy = 4.0 * (g - xy); break;
        default: N = 4.0 * xx * g; gx = 4.0 * (g - xx); gy = -4.0 * xx; break;
    }
}

// lattice coordinates of the six nodes of cell (qx, qy, upper) in local order
__device__ __forceinline__ void p2_cell_nodes(int qx, int qy, bool upper, int (&nI)[6], int (&nJ)[6]) {
    const int I0 = 2 * qx, J0 = 2 * qy;
    nI[0] = I0; nJ[0] = J0;
    if (!upper) { nI[1] = I0 + 2; nJ[1] = J0; nI[2] = I0 + 2; nJ[2] = J0 + 2; }
    else { nI[1] = I0 + 2; nJ[1] = J0 + 2; nI[2] = I0; nJ[2] = J0 + 2; }
    nI[3] = (nI[0] + nI[1]) / 2; nJ[3] = (nJ[0] + nJ[1]) / 2;
    nI[4] = (nI[1] + nI[2]) / 2; nJ[4] = (nJ[1] + nJ[2]) / 2;
    nI[5] = (nI[2] + nI[0]) / 2; nJ[5] = (nJ[2] + nJ[0]) / 2;
}

__global__ __launch_bounds__(256) void darcy_p2_rows(DarcyP2Args a) {
    const int W = 2 * a.nx - 1, H = 2 * a.ny - 1;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (int64_t)W * H) return;
    const int I = (int)(row % W), J = (int)(row / W);
    double slot[25];
#pragma unroll
    for (int s = 0; s < 25; ++s) slot[s] = 0.0;
    unsigned present = 0u;
    double fi = 0.0;
    // the point's cells in ascending cell number: (quad offset from (I / 2, J / 2) rounded down, upper?, local index)
    int cand[6][4];
    int nc;
    if (!(I & 1) && !(J & 1)) {
        const int t[6][4] = {{-1, -1, 0, 2}, {-1, 0, 0, 1}, {0, 0, 0, 0}, {-1, -1, 1, 1}, {0, -1, 1, 2}, {0, 0, 1, 0}};
        nc = 6;
        for (int e = 0; e < 6; ++e) for (int u = 0; u < 4; ++u) cand[e][u] = t[e][u];
    } else if ((I & 1) && !(J & 1)) {           // horizontal edge: (1-2) of the lower cell above it, (2-3) of the upper cell below
        const int t[2][4] = {{0, 0, 0, 3}, {0, -1, 1, 4}};
        nc = 2;
        for (int e = 0; e < 2; ++e) for (int u = 0; u < 4; ++u) cand[e][u] = t[e][u];
    } else if (!(I & 1) && (J & 1)) {           // vertical edge: (2-3) of the lower cell left of it, (3-1) of the upper cell right
        const int t[2][4] = {{-1, 0, 0, 4}, {0, 0, 1, 5}};
        nc = 2;
        for (int e = 0; e < 2; ++e) for (int u = 0; u < 4; ++u) cand[e][u] = t[e][u];
    } else {                                    // diagonal: (3-1) of the lower, (1-2) of the upper cell of its quad
        const int t[2][4] = {{0, 0, 0, 5}, {0, 0, 1, 3}};
        nc = 2;
        for (int e = 0; e < 2; ++e) for (int u = 0; u < 4; ++u) cand[e][u] = t[e][u];
    }
    const double qxi[4] = {1.0 / 3.0, 0.2, 0.6, 0.2}, qeta[4] = {1.0 / 3.0, 0.2, 0.2, 0.6};
    const double qw[4] = {-27.0 / 96.0, 25.0 / 96.0, 25.0 / 96.0, 25.0 / 96.0};
    for (int e = 0; e < nc; ++e) {
        const int qx = I / 2 + cand[e][0], qy = J / 2 + cand[e][1];
        if (qx < 0 || qy < 0 || qx >= a.nx - 1 || qy >= a.ny - 1) continue;
        const bool upper = cand[e][2] != 0;
        const int li = cand[e][3];
        int nI[6], nJ[6];
        p2_cell_nodes(qx, qy, upper, nI, nJ);
        const double x1 = lin_coord(nI[0] / 2, a.nx), x2 = lin_coord(nI[1] / 2, a.nx), x3 = lin_coord(nI[2] / 2, a.nx);
        const double y1 = lin_coord(nJ[0] / 2, a.ny), y2 = lin_coord(nJ[1] / 2, a.ny), y3 = lin_coord(nJ[2] / 2, a.ny);
        const double ja = x1 - x3, jb = x2 - x3, jc = y1 - y3, jd = y2 - y3;           // J = [[ja, jb], [jc, jd]]
        const double det = ja * jd - jb * jc;
        double ge[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, fe = 0.0;
        for (int q = 0; q < 4; ++q) {
            const double xx = qxi[q], xy = qeta[q], g = 1.0 - xx - xy;
            // spatial_coordinate: unfused and in the oracle's order -- the nearest-grid-point lookup below has exact ties (a
            // quadrature point half-way between two table points) that a differently rounded x_q resolves the other way
            double xq, yq;
            {
#pragma clang fp contract(off)
                xq = (xx * x1 + xy * x2) + g * x3;
                yq = (xx * y1 + xy * y2) + g * y3;
            }
            const double coeff = a.table[(int64_t)nearest_grid_index(xq, a.ng) * a.ng + nearest_grid_index(yq, a.ng)];
            const double dO = qw[q] * fabs(det);
            double Ni, rx, ry;
            p2_tri_shape_grad(li, xx, xy, Ni, rx, ry);
            const double gix = (jd * rx - jc * ry) / det, giy = (-jb * rx + ja * ry) / det;      // J^-T grad_xi N_i
            fe += a.beta * Ni * dO;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double Nj, sx, sy;
                p2_tri_shape_grad(j, xx, xy, Nj, sx, sy);
                const double gjx = (jd * sx - jc * sy) / det, gjy = (-jb * sx + ja * sy) / det;
                ge[j] += coeff * (gix * gjx + giy * gjy) * dO;
            }
        }
        fi += fe;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int s = (nJ[j] - J + 2) * 5 + (nI[j] - I + 2);
            slot[s] += ge[j];
            present |= 1u << s;
        }
    }
    int64_t p = a.rowptr[row];
    for (int s = 0; s < 25; ++s)
        if (present & (1u << s)) a.vals[p++] = slot[s];
    a.f[row] = fi;
    a.diag[row] = fabs(slot[12]);
}

}  // namespace gmrf
