// FEM block assembly on the device, first piece: the Darcy stiffness matrix
//     G[i][j] = int a(x) grad(phi_i) . grad(phi_j),   f[i] = beta int phi_i
// of /root/reference/src/problems/darcy.jl:27-60 (cell loop, coefficient looked up at the quadrature point
// by nearest grid point, src/datasets/darcy.jl:30-34; `apply!(G, f, ch)` for the Dirichlet nodes, :61)
// on the structured P1 mesh of the BASELINE Darcy configs: nx x ny nodes on the unit square, x fastest,
// every quad cut by the diagonal n00 - n11 into the triangles (n00, n10, n11) and (n00, n11, n01), one
// quadrature point (the centroid) per cell.  "PDE Discretization" is the other per-problem timer of the
// reference's loop (scripts/darcy/solve_darcy_gmrf-fem.jl:179); with this kernel the coefficient table is
// all that crosses the bus per problem: the values land in the CSR order gmrf_assemble_precision takes
// as `J` (A = G, Q_post = Q + Q_eps A'A).
//
// Gather instead of scatter: a thread owns one node row and walks its (at most six) cells in ascending
// cell number (all lower triangles before all upper ones, like the cell iterator), so every entry is a
// sum in a fixed order and no atomics are needed.  The 7-point stencil (offsets -nx-1, -nx, -1, 0, +1,
// +nx, +nx+1, clipped at the border) is the row's CSR entry list in ascending column order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

struct DarcyP1Args {
    int nx, ny, ng;                 // mesh nodes per direction, coefficient table size (ng x ng)
    const double* table;            // coeff[ix_grid * ng + iy_grid]: table[x index][y index]
    const int64_t* rowptr;          // CSR of the 7-point stencil
    double beta;
    double* vals;                   // [nnz]
    double* f;                      // [n]
    double* diag;                   // [n] raw diagonal (before the constraints), for meandiag
};

__device__ __forceinline__ double lin_coord(int i, int n) {      // numpy.linspace(0, 1, n)[i]
    return (i == n - 1) ? 1.0 : (double)i * (1.0 / (double)(n - 1));
}

// argmin_k |grid[k] - p| with the first minimum winning (Julia argmin, src/datasets/darcy.jl:31-32)
__device__ __forceinline__ int nearest_grid_index(double p, int ng) {
    int k0 = (int)floor(p * (double)(ng - 1));
    k0 = max(0, min(k0, ng - 1));
    int best = k0;
    double bd = fabs(lin_coord(k0, ng) - p);
    for (int k = max(0, k0 - 1); k <= min(ng - 1, k0 + 2); ++k) {
        const double d = fabs(lin_coord(k, ng) - p);
        if (d < bd || (d == bd && k < best)) { bd = d; best = k; }
    }
    return best;
}

// One cell: nodes (node ids in cell order) -> local stiffness row `li` (three values) and the cell's share of f.
__device__ __forceinline__ void darcy_cell_row(const DarcyP1Args& a, int qx, int qy, bool upper, int li, double (&ke)[3],
                                               double& fe) {
    // cell nodes: lower (n00, n10, n11), upper (n00, n11, n01)
    const int nxs[3] = {qx, upper ? qx + 1 : qx + 1, upper ? qx : qx + 1};
    const int nys[3] = {qy, upper ? qy + 1 : qy, qy + 1};
    double x[3], y[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) { x[v] = lin_coord(nxs[v], a.nx); y[v] = lin_coord(nys[v], a.ny); }
    // P1 gradients: grad phi_v = (b_v, c_v) / (2 area)
    const double b[3] = {y[1] - y[2], y[2] - y[0], y[0] - y[1]};
    const double c[3] = {x[2] - x[1], x[0] - x[2], x[1] - x[0]};
    const double area2 = x[0] * b[0] + x[1] * b[1] + x[2] * b[2];
    const double area = 0.5 * fabs(area2);
    // quadrature point = centroid; coefficient by nearest grid point
    const double xq = ((x[0] + x[1]) + x[2]) / 3.0, yq = ((y[0] + y[1]) + y[2]) / 3.0;
    const double coeff = a.table[(int64_t)nearest_grid_index(xq, a.ng) * a.ng + nearest_grid_index(yq, a.ng)];
#pragma unroll
    for (int v = 0; v < 3; ++v) ke[v] = (b[li] * b[v] + c[li] * c[v]) / (4.0 * area) * coeff;
    fe = a.beta * (area / 3.0);
}

// stencil slot (0..6) of the neighbour at (dx, dy): -nx-1, -nx, -1, 0, +1, +nx, +nx+1
__device__ __forceinline__ int stencil_slot(int dx, int dy) {
    return dy == -1 ? (dx == -1 ? 0 : 1) : (dy == 0 ? (dx == -1 ? 2 : (dx == 0 ? 3 : 4)) : (dx == 0 ? 5 : 6));
}

__global__ __launch_bounds__(256) void darcy_p1_rows(DarcyP1Args a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)a.nx * a.ny;
    if (i >= n) return;
    const int ix = (int)(i % a.nx), iy = (int)(i / a.nx);
    double slot[7] = {0, 0, 0, 0, 0, 0, 0};
    double fi = 0.0;
    // the node's cells in ascending cell number: lower triangles of the quads (ix-1,iy-1), (ix-1,iy), (ix,iy), then the
    // upper triangles of (ix-1,iy-1), (ix,iy-1), (ix,iy); (quad, triangle, local index of this node in the cell)
    const int cq[6][4] = {{-1, -1, 0, 2}, {-1, 0, 0, 1}, {0, 0, 0, 0}, {-1, -1, 1, 1}, {0, -1, 1, 2}, {0, 0, 1, 0}};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int qx = ix + cq[e][0], qy = iy + cq[e][1];
        if (qx < 0 || qy < 0 || qx >= a.nx - 1 || qy >= a.ny - 1) continue;
        const bool upper = cq[e][2] != 0;
        double ke[3], fe;
        darcy_cell_row(a, qx, qy, upper, cq[e][3], ke, fe);
        const int nxs[3] = {qx, qx + 1, upper ? qx : qx + 1};
        const int nys[3] = {qy, upper ? qy + 1 : qy, qy + 1};
#pragma unroll
        for (int v = 0; v < 3; ++v) slot[stencil_slot(nxs[v] - ix, nys[v] - iy)] += ke[v];
        fi += fe;
    }
    // the row's entries in ascending column order (clipped stencil)
    int64_t p = a.rowptr[i];
    const bool has[7] = {ix > 0 && iy > 0, iy > 0, ix > 0, true, ix < a.nx - 1, iy < a.ny - 1, ix < a.nx - 1 && iy < a.ny - 1};
#pragma unroll
    for (int s = 0; s < 7; ++s)
        if (has[s]) a.vals[p++] = slot[s];
    a.f[i] = fi;
    a.diag[i] = fabs(slot[3]);
}

// meandiag: sum of |G_ii| over all nodes in a fixed order (one workgroup), divided by n
__global__ __launch_bounds__(256) void darcy_meandiag(const double* __restrict__ diag, int64_t n, double* __restrict__ out) {
    __shared__ double red[256];
    const int64_t chunk = (n + 255) / 256, lo = (int64_t)threadIdx.x * chunk, hi = min(n, lo + chunk);
    double s = 0.0;
    for (int64_t i = lo; i < hi; ++i) s += diag[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (double)n;
}

// apply!(G, f, ch) for homogeneous Dirichlet data on the boundary nodes: constrained rows and columns are
// zeroed, the constrained diagonal entries become meandiag, f vanishes there (src/problems/darcy.jl:61)
__global__ __launch_bounds__(256) void darcy_p1_constrain(DarcyP1Args a, const double* __restrict__ meandiag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)a.nx * a.ny;
    if (i >= n) return;
    const int ix = (int)(i % a.nx), iy = (int)(i / a.nx);
    const bool bi = ix == 0 || iy == 0 || ix == a.nx - 1 || iy == a.ny - 1;
    const int dxs[7] = {-1, 0, -1, 0, 1, 0, 1}, dys[7] = {-1, -1, 0, 0, 0, 1, 1};
    int64_t p = a.rowptr[i];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int jx = ix + dxs[s], jy = iy + dys[s];
        if (jx < 0 || jy < 0 || jx >= a.nx || jy >= a.ny) continue;
        const bool bj = jx == 0 || jy == 0 || jx == a.nx - 1 || jy == a.ny - 1;
        if (bi || bj) a.vals[p] = (s == 3) ? meandiag[0] : 0.0;
        ++p;
    }
    if (bi) a.f[i] = 0.0;
}

}  // namespace gmrf
