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

// numpy.linspace(0, 1, n)[i].  The product must stay a product: fused into a following subtraction (hipcc contracts
// a * b - c by default) the coordinate differences and the distances of the nearest-grid-point lookup round differently
// from the host's, which decides exact ties (a quadrature point half-way between two table points) the other way.
__device__ __forceinline__ double lin_coord(int i, int n) {
#pragma clang fp contract(off)
    return (i == n - 1) ? 1.0 : (double)i * (1.0 / (double)(n - 1));
}

// argmin_k |grid[k] - p| with the first minimum winning (Julia argmin, src/datasets/darcy.jl:31-32)
__device__ __forceinline__ int nearest_grid_index(double p, int ng) {
#pragma clang fp contract(off)
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


// ------------------------------------------------------------------------------------------------
// FEM block assembly on the device, second piece: residual and tangent of the
// implicit-Euler Burgers space-time system,
//     J(w) = J_static + dt J_adv(w),      f(w) = J_static w + dt v_adv(w)
// -- f_and_J / nonlinear_primal_tangent, /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:118-149 --
// with  J_static = M_{t+1} - M_t + dt nu G_{t+1}  (:123-130; assemble_burgers_mass_diffusion_matrices,
// src/problems/burgers.jl:60-98) and, per time slice, the advection tangent and residual of
// assemble_burgers_advection_matrix (src/problems/burgers.jl:5-59, cell loop :22-51):
//     Ge[i][j] += N_i (N_j grad(u) + u grad(N_j)) dOmega,      ve[i] += N_i u grad(u) dOmega
// on the periodic P1 line of the BASELINE Burgers configs (ns nodes on [0,1), cell e = (e, e + 1 mod ns),
// 3-point Gauss rule as QuadratureRule{1,RefCube}(3)); the periodic constraint of the reference's mesh is
// the wrap-around of this one (no slave dofs).  Time-major index (t-1) ns + s; rows = slices 2 .. nt.
//
// Gather instead of scatter: a thread owns the row (t, i), evaluates its two cells (left: nodes i-1, i; right:
// nodes i, i+1) with the reference's quadrature loop and writes the row's 6 entries in ascending column order
// -- the CSR order gmrf_assemble_precision takes as `J` -- and f(t, i).  Nothing but w crosses the bus per
// Gauss-Newton iteration.
struct BurgersP1Args {
    int ns, nt;
    double dt, nu;
    const double* w;                // [nt * ns]
    double* vals;                   // [(nt - 1) * ns * 6]
    double* f;                      // [(nt - 1) * ns]
};

// one P1 cell of length h with nodal values (w0, w1): advection tangent Ge, residual ve, mass Me, diffusion De
__host__ __device__ inline void burgers_p1_cell(double h, double w0, double w1, double (&Ge)[2][2], double (&ve)[2],
                                                double (&Me)[2][2], double (&De)[2][2]) {
    const double xi[3] = {-0.7745966692414834, 0.0, 0.7745966692414834};          // sqrt(3/5)
    const double wq[3] = {0.5555555555555556, 0.8888888888888888, 0.5555555555555556};
    const double jac = 0.5 * h;                                   // dx / dxi
    const double dN[2] = {-0.5 / jac, 0.5 / jac};                 // shape_gradient
    for (int i = 0; i < 2; ++i) { ve[i] = 0.0; for (int j = 0; j < 2; ++j) { Ge[i][j] = 0.0; Me[i][j] = 0.0; De[i][j] = 0.0; } }
    for (int q = 0; q < 3; ++q) {
        const double dOm = jac * wq[q];                           // getdetJdV
        const double N[2] = {0.5 * (1.0 - xi[q]), 0.5 * (1.0 + xi[q])};
        const double cur_u = N[0] * w0 + N[1] * w1;               // function_value
        double grad_u = 0.0;                                      // :36-39
        grad_u += dN[0] * w0;
        grad_u += dN[1] * w1;
        for (int i = 0; i < 2; ++i) {
            for (int j = 0; j < 2; ++j) {
                Ge[i][j] += N[i] * (N[j] * grad_u + cur_u * dN[j]) * dOm;      // :46
                Me[i][j] += N[i] * N[j] * dOm;
                De[i][j] += dN[i] * dN[j] * dOm;
            }
            ve[i] += N[i] * cur_u * grad_u * dOm;                 // :48
        }
    }
}

__global__ __launch_bounds__(256) void burgers_p1_rows(BurgersP1Args a) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)(a.nt - 1) * a.ns;
    if (gid >= rows) return;
    const int t = (int)(gid / a.ns) + 1, i = (int)(gid % a.ns);   // slice t (0-based), rows belong to slices 1 .. nt-1
    const int im = (i == 0) ? a.ns - 1 : i - 1, ip = (i == a.ns - 1) ? 0 : i + 1;
    const double h = 1.0 / (double)a.ns;
    const double* wt = a.w + (int64_t)t * a.ns;
    const double* wp = a.w + (int64_t)(t - 1) * a.ns;
    const double wl = wt[im], wc = wt[i], wr = wt[ip];
    double GeL[2][2], veL[2], MeL[2][2], DeL[2][2], GeR[2][2], veR[2], MeR[2][2], DeR[2][2];
    burgers_p1_cell(h, wl, wc, GeL, veL, MeL, DeL);               // cell (i-1, i): this node is local 1
    burgers_p1_cell(h, wc, wr, GeR, veR, MeR, DeR);               // cell (i, i+1): this node is local 0
    // assembled rows (two addends on the diagonal: the order of the two cells does not matter)
    const double m_m = MeL[1][0], m_0 = MeL[1][1] + MeR[0][0], m_p = MeR[0][1];
    const double d_m = DeL[1][0], d_0 = DeL[1][1] + DeR[0][0], d_p = DeR[0][1];
    const double g_m = GeL[1][0], g_0 = GeL[1][1] + GeR[0][0], g_p = GeR[0][1];
    const double v_i = veL[1] + veR[0];
    const double dtnu = a.dt * a.nu;
    // J_static = M_{t+1} - M_t + dt nu G_{t+1}
    const double sp_m = -m_m, sp_0 = -m_0, sp_p = -m_p;                              // block t-1
    const double st_m = m_m + dtnu * d_m, st_0 = m_0 + dtnu * d_0, st_p = m_p + dtnu * d_p;   // block t
    // ascending column order inside a block: the wrap-around columns of the first / last node move
    int o_m = 0, o_0 = 1, o_p = 2;
    if (i == 0) { o_0 = 0; o_p = 1; o_m = 2; }
    else if (i == a.ns - 1) { o_p = 0; o_m = 1; o_0 = 2; }
    double* v = a.vals + gid * 6;
    v[o_m] = sp_m; v[o_0] = sp_0; v[o_p] = sp_p;
    v[3 + o_m] = st_m + a.dt * g_m; v[3 + o_0] = st_0 + a.dt * g_0; v[3 + o_p] = st_p + a.dt * g_p;
    // f = J_static * w + dt * v_adv : the product sums a row in ascending column order
    double sv[6], xv[6];
    sv[o_m] = sp_m; sv[o_0] = sp_0; sv[o_p] = sp_p; sv[3 + o_m] = st_m; sv[3 + o_0] = st_0; sv[3 + o_p] = st_p;
    xv[o_m] = wp[im]; xv[o_0] = wp[i]; xv[o_p] = wp[ip]; xv[3 + o_m] = wl; xv[3 + o_0] = wc; xv[3 + o_p] = wr;
    double acc = 0.0;
    for (int e = 0; e < 6; ++e) acc += sv[e] * xv[e];
    a.f[gid] = acc + a.dt * v_i;
}

// ------------------------------------------------------------------------------------------------
// The same residual and tangent on the QUADRATIC periodic line -- the element the reference's Burgers scripts run on
// (`periodic_unit_interval_discretization`, /root/reference/src/utils.jl:42-49: generate_grid(QuadraticLine, (N_x,)),
// Lagrange{RefLine,2}, QuadratureRule{RefLine}(3)).  ns = 2 N_x dofs numbered by position (x_i = i / ns): even i are cell
// boundaries, odd i midpoints; cell e has the local dofs (left, right, middle) = (2 e, 2 e + 2 mod ns, 2 e + 1) with
// N = (xi (xi - 1) / 2, xi (xi + 1) / 2, 1 - xi^2).  A vertex row couples with its two cells (columns i-2 .. i+2), a
// midpoint row with its own cell (i-1, i, i+1): 10 / 6 entries per row of J (slices t-1 and t), ascending columns.
__host__ __device__ inline void burgers_p2_cell(double h, const double (&w)[3], double (&Ge)[3][3], double (&ve)[3],
                                                double (&Me)[3][3], double (&De)[3][3]) {
    const double xi[3] = {-0.7745966692414834, 0.0, 0.7745966692414834};
    const double wq[3] = {0.5555555555555556, 0.8888888888888888, 0.5555555555555556};
    const double jac = 0.5 * h;
    for (int i = 0; i < 3; ++i) { ve[i] = 0.0; for (int j = 0; j < 3; ++j) { Ge[i][j] = 0.0; Me[i][j] = 0.0; De[i][j] = 0.0; } }
    for (int q = 0; q < 3; ++q) {
        const double x = xi[q], dOm = jac * wq[q];
        const double N[3] = {0.5 * x * (x - 1.0), 0.5 * x * (x + 1.0), 1.0 - x * x};
        const double dN[3] = {(x - 0.5) / jac, (x + 0.5) / jac, (-2.0 * x) / jac};
        double cur_u = 0.0, grad_u = 0.0;
        for (int k = 0; k < 3; ++k) cur_u += N[k] * w[k];         // function_value :34
        for (int k = 0; k < 3; ++k) grad_u += dN[k] * w[k];       // :36-39
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Ge[i][j] += N[i] * (N[j] * grad_u + cur_u * dN[j]) * dOm;      // :46
                Me[i][j] += N[i] * N[j] * dOm;
                De[i][j] += dN[i] * dN[j] * dOm;
            }
            ve[i] += N[i] * cur_u * grad_u * dOm;                 // :48
        }
    }
}

// first entry of row (t, i) in the value array: 16 entries per cell and slice (10 for the vertex row, 6 for the midpoint row)
__host__ __device__ inline int64_t burgers_p2_row_offset(int64_t ns, int64_t t1, int64_t i) {      // t1 = t - 1 >= 0
    return (t1 * (ns / 2) + i / 2) * 16 + ((i & 1) ? 10 : 0);
}

__global__ __launch_bounds__(256) void burgers_p2_rows(BurgersP1Args a) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)(a.nt - 1) * a.ns;
    if (gid >= rows) return;
    const int ns = a.ns, nc = ns / 2;
    const int t = (int)(gid / ns) + 1, i = (int)(gid % ns);
    const double h = 1.0 / (double)nc;
    const double* wt = a.w + (int64_t)t * ns;
    const double* wp = a.w + (int64_t)(t - 1) * ns;
    auto wrap = [ns](int c) { return c < 0 ? c + ns : (c >= ns ? c - ns : c); };
    // the row's columns (before the wrap) and its assembled mass / diffusion / advection entries
    int cols[5]; double m[5], d[5], g[5]; int cnt; double v_i;
    double Ge[3][3], ve[3], Me[3][3], De[3][3];
    if (i & 1) {                                       // midpoint of cell e = (i - 1) / 2: local dof 2
        const double w3[3] = {wt[i - 1], wt[wrap(i + 1)], wt[i]};
        burgers_p2_cell(h, w3, Ge, ve, Me, De);
        cnt = 3;
        cols[0] = i - 1; cols[1] = i; cols[2] = i + 1;
        const int loc[3] = {0, 2, 1};                  // columns left, middle, right in local numbering
        for (int c = 0; c < 3; ++c) { m[c] = Me[2][loc[c]]; d[c] = De[2][loc[c]]; g[c] = Ge[2][loc[c]]; }
        v_i = ve[2];
    } else {                                           // vertex: right end (local 1) of cell i/2 - 1, left end (local 0) of cell i/2
        const double wl[3] = {wt[wrap(i - 2)], wt[i], wt[wrap(i - 1)]};
        const double wr[3] = {wt[i], wt[wrap(i + 2)], wt[wrap(i + 1)]};
        double GeR[3][3], veR[3], MeR[3][3], DeR[3][3];
        burgers_p2_cell(h, wl, Ge, ve, Me, De);
        burgers_p2_cell(h, wr, GeR, veR, MeR, DeR);
        cnt = 5;
        for (int c = 0; c < 5; ++c) cols[c] = i - 2 + c;
        m[0] = Me[1][0]; m[1] = Me[1][2]; m[2] = Me[1][1] + MeR[0][0]; m[3] = MeR[0][2]; m[4] = MeR[0][1];
        d[0] = De[1][0]; d[1] = De[1][2]; d[2] = De[1][1] + DeR[0][0]; d[3] = DeR[0][2]; d[4] = DeR[0][1];
        g[0] = Ge[1][0]; g[1] = Ge[1][2]; g[2] = Ge[1][1] + GeR[0][0]; g[3] = GeR[0][2]; g[4] = GeR[0][1];
        v_i = ve[1] + veR[0];
    }
    // ascending column order after the wrap: a rotation of the list (columns < 0 move to the end, columns >= ns to the front)
    int first = 0;
    for (int c = 0; c < cnt; ++c) if (cols[c] >= ns) { first = c; break; }
    if (cols[0] < 0) { first = 0; while (cols[first] < 0) ++first; }
    const double dtnu = a.dt * a.nu;
    double* v = a.vals + burgers_p2_row_offset(ns, t - 1, i);
    double acc = 0.0;
    double prev_terms[5], cur_terms[5];
    for (int k = 0; k < cnt; ++k) {
        const int c = (first + k) % cnt, col = wrap(cols[c]);
        const double sp = -m[c], st = m[c] + dtnu * d[c];          // J_static = M_{t+1} - M_t + dt nu G_{t+1}
        v[k] = sp;
        v[cnt + k] = st + a.dt * g[c];
        prev_terms[k] = sp * wp[col];
        cur_terms[k] = st * wt[col];
    }
    // f = J_static * w + dt * v_adv: the product sums a row in ascending column order (slice t-1 first)
    for (int k = 0; k < cnt; ++k) acc += prev_terms[k];
    for (int k = 0; k < cnt; ++k) acc += cur_terms[k];
    a.f[gid] = acc + a.dt * v_i;
}

}  // namespace gmrf
