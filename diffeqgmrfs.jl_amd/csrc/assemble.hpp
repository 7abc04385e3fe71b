// Posterior assembly on the device: the step right before the factorisation
// in the reference's Gauss-Newton loop, /root/reference/scripts/solve_burger.jl:143-149
//     A   = Q + noise * J' * J
//     rhs = Qx_prior + noise * J' * (J * x + obs_diff)
// and in `condition_on_observations` (Q + A' Q_eps A; scripts/darcy/solve_darcy_gmrf-fem.jl:161).
// The sparsity patterns are fixed over the iterations, the values of J change: the symbolic
// phase (host, once) lists for every entry of the result the products J[k,i] * J[k,j] that feed
// it; the numeric phase is three gather kernels on values that stay in HBM, and its output is the
// `nzval` array gmrf_bt_refactor_values takes -- no host round trip per iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

// out[e] = (qmap[e] >= 0 ? q[qmap[e]] : 0) + noise * sum_{p in [pptr[e], pptr[e+1])} jv[pa[p]] * jv[pb[p]]
// The products of an entry are listed with ascending row k of J: a fixed summation order.
__global__ __launch_bounds__(256) void assemble_precision(const int64_t* __restrict__ pptr, const int32_t* __restrict__ pa,
                                                          const int32_t* __restrict__ pb, const int64_t* __restrict__ qmap,
                                                          const double* __restrict__ q, const double* __restrict__ jv,
                                                          double noise, int64_t nnz_out, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz_out) return;
    double s = 0.0;
    for (int64_t p = pptr[e]; p < pptr[e + 1]; ++p) s = fma(jv[pa[p]], jv[pb[p]], s);
    const int64_t m = qmap[e];
    out[e] = (m >= 0 ? q[m] : 0.0) + noise * s;
}

// out[i] = base[i] + noise * sum_p jv[src[p]] * v[row[p]]   over the entries of column i of J  (J' v)
__global__ __launch_bounds__(256) void assemble_jt_apply(const int64_t* __restrict__ cptr, const int32_t* __restrict__ row,
                                                         const int32_t* __restrict__ src, const double* __restrict__ jv,
                                                         const double* __restrict__ v, const double* __restrict__ base,
                                                         double noise, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int64_t p = cptr[i]; p < cptr[i + 1]; ++p) s = fma(jv[src[p]], v[row[p]], s);
    out[i] = (base ? base[i] : 0.0) + noise * s;
}

// out[k] = add[k] + sum_p jv[p] * x[col[p]]   over row k of J  (J x + obs_diff)
__global__ __launch_bounds__(256) void assemble_j_apply(const int64_t* __restrict__ rptr, const int32_t* __restrict__ col,
                                                        const double* __restrict__ jv, const double* __restrict__ x,
                                                        const double* __restrict__ add, int64_t m, double* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    double s = 0.0;
    for (int64_t p = rptr[k]; p < rptr[k + 1]; ++p) s = fma(jv[p], x[col[p]], s);
    out[k] = (add ? add[k] : 0.0) + s;
}

}  // namespace gmrf
