// Dense fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// This is the workhorse of the block factorisation: it replaces the dtrsm / dsyrk / dgemm
// calls Julia's LinearAlgebra makes for /root/reference/src/tridiagonal_cholesky.jl:74,77
// (C = B L^-T, D - C C^T) and the level-3 parts of dpotrf / dtrtri on one block.
//
//   C[m][n] = beta * C[m][n] + alpha * sum_k a(m,k) * b(k,n)
//   a(m,k) = A_T ? A[k*lda + m] : A[m*lda + k]        (A_T: A is stored K x M)
//   b(k,n) = B_N ? B[k*ldb + n] : B[n*ldb + k]        (B_N: B is stored K x N, else N x K)
//
// 64x64 output tile per 256-thread workgroup (4 waves, each a 32x32 quadrant = 2x2 MFMA
// tiles), K stepped by 16 through a double-buffered, padded LDS image [row][k] so that the
// MFMA operand reads (lane l: row l&15, k l>>4) are bank-conflict free for ds_read_b64
// (row stride 18 doubles: 18*i mod 32 is a permutation of the even residues).
// Triangular operands skip the K range that is structurally zero (the zero part inside
// the boundary tile must hold real zeros).  All of M, N multiples of 64, K of 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

enum : int {
    TRI_A_LOWER = 1,   // a(m,k) == 0 for k > m
    TRI_A_UPPER = 2,   // a(m,k) == 0 for k < m
    TRI_B_LOWER = 4,   // b(k,n) == 0 for k < n
    TRI_B_UPPER = 8    // b(k,n) == 0 for k > n
};

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;
    int64_t strideA, strideB, strideC;   // batch strides (blockIdx.z), in elements
    int M, N, K;
    int tri;
    int lower_only;                      // skip tiles strictly above the block diagonal
    double alpha, beta;
};

constexpr int GEMM_BM = 64;
constexpr int GEMM_BN = 64;
constexpr int GEMM_BK = 16;
constexpr int GEMM_LD = 18;              // padded LDS row stride in doubles

// One 64(row) x 16(k) operand tile: global -> registers (two v2d per thread).
template <bool ROWS_CONTIG>
__device__ __forceinline__ void gemm_load_tile(const double* __restrict__ P, int64_t ld,
                                               int row0, int k0, int t, v2d& r0, v2d& r1) {
    if (!ROWS_CONTIG) {          // stored [row][k]: 8 threads cover one row's 16 k
        const int r = t >> 3, kk = (t & 7) * 2;
        const double* p = P + (int64_t)(row0 + r) * ld + k0 + kk;
        r0 = *reinterpret_cast<const v2d*>(p);
        r1 = *reinterpret_cast<const v2d*>(p + 32 * ld);
    } else {                     // stored [k][row]: 32 threads cover one k's 64 rows
        const int kk = t >> 5, r = (t & 31) * 2;
        const double* p = P + (int64_t)(k0 + kk) * ld + row0 + r;
        r0 = *reinterpret_cast<const v2d*>(p);
        r1 = *reinterpret_cast<const v2d*>(p + 8 * ld);
    }
}

template <bool ROWS_CONTIG>
__device__ __forceinline__ void gemm_store_tile(double* sm, int t, const v2d& r0, const v2d& r1) {
    if (!ROWS_CONTIG) {
        const int r = t >> 3, kk = (t & 7) * 2;
        *reinterpret_cast<v2d*>(sm + r * GEMM_LD + kk) = r0;
        *reinterpret_cast<v2d*>(sm + (r + 32) * GEMM_LD + kk) = r1;
    } else {
        const int kk = t >> 5, r = (t & 31) * 2;
        sm[r * GEMM_LD + kk] = r0.x;
        sm[(r + 1) * GEMM_LD + kk] = r0.y;
        sm[r * GEMM_LD + kk + 8] = r1.x;
        sm[(r + 1) * GEMM_LD + kk + 8] = r1.y;
    }
}

template <bool A_T, bool B_N>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma(GemmArgs g) {
    const int bm = blockIdx.y, bn = blockIdx.x;
    if (g.lower_only && bn > bm) return;
    const int m0 = bm * GEMM_BM, n0 = bn * GEMM_BN;
    const double* __restrict__ A = g.A + (int64_t)blockIdx.z * g.strideA;
    const double* __restrict__ B = g.B + (int64_t)blockIdx.z * g.strideB;
    double* __restrict__ C = g.C + (int64_t)blockIdx.z * g.strideC;

    int kb = 0, ke = g.K;
    if (g.tri & TRI_A_LOWER) ke = min(ke, m0 + GEMM_BM);
    if (g.tri & TRI_A_UPPER) kb = max(kb, m0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, n0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, n0 + GEMM_BN);

    __shared__ __attribute__((aligned(16))) double As[2][GEMM_BM * GEMM_LD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GEMM_BN * GEMM_LD];

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int li = lane & 15, lq = lane >> 4;

    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int nkt = (ke > kb) ? (ke - kb) / GEMM_BK : 0;
    v2d ra0, ra1, rb0, rb1;
    if (nkt > 0) {
        gemm_load_tile<A_T>(A, g.lda, m0, kb, t, ra0, ra1);
        gemm_load_tile<B_N>(B, g.ldb, n0, kb, t, rb0, rb1);
        gemm_store_tile<A_T>(As[0], t, ra0, ra1);
        gemm_store_tile<B_N>(Bs[0], t, rb0, rb1);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nkt);
        if (more) {
            const int k0 = kb + (kt + 1) * GEMM_BK;
            gemm_load_tile<A_T>(A, g.lda, m0, k0, t, ra0, ra1);
            gemm_load_tile<B_N>(B, g.ldb, n0, k0, t, rb0, rb1);
        }
        const double* as = As[cur];
        const double* bs = Bs[cur];
#pragma unroll
        for (int ks = 0; ks < GEMM_BK / 4; ++ks) {
            const double a0 = as[(wm + li) * GEMM_LD + ks * 4 + lq];
            const double a1 = as[(wm + 16 + li) * GEMM_LD + ks * 4 + lq];
            const double b0 = bs[(wn + li) * GEMM_LD + ks * 4 + lq];
            const double b1 = bs[(wn + 16 + li) * GEMM_LD + ks * 4 + lq];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            gemm_store_tile<A_T>(As[cur ^ 1], t, ra0, ra1);
            gemm_store_tile<B_N>(Bs[cur ^ 1], t, rb0, rb1);
        }
        __syncthreads();
    }

    // f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + lq + 4 * r;
                const int col = n0 + wn + j * 16 + li;
                double* c = C + (int64_t)row * g.ldc + col;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * (*c);
                *c = v;
            }
}

// Host-side launcher.  tri/lower_only semantics as in GemmArgs.
inline hipError_t launch_gemm(hipStream_t st, bool a_t, bool b_n, const GemmArgs& g, int batch) {
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return hipSuccess;
    dim3 grid(g.N / GEMM_BN, g.M / GEMM_BM, batch), block(256);
    if (!a_t && !b_n) hipLaunchKernelGGL((gemm_f64_mfma<false, false>), grid, block, 0, st, g);
    else if (!a_t && b_n) hipLaunchKernelGGL((gemm_f64_mfma<false, true>), grid, block, 0, st, g);
    else if (a_t && !b_n) hipLaunchKernelGGL((gemm_f64_mfma<true, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f64_mfma<true, true>), grid, block, 0, st, g);
    return hipGetLastError();
}

}  // namespace gmrf
