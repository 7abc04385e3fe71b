// Batches, round 4, OPT-IN (set_eager bit 14 / GMRF_GEMM128=0): the 128^3 products of a 256-column diagonal panel without the
// GEMM kernel.  Built, parity-green, measured SLOWER than the GEMM launches it replaces -- kept as the record of the experiment.
//
// A 256-column panel P = [A 0; BA B] of the in-block Cholesky of a batch (potrf_block, gmrf_hip.hip) costs, between and after
// its two potrf_diag128 launches, four products of 128 x 128 x 128 per problem:
//     L_BA = S_BA X_A^T,   S_BB -= L_BA L_BA^T          (before potrf_diag128(B))
//     T = L_BA X_A,        X_BA = -X_B T                (after it: the level-128 doubling step of the panel's inverse)
// As GEMM launches they are 4 x 256 launches of 256 workgroups per batch-64 factorisation: 9.94 ms = 13 % of the GEMM time of
// a step at 20.7 TF/s (profiles/r03_bench_line_default.json, gemm_by_shape) -- four 64 x 64 tiles per problem cannot fill a
// launch.  Here ONE workgroup per problem does a pair of them, tile product after tile product (twelve 64^3 products per
// launch, operands staged through three LDS tiles), like potrf_diag128 beside it.  Same mathematics as the GEMM route (products
// from zero, one subtraction / one negation) and the same k order inside every product: the two routes agree BITWISE
// (test_two_level_panel_factor_of_batches).  Measured in one gpurun call (darcy256, 4 streams x batch 64): the time-weighted
// GEMM fraction rises 0.612 -> 0.659 (the slow launches are gone from the GEMM classes) and the job FALLS 48.1 k -> 45.6 k solves/s:
// twelve dependent products with their tile loads take one CU 57 us per launch (29 ms per batch factorisation and handle)
// against 39 us for the four GEMM launches, which spread each product over four CUs; a stream's factor phase 90.9 -> 109.2 ms.
// What would be needed to win: the products of a panel split over 2 - 4 workgroups with flag hand-offs (the potrf_persist
// machinery), with slim workgroups that leave the CU to other streams' GEMMs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "potrf_step.hpp"

namespace gmrf {

struct Panel256Args {
    double* S; double* L; double* X;
    int64_t ld;
    int j;                  // first tile of the panel: A = tiles j, j+1, B = tiles j+2, j+3
    int64_t pS, pL, pX;
    double* Lba;            // where L[j+2 .., j .. j+1] lives (the factor's own block, or its slot in the split inverse) ...
    int64_t ldl, pLba;      // ... as a 128 x 128 window with this row stride / problem stride
};

constexpr size_t PANEL256_LDS = 3 * TILE_ELEMS * sizeof(double);

// acc[Jb] += A_strip X[16 Jb ..][.]^T with X a lower-triangular tile (b(k, n) = X[n][k] = 0 for k > n): k groups 0 .. 2 Jb + 1
__device__ __forceinline__ void strip_nt_xlow(const double* arow, const double* Xs, v4d (&acc)[4], int li, int lq) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(arow + k);
#pragma unroll
        for (int Jb = kg / 2; Jb < 4; ++Jb) {
            const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, acc[Jb], 0, 0, 0);
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, acc[Jb], 0, 0, 0);
        }
    }
}
// acc[Jb] += A_strip B with B stored [k][n]; LOW: B lower triangular (zero for k < n): k groups 2 Jb .. 7
template <bool LOW>
__device__ __forceinline__ void strip_nn(const double* arow, const double* Bs, v4d (&acc)[4], int li, int lq) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(arow + k);
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) {
            if (!LOW || kg >= 2 * Jb) {
                acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, Bs[k * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
                acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, Bs[(k + 1) * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
            }
        }
    }
}
// this wave's 16-row strip of a 64 x 64 tile in the MFMA C/D layout <-> global memory (plain accesses)
__device__ __forceinline__ void strip_store(double* g, int64_t ld, const v4d (&t)[4], bool negate, int wave, int li, int lq) {
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[(int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li] = negate ? -t[Jb][q] : t[Jb][q];
}
// g[strip] -= t for the column blocks Jb < jb_end
__device__ __forceinline__ void strip_sub(double* g, int64_t ld, const v4d (&t)[4], int jb_end, int wave, int li, int lq) {
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (Jb < jb_end) {
                double* p = g + (int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li;
                *p = *p - t[Jb][q];
            }
}
__device__ __forceinline__ void zero4(v4d (&a)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// MODE 0: L_BA = S_BA X_A^T and S_BB -= L_BA L_BA^T.   MODE 1: X_BA = -X_B (L_BA X_A).   grid (1, problems), 256 threads.
template <int MODE>
__global__ __launch_bounds__(256, 1) void potrf_panel256(Panel256Args pa) {
    pa.S += (int64_t)blockIdx.y * pa.pS;
    pa.L += (int64_t)blockIdx.y * pa.pL;
    pa.X += (int64_t)blockIdx.y * pa.pX;
    pa.Lba += (int64_t)blockIdx.y * pa.pLba;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = As + TILE_ELEMS;
    double* Cs = Bs + TILE_ELEMS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = pa.ld, ldl = pa.ldl;
    const int64_t oa = (int64_t)pa.j * 64, ob = oa + 128;
    auto Xt = [&](int r, int c) { return pa.X + (oa + 64 * r) * ld + oa + 64 * c; };     // tile (r, c) of the panel's 256 x 256 inverse
    auto St = [&](int r, int c) { return pa.S + (oa + 64 * r) * ld + oa + 64 * c; };
    auto Lt = [&](int r, int c) { return pa.Lba + (int64_t)(64 * (r - 2)) * ldl + 64 * c; };   // tile (r, c), r in {2, 3}, c in {0, 1}, of L_BA
    const double* arow_a = As + (16 * wave + li) * TLD;
    const double* arow_c = Cs + (16 * wave + li) * TLD;
    (void)ob;
    if (MODE == 0) {
        // ---- L[r, 0] = S[r, 0] X00^T,  L[r, 1] = S[r, 0] X10^T + S[r, 1] X11^T        (r = 2, 3)
        for (int r = 2; r < 4; ++r) {
            tile_g2s(St(r, 0), ld, As, tid);
            tile_g2s(Xt(0, 0), ld, Bs, tid);
            tile_g2s(Xt(1, 0), ld, Cs, tid);
            __syncthreads();
            v4d l0[4], l1[4];
            zero4(l0); zero4(l1);
            strip_nt_xlow(arow_a, Bs, l0, li, lq);
            strip_nt<4>(arow_a, Cs, l1, li, lq);
            __syncthreads();
            tile_g2s(St(r, 1), ld, As, tid);
            tile_g2s(Xt(1, 1), ld, Bs, tid);
            __syncthreads();
            strip_nt_xlow(arow_a, Bs, l1, li, lq);
            strip_store(Lt(r, 0), ldl, l0, false, wave, li, lq);
            strip_store(Lt(r, 1), ldl, l1, false, wave, li, lq);
            __syncthreads();                               // (the next loads overwrite As / Bs; L is re-read below)
        }
        // ---- S_BB -= L_BA L_BA^T on the lower blocks: tiles (2,2), (3,2), (3,3); products from zero, one subtraction
        v4d a22[4], a32[4], a33[4];
        zero4(a22); zero4(a32); zero4(a33);
        for (int c = 0; c < 2; ++c) {
            tile_g2s(Lt(2, c), ldl, As, tid);
            tile_g2s(Lt(3, c), ldl, Cs, tid);
            __syncthreads();
            strip_nt_diag(wave, arow_a, As, a22, li, lq);
            strip_nt<4>(arow_c, As, a32, li, lq);
            strip_nt_diag(wave, arow_c, Cs, a33, li, lq);
            __syncthreads();
        }
        strip_sub(St(2, 2), ld, a22, wave + 1, wave, li, lq);
        strip_sub(St(3, 2), ld, a32, 4, wave, li, lq);
        strip_sub(St(3, 3), ld, a33, wave + 1, wave, li, lq);
    } else {
        // ---- T[r, c] = L[r, 0] X[0, c] + L[r, 1] X[1, c]  (X[0, 1] = 0),  X[2, c] = -X22 T[2, c],  X[3, c] = -(X32 T[2, c] + X33 T[3, c])
        for (int c = 0; c < 2; ++c) {
            v4d t2[4], t3[4];
            zero4(t2); zero4(t3);
            if (c == 0) {
                tile_g2s(Lt(2, 0), ldl, As, tid);
                tile_g2s(Lt(3, 0), ldl, Cs, tid);
                tile_g2s(Xt(0, 0), ld, Bs, tid);
                __syncthreads();
                strip_nn<true>(arow_a, Bs, t2, li, lq);
                strip_nn<true>(arow_c, Bs, t3, li, lq);
                __syncthreads();
                tile_g2s(Lt(2, 1), ldl, As, tid);
                tile_g2s(Lt(3, 1), ldl, Cs, tid);
                tile_g2s(Xt(1, 0), ld, Bs, tid);
                __syncthreads();
                strip_nn<false>(arow_a, Bs, t2, li, lq);
                strip_nn<false>(arow_c, Bs, t3, li, lq);
            } else {
                tile_g2s(Lt(2, 1), ldl, As, tid);
                tile_g2s(Lt(3, 1), ldl, Cs, tid);
                tile_g2s(Xt(1, 1), ld, Bs, tid);
                __syncthreads();
                strip_nn<true>(arow_a, Bs, t2, li, lq);
                strip_nn<true>(arow_c, Bs, t3, li, lq);
            }
            __syncthreads();
            // T[2, c] -> Bs as a [k][n] image; X22 -> As, X32 -> Cs
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb) store_d16(Bs + (16 * wave) * TLD + 16 * Jb, TLD, t2[Jb], li, lq);
            tile_g2s(Xt(2, 2), ld, As, tid);
            tile_g2s(Xt(3, 2), ld, Cs, tid);
            __syncthreads();
            v4d r2[4], r3[4];
            zero4(r2); zero4(r3);
            strip_tri_nn_w(wave, arow_a, Bs, r2, li, lq);
            strip_nn<false>(arow_c, Bs, r3, li, lq);
            strip_store(Xt(2, c), ld, r2, true, wave, li, lq);
            __syncthreads();
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb) store_d16(Bs + (16 * wave) * TLD + 16 * Jb, TLD, t3[Jb], li, lq);
            tile_g2s(Xt(3, 3), ld, As, tid);
            __syncthreads();
            strip_tri_nn_w(wave, arow_a, Bs, r3, li, lq);
            strip_store(Xt(3, c), ld, r3, true, wave, li, lq);
            __syncthreads();
        }
    }
}

}  // namespace gmrf
