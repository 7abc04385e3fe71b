// Dense fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// This is the workhorse of the block factorisation: it replaces the dtrsm / dsyrk / dgemm
// calls Julia's LinearAlgebra makes for /root/reference/src/tridiagonal_cholesky.jl:74,77
// (C = B L^-T, D - C C^T) and the level-3 parts of dpotrf / dtrtri on one block.
//
//   C[m][n] = beta * D[m][n] + alpha * sum_k a(m,k) * b(k,n)         (D = C unless given)
//   a(m,k) = A_T ? A[k*lda + m] : A[m*lda + k]        (A_T: A is stored K x M)
//   b(k,n) = B_N ? B[k*ldb + n] : B[n*ldb + k]        (B_N: B is stored K x N, else N x K)
//
// 64x64 output tile per 256-thread workgroup (4 waves, each a 32x32 quadrant = 2x2 MFMA
// tiles), K stepped by 32 (16 when K is not a multiple of 32) through a double-buffered,
// padded LDS image [row][k] (details at the kernel).
// Triangular operands skip the K range that is structurally zero (the zero part inside
// the boundary tile must hold real zeros).  All of M, N multiples of 64, K of 16.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>

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
    int64_t strideA, strideB, strideC;   // inner batch strides (blockIdx.z % nb1), in elements
    int64_t pA, pB, pC;                  // outer (problem) strides (blockIdx.z / nb1)
    int nb1;                             // inner batch count (>= 1)
    int M, N, K;
    int tri;
    int lower_only;                      // skip tiles strictly above the block diagonal (launcher: 1 = triangular grid, 2 = early exit)
    double alpha, beta;
    const double* D;                     // addend: C = beta * D + alpha * A B  (nullptr: D = C, in place)
    int64_t ldd, pD;                     // its row stride and problem stride
    unsigned long long* stamps;          // diagnostic (tests): s_memtime / s_memrealtime of block 0
    // Optional per-tile K bounds (device arrays indexed by 64-wide tile, values multiples of 64): the
    // staircase of the coupling blocks C_i (row tile t of C is zero left of column kst[t]; column
    // tile u is zero below row mend[u]).  K starts at max(kb_m[bm], kb_n[bn]) and ends at ke_n[bn].
    const int* kb_m = nullptr;
    const int* kb_n = nullptr;
    const int* ke_n = nullptr;
};

constexpr int GEMM_BM = 64;
constexpr int GEMM_BN = 64;

// One 64(row) x BK(k) operand tile: global -> registers (BK/8 v2d per thread).
template <bool ROWS_CONTIG, int BK>
__device__ __forceinline__ void gemm_load_tile(const double* __restrict__ P, int64_t ld, int row0, int k0,
                                               int t, v2d (&rg)[BK / 8]) {
    if (!ROWS_CONTIG) {          // stored [row][k]: BK/2 threads cover one row
        constexpr int TPR = BK / 2, RPP = 256 / TPR;      // threads per row, rows per pass
        const int r = t / TPR, kk = (t % TPR) * 2;
        const double* p = P + (int64_t)(row0 + r) * ld + k0 + kk;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) rg[i] = *reinterpret_cast<const v2d*>(p + (int64_t)(i * RPP) * ld);
    } else {                     // stored [k][row]: 32 threads cover one k's 64 rows
        const int kk = t >> 5, r = (t & 31) * 2;
        const double* p = P + (int64_t)(k0 + kk) * ld + row0 + r;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) rg[i] = *reinterpret_cast<const v2d*>(p + (int64_t)(i * 8) * ld);
    }
}

// The same loads as gemm_load_tile, split into a per-thread 32-bit byte offset computed ONCE per output tile
// (rows and the k position inside a K step) and a wave-uniform base that advances per K step: the loads take the
// scalar-base + 32-bit-offset form and the K loop carries no 64-bit address arithmetic (it was 22 VALU
// instructions per K step beside 32 MFMAs).  Offsets stay below 2^32 for blocks up to 16384 x 16384.
template <bool ROWS_CONTIG, int BK>
__device__ __forceinline__ void gemm_tile_offsets(int64_t ld, int row0, int t, uint32_t (&off)[BK / 8]) {
    if (!ROWS_CONTIG) {
        constexpr int TPR = BK / 2, RPP = 256 / TPR;
        const int r = t / TPR, kk = (t % TPR) * 2;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) off[i] = (uint32_t)(((int64_t)(row0 + r + i * RPP) * ld + kk) * 8);
    } else {
        const int kk = t >> 5, r = (t & 31) * 2;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) off[i] = (uint32_t)(((int64_t)(kk + i * 8) * ld + row0 + r) * 8);
    }
}
template <bool ROWS_CONTIG>
__device__ __forceinline__ const char* gemm_tile_base(const double* P, int64_t ld, int k0) {
    return reinterpret_cast<const char*>(ROWS_CONTIG ? P + (int64_t)k0 * ld : P + k0);
}
template <int BK>
__device__ __forceinline__ void gemm_load_tile_off(const char* base, uint32_t (&off)[BK / 8], v2d (&rg)[BK / 8]) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        // opaque to the optimiser at this point: otherwise the zero-extension of the offset is hoisted out of the K
        // loop as a 64-bit VGPR pair and instruction selection (per basic block) no longer sees base + zext(offset)
        asm volatile("" : "+v"(off[i]));
        rg[i] = *reinterpret_cast<const v2d*>(base + off[i]);
    }
}

// [k][row]-stored operand kept as it lies in memory: LDS image [BK][64], straight 16-byte copies (a half wave
// writes one 512-byte k row: no bank conflicts; the transposing gemm_store_tile<true> puts lanes 4 apart on the
// same banks -- SQ_LDS_BANK_CONFLICT was 0.64 of the LDS-active cycles of gemm_f64_mfma<false, true>).
template <int BK>
__device__ __forceinline__ void gemm_store_tile_natural(double* sm, int t, const v2d (&rg)[BK / 8]) {
    const int kk = t >> 5, r = (t & 31) * 2;
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) *reinterpret_cast<v2d*>(sm + (kk + 8 * i) * 64 + r) = rg[i];
}

template <bool ROWS_CONTIG, int BK>
__device__ __forceinline__ void gemm_store_tile(double* sm, int t, const v2d (&rg)[BK / 8]) {
    constexpr int LD = BK + 4;
    if (!ROWS_CONTIG) {
        constexpr int TPR = BK / 2, RPP = 256 / TPR;
        const int r = t / TPR, kk = (t % TPR) * 2;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) *reinterpret_cast<v2d*>(sm + (r + i * RPP) * LD + kk) = rg[i];
    } else {
        const int kk = t >> 5, r = (t & 31) * 2;
#pragma unroll
        for (int i = 0; i < BK / 8; ++i) {
            sm[r * LD + kk + 8 * i] = rg[i].x;
            sm[(r + 1) * LD + kk + 8 * i] = rg[i].y;
        }
    }
}

// Tile order over a 1-D grid (both kernels).  Workgroup ids go round-robin over the 8 XCDs, so
// with a multiple of 8 problems id % 8 picks the problem group: every XCD works on whole
// problems and their operand panels are shared in ITS L2.  Inside a group the tiles with the
// longest K range (triangular operands) are issued first and the short ones fill the tail.
template <int BT>
__device__ __forceinline__ void gemm_tile_order(const GemmArgs& g, int& bm, int& bn, int& z) {
    const int nx = g.N / BT, ny = g.M / BT;
    const int tpp = g.lower_only == 1 ? nx * (nx + 1) / 2 : nx * ny;
    const int nz = (int)gridDim.x / tpp;
    const int groups = (nz % 8 == 0) ? 8 : 1;
    const int xg = (int)blockIdx.x % groups, q = (int)blockIdx.x / groups, nzg = nz / groups;
    int zq;
    if (g.lower_only == 1) {
        const int tile = q % tpp;
        zq = q / tpp;
        bm = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
        while (bm * (bm + 1) / 2 > tile) --bm;
        while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
        bn = tile - bm * (bm + 1) / 2;
    } else {
        const bool cls_n = (g.tri & (TRI_B_LOWER | TRI_B_UPPER)) || !(g.tri & (TRI_A_LOWER | TRI_A_UPPER));
        const bool desc = cls_n ? ((g.tri & TRI_B_UPPER) != 0 || (g.ke_n && !g.kb_n)) : (g.tri & TRI_A_LOWER) != 0;
        const int ncls = cls_n ? nx : ny, other = cls_n ? ny : nx;
        int c = q / (other * nzg);
        const int rem = q % (other * nzg);
        const int o = rem % other;
        zq = rem / other;
        if (desc) c = ncls - 1 - c;
        bn = cls_n ? c : o;
        bm = cls_n ? o : c;
    }
    z = xg + groups * zq;
}

// K is stepped by BK (16 or 32) through a double-buffered LDS image [row][k] with row stride
// BK + 4 doubles.  A lane fetches TWO consecutive k (one ds_read_b128, conflict free at this
// stride for the 4 x 16-lane groups of that instruction) and feeds them to two MFMAs: MFMA 1
// sums k in {0,2,4,6} of an 8-wide k group, MFMA 2 the odd ones -- any assignment of k to the
// instruction's four k slots is valid as long as A and B use the same one.  The fragments of
// the next k group are requested before the MFMAs of the current one are issued, so the LDS
// latency hides behind 8 x 64 MFMA cycles.  BK = 32 gives every global load 32 MFMAs (2048
// cycles) per wave to land before it is needed.
struct GemmFrag { v2d a0, a1, b0, b1; };

// B_NAT: the B image is [k][64] (gemm_store_tile_natural).  A lane then reads the two neighbouring columns
// wn + 2 li, wn + 2 li + 1 of rows k and k + 1: b0 = row k, b1 = row k + 1, .x feeds the MFMA tile of the even
// columns of the wave's 32-column range, .y the odd ones (the two output tiles interleave; same k slots per
// MFMA as the [n][k] image, hence bitwise the same sums).  A 16-lane group reads 256 contiguous bytes.
template <int LD, bool B_NAT>
__device__ __forceinline__ GemmFrag gemm_read_frag(const double* as, const double* bs, int wm, int wn, int li,
                                                   int lq, int kg) {
    GemmFrag f;
    const int k = kg * 8 + 2 * lq;
    f.a0 = *reinterpret_cast<const v2d*>(as + (wm + li) * LD + k);
    f.a1 = *reinterpret_cast<const v2d*>(as + (wm + 16 + li) * LD + k);
    if (B_NAT) {
        f.b0 = *reinterpret_cast<const v2d*>(bs + k * 64 + wn + 2 * li);
        f.b1 = *reinterpret_cast<const v2d*>(bs + (k + 1) * 64 + wn + 2 * li);
    } else {
        f.b0 = *reinterpret_cast<const v2d*>(bs + (wn + li) * LD + k);
        f.b1 = *reinterpret_cast<const v2d*>(bs + (wn + 16 + li) * LD + k);
    }
    return f;
}

template <bool A_T, bool B_N, int BK>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma(GemmArgs g) {
    constexpr int LD = BK + 4;
    int bm, bn, z;
    gemm_tile_order<GEMM_BM>(g, bm, bn, z);
    if (g.lower_only == 2 && bn > bm) return;
    const int m0 = bm * GEMM_BM, n0 = bn * GEMM_BN;
    const int zi = z % g.nb1, zp = z / g.nb1;
    const double* __restrict__ A = g.A + (int64_t)zi * g.strideA + (int64_t)zp * g.pA;
    const double* __restrict__ B = g.B + (int64_t)zi * g.strideB + (int64_t)zp * g.pB;
    double* C = g.C + (int64_t)zi * g.strideC + (int64_t)zp * g.pC;
    const double* Dm = g.D ? g.D + (int64_t)zp * g.pD : C;
    const int64_t ldd = g.D ? g.ldd : g.ldc;

    int kb = 0, ke = g.K;
    if (g.tri & TRI_A_LOWER) ke = min(ke, m0 + GEMM_BM);
    if (g.tri & TRI_A_UPPER) kb = max(kb, m0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, n0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, n0 + GEMM_BN);
    if (g.kb_m) kb = max(kb, g.kb_m[bm]);
    if (g.kb_n) kb = max(kb, g.kb_n[bn]);
    if (g.ke_n) ke = min(ke, g.ke_n[bn]);

    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const bool single = (g.tri & 2048) != 0;         // experiment: one LDS stage (half the LDS, twice the workgroups per CU)
    double* As0 = gsm;                               // [2][64 * LD]
    double* Bs0 = gsm + (single ? 1 : 2) * GEMM_BM * LD;            // [2][64 * LD]

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int li = lane & 15, lq = lane >> 4;

    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int nkt = (ke > kb) ? (ke - kb) / BK : 0;
    unsigned long long c0 = 0, r0 = 0;
    if (g.stamps) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    v2d ra[BK / 8], rb[BK / 8];
    uint32_t offa[BK / 8], offb[BK / 8];
    gemm_tile_offsets<A_T, BK>(g.lda, m0, t, offa);
    gemm_tile_offsets<B_N, BK>(g.ldb, n0, t, offb);
    if (nkt > 0) {
        gemm_load_tile_off<BK>(gemm_tile_base<A_T>(A, g.lda, kb), offa, ra);
        gemm_load_tile_off<BK>(gemm_tile_base<B_N>(B, g.ldb, kb), offb, rb);
        gemm_store_tile<A_T, BK>(As0, t, ra);
        if (B_N) gemm_store_tile_natural<BK>(Bs0, t, rb);
        else gemm_store_tile<false, BK>(Bs0, t, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = single ? 0 : (kt & 1);
        const int nxt = single ? 0 : (cur ^ 1);
        const bool more = (kt + 1 < nkt);
        if (more) {
            const int k0 = kb + (kt + 1) * BK;
            gemm_load_tile_off<BK>(gemm_tile_base<A_T>(A, g.lda, k0), offa, ra);
            gemm_load_tile_off<BK>(gemm_tile_base<B_N>(B, g.ldb, k0), offb, rb);
        }
        const double* as = As0 + cur * GEMM_BM * LD;
        const double* bs = Bs0 + cur * GEMM_BN * LD;
        // two fragment sets, alternating (indices are compile-time after unrolling: no register copies)
        GemmFrag fr[2];
        fr[0] = gemm_read_frag<LD, B_N>(as, bs, wm, wn, li, lq, 0);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            if (kg + 1 < BK / 8) fr[(kg + 1) & 1] = gemm_read_frag<LD, B_N>(as, bs, wm, wn, li, lq, kg + 1);
            const GemmFrag& f = fr[kg & 1];
            __builtin_amdgcn_sched_barrier(0);
            // operands of output tile j at k slot p: [n][k] image: column tile j = b_j, slot = .x / .y;
            // [k][n] image (B_N): row k + p = b_p, even / odd columns = .x / .y
            const double b00 = f.b0.x, b01 = B_N ? f.b0.y : f.b1.x, b10 = B_N ? f.b1.x : f.b0.y, b11 = f.b1.y;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a0.x, b00, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a0.x, b01, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a1.x, b00, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a1.x, b01, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a0.y, b10, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a0.y, b11, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a1.y, b10, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a1.y, b11, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (single) __syncthreads();
        if (more) {
            gemm_store_tile<A_T, BK>(As0 + nxt * GEMM_BM * LD, t, ra);
            if (B_N) gemm_store_tile_natural<BK>(Bs0 + nxt * GEMM_BN * LD, t, rb);
            else gemm_store_tile<false, BK>(Bs0 + nxt * GEMM_BN * LD, t, rb);
        }
        __syncthreads();
    }

    if (g.stamps && blockIdx.x == 0 && t == 0) {
        g.stamps[0] = __builtin_amdgcn_s_memtime() - c0;
        g.stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.
    const double alpha = g.alpha, beta = g.beta;
    if (B_N) {
        // the two tiles of a row group interleave: a lane owns the columns wn + 2 li, wn + 2 li + 1
        const bool vec = ((g.ldc | ldd) & 1) == 0 && ((((uintptr_t)C) | ((uintptr_t)Dm)) & 15) == 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + lq + 4 * r;
                const int col = n0 + wn + 2 * li;
                double* c = C + (int64_t)row * g.ldc + col;
                v2d v = (v2d){alpha * acc[i][0][r], alpha * acc[i][1][r]};
                if (vec) {
                    if (beta != 0.0) {
                        const v2d d = *reinterpret_cast<const v2d*>(Dm + (int64_t)row * ldd + col);
                        v.x += beta * d.x; v.y += beta * d.y;
                    }
                    *reinterpret_cast<v2d*>(c) = v;
                } else {
                    if (beta != 0.0) { v.x += beta * Dm[(int64_t)row * ldd + col]; v.y += beta * Dm[(int64_t)row * ldd + col + 1]; }
                    c[0] = v.x; c[1] = v.y;
                }
            }
        return;
    }
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
                if (beta != 0.0) v += beta * Dm[(int64_t)row * ldd + col];
                *c = v;
            }
}


// ---------------------------------------------------------------------------------------------
// 128 x 128 output tile per workgroup: 4 waves, each a 64 x 64 quadrant = 4 x 4 MFMA tiles
// (128 accumulator VGPRs).  Every LDS / global instruction issued beside the fp64 MFMA stream
// costs ~26 cycles of MFMA issue (measured, tools/mb4.hip), so the lever is instructions per
// MFMA: an 8-wide k group takes 8 ds_read_b128 for 32 MFMAs here against 4 for 8 in the 64 x 64
// kernel, and the staging traffic per MFMA halves as well.  K steps by 16, double buffered.
//
//   A is stored [m][k]  -> LDS [128][20] (k contiguous, same fragment scheme as above).
//   B_N = false: B stored [n][k] -> LDS [128][20], tile j of a wave = columns 16 j + (lane & 15).
//   B_N = true : B stored [k][n] -> LDS [16][128] as it lies in memory (straight b128 copies,
//                no transposing ds_write_b64).  A lane reads the two neighbouring columns
//                2 li, 2 li + 1 of row k = 8 kg + 2 lq + p: .x feeds the MFMA of the even
//                columns of a 32-column group, .y the odd ones -- the output tiles 2 jp and
//                2 jp + 1 interleave and are written back as one 16-byte store per lane.
//                The four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) each cover
//                one full 256-byte row image (rows 2 lq apart, 2 * 128 doubles = 0 mod 256 B).
// Used when M, N are multiples of 128, A is not transposed and the launch has enough tiles.
constexpr int GEMM_BIG = 128;
constexpr int GEMM_BIG_BK = 16;
constexpr int GEMM_BIG_LDK = GEMM_BIG_BK + 4;

template <bool B_N>
constexpr size_t gemm_big_lds_bytes() {
    return (size_t)2 * (GEMM_BIG * GEMM_BIG_LDK + (B_N ? GEMM_BIG_BK * GEMM_BIG : GEMM_BIG * GEMM_BIG_LDK)) * sizeof(double);
}

template <bool B_N>
__global__ __launch_bounds__(256, 2) void gemm_f64_big(GemmArgs g) {
    constexpr int BT = GEMM_BIG, BK = GEMM_BIG_BK, LDK = GEMM_BIG_LDK;
    constexpr int A_STAGE = BT * LDK;
    constexpr int B_STAGE = B_N ? BK * BT : BT * LDK;
    int bm, bn, z;
    gemm_tile_order<BT>(g, bm, bn, z);
    const int m0 = bm * BT, n0 = bn * BT;
    const int zi = z % g.nb1, zp = z / g.nb1;
    const double* __restrict__ A = g.A + (int64_t)zi * g.strideA + (int64_t)zp * g.pA;
    const double* __restrict__ B = g.B + (int64_t)zi * g.strideB + (int64_t)zp * g.pB;
    double* C = g.C + (int64_t)zi * g.strideC + (int64_t)zp * g.pC;
    const double* Dm = g.D ? g.D + (int64_t)zp * g.pD : C;
    const int64_t ldd = g.D ? g.ldd : g.ldc;

    int kb = 0, ke = g.K;
    if (g.tri & TRI_A_LOWER) ke = min(ke, m0 + BT);
    if (g.tri & TRI_A_UPPER) kb = max(kb, m0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, n0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, n0 + BT);
    // staircase bounds are kept per 64-wide tile and are monotone: a 128-wide tile starts at its first half's
    // bound and ends at its second half's
    if (g.kb_m) kb = max(kb, g.kb_m[2 * bm]);
    if (g.kb_n) kb = max(kb, g.kb_n[2 * bn]);
    if (g.ke_n) ke = min(ke, g.ke_n[2 * bn + 1]);

    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double* As0 = gsm;                      // [2][A_STAGE]
    double* Bs0 = gsm + 2 * A_STAGE;        // [2][B_STAGE]

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int li = lane & 15, lq = lane >> 4;

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // staging: [row][k] operands -- 8 threads per row, 32 rows per pass, 4 passes;
    //          [k][n] operand   -- one wave per k row (64 lanes x 16 B), 4 rows per wave.
    const int sr = t >> 3, sk = (t & 7) * 2;
    const double* ap = A + (int64_t)(m0 + sr) * g.lda + sk;
    const double* bp = B_N ? B + (int64_t)w * g.ldb + n0 + 2 * lane : B + (int64_t)(n0 + sr) * g.ldb + sk;
    const int64_t a_step = 32 * g.lda, b_step = B_N ? 4 * g.ldb : 32 * g.ldb;
    double* a_st = As0 + sr * LDK + sk;
    double* b_st = B_N ? Bs0 + w * BT + 2 * lane : Bs0 + sr * LDK + sk;
    constexpr int A_ST_STEP = 32 * LDK, B_ST_STEP = B_N ? 4 * BT : 32 * LDK;

    const int nkt = (ke > kb) ? (ke - kb) / BK : 0;
    v2d ra[4], rb[4];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const v2d*>(ap + k0 + i * a_step);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rb[i] = *reinterpret_cast<const v2d*>(B_N ? bp + (int64_t)k0 * g.ldb + i * b_step : bp + k0 + i * b_step);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(a_st + buf * A_STAGE + i * A_ST_STEP) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(b_st + buf * B_STAGE + i * B_ST_STEP) = rb[i];
    };
    if (nkt > 0) {
        load_stage(kb);
        store_stage(0);
    }
    __syncthreads();

    const int a_frag = (wm + li) * LDK + 2 * lq;
    const int b_frag = B_N ? (2 * lq) * BT + wn + 2 * li : (wn + li) * LDK + 2 * lq;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nkt);
        if (more) load_stage(kb + (kt + 1) * BK);
        const double* as = As0 + cur * A_STAGE + a_frag;
        const double* bs = Bs0 + cur * B_STAGE + b_frag;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            v2d fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v2d*>(as + i * 16 * LDK + kg * 8);
            if (B_N) {
                // fb[2 jp + p] = columns (2 li, 2 li + 1) of group jp at k parity p
#pragma unroll
                for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        fb[2 * jp + p] = *reinterpret_cast<const v2d*>(bs + (kg * 8 + p) * BT + jp * 32);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const v2d*>(bs + j * 16 * LDK + kg * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double av = p ? fa[i].y : fa[i].x;
                        double bv;
                        if (B_N) bv = (j & 1) ? fb[2 * (j >> 1) + p].y : fb[2 * (j >> 1) + p].x;
                        else bv = p ? fb[j].y : fb[j].x;
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_stage(cur ^ 1);
        __syncthreads();
    }

    // f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm + i * 16 + lq + 4 * r;
            double* crow = C + (int64_t)row * g.ldc + n0 + wn;
            const double* drow = Dm + (int64_t)row * ldd + n0 + wn;
            if (B_N) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    v2d v = (v2d){alpha * acc[i][2 * jp][r], alpha * acc[i][2 * jp + 1][r]};
                    if (beta != 0.0) {
                        const v2d o = *reinterpret_cast<const v2d*>(drow + jp * 32 + 2 * li);
                        v.x += beta * o.x; v.y += beta * o.y;
                    }
                    *reinterpret_cast<v2d*>(crow + jp * 32 + 2 * li) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v += beta * drow[j * 16 + li];
                    crow[j * 16 + li] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Low-latency kernel for launches that leave most CUs idle (one problem: 64 .. 78 tiles of 64 x 64
// on 256 CUs, every tile a serial K loop bound by ONE CU's matrix pipes): 32 x 32 output tiles, so
// the same product spreads over four times as many CUs; 4 waves, one 16 x 16 MFMA tile each.
// A K step is then ~0.25 us of MFMAs, far less than a memory latency: operand tiles are requested
// THREE steps ahead into a ring of three register sets and reach the other LDS buffer one step
// ahead.  Same fragment scheme and summation order per output element as gemm_f64_mfma (bitwise
// identical results; skipped K ranges hold exact zeros).  K must be a multiple of 32.
constexpr int GEMM_LL = 32;

template <bool A_T, bool B_N>
__global__ __launch_bounds__(256, 2) void gemm_f64_ll(GemmArgs g) {
    constexpr int BT = GEMM_LL, BK = 32, LD = BK + 4;
    int bm, bn, z;
    gemm_tile_order<BT>(g, bm, bn, z);
    if (g.lower_only == 2 && bn > bm) return;
    const int m0 = bm * BT, n0 = bn * BT;
    const int zi = z % g.nb1, zp = z / g.nb1;
    const double* __restrict__ A = g.A + (int64_t)zi * g.strideA + (int64_t)zp * g.pA;
    const double* __restrict__ B = g.B + (int64_t)zi * g.strideB + (int64_t)zp * g.pB;
    double* C = g.C + (int64_t)zi * g.strideC + (int64_t)zp * g.pC;
    const double* Dm = g.D ? g.D + (int64_t)zp * g.pD : C;
    const int64_t ldd = g.D ? g.ldd : g.ldc;

    int kb = 0, ke = g.K;
    if (g.tri & TRI_A_LOWER) ke = min(ke, m0 + BT);
    if (g.tri & TRI_A_UPPER) kb = max(kb, m0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, n0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, n0 + BT);
    if (g.kb_m) kb = max(kb, g.kb_m[bm >> 1]);
    if (g.kb_n) kb = max(kb, g.kb_n[bn >> 1]);
    if (g.ke_n) ke = min(ke, g.ke_n[bn >> 1]);

    __shared__ __attribute__((aligned(16))) double As0[2 * BT * LD];
    __shared__ __attribute__((aligned(16))) double Bs0[2 * BT * LD];
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * 16, wn = (w & 1) * 16;
    const int li = lane & 15, lq = lane >> 4;

    // staging: 512 16-byte pieces per operand tile, two per thread
    auto load_tile = [&](bool rows_contig, const double* P, int64_t ld, int row0, int k0, v2d (&rg)[2]) {
        if (!rows_contig) {      // stored [row][k]: 16 threads per row, rows r and r + 16
            const double* p = P + (int64_t)(row0 + (t >> 4)) * ld + k0 + (t & 15) * 2;
            rg[0] = *reinterpret_cast<const v2d*>(p);
            rg[1] = *reinterpret_cast<const v2d*>(p + 16 * ld);
        } else {                 // stored [k][row]: 16 threads per k, k and k + 16
            const double* p = P + (int64_t)(k0 + (t >> 4)) * ld + row0 + (t & 15) * 2;
            rg[0] = *reinterpret_cast<const v2d*>(p);
            rg[1] = *reinterpret_cast<const v2d*>(p + 16 * ld);
        }
    };
    auto store_tile = [&](bool rows_contig, double* sm, const v2d (&rg)[2]) {
        if (!rows_contig) {
            double* q = sm + (t >> 4) * LD + (t & 15) * 2;
            *reinterpret_cast<v2d*>(q) = rg[0];
            *reinterpret_cast<v2d*>(q + 16 * LD) = rg[1];
        } else {
            const int kk = t >> 4, r = (t & 15) * 2;
            sm[r * LD + kk] = rg[0].x; sm[(r + 1) * LD + kk] = rg[0].y;
            sm[r * LD + kk + 16] = rg[1].x; sm[(r + 1) * LD + kk + 16] = rg[1].y;
        }
    };

    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const int nkt = (ke > kb) ? (ke - kb) / BK : 0;
    v2d ra[3][2], rb[3][2];
    auto request = [&](int stage, v2d (&a2)[2], v2d (&b2)[2]) {
        if (stage < nkt) {
            load_tile(A_T, A, g.lda, m0, kb + stage * BK, a2);
            load_tile(B_N, B, g.ldb, n0, kb + stage * BK, b2);
        }
    };
    request(0, ra[0], rb[0]);
    request(1, ra[1], rb[1]);
    request(2, ra[2], rb[2]);
    if (nkt > 0) {
        store_tile(A_T, As0, ra[0]);
        store_tile(B_N, Bs0, rb[0]);
    }
    __syncthreads();
    // step kt: request stage kt + 3 into the set stage kt came from, multiply buffer kt & 1,
    // write stage kt + 1 (set `nxt`) into the other buffer
    auto step = [&](int kt, v2d (&a_far)[2], v2d (&b_far)[2], v2d (&a_nxt)[2], v2d (&b_nxt)[2]) {
        const int cur = kt & 1;
        request(kt + 3, a_far, b_far);
        const double* as = As0 + cur * BT * LD + (wm + li) * LD + 2 * lq;
        const double* bs = Bs0 + cur * BT * LD + (wn + li) * LD + 2 * lq;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const v2d a = *reinterpret_cast<const v2d*>(as + kg * 8);
            const v2d b = *reinterpret_cast<const v2d*>(bs + kg * 8);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc, 0, 0, 0);
        }
        if (kt + 1 < nkt) {
            store_tile(A_T, As0 + (cur ^ 1) * BT * LD, a_nxt);
            store_tile(B_N, Bs0 + (cur ^ 1) * BT * LD, b_nxt);
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nkt; kt += 3) {
        step(kt, ra[0], rb[0], ra[1], rb[1]);
        if (kt + 1 < nkt) step(kt + 1, ra[1], rb[1], ra[2], rb[2]);
        if (kt + 2 < nkt) step(kt + 2, ra[2], rb[2], ra[0], rb[0]);
    }
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + lq + 4 * r;
        const int col = n0 + wn + li;
        double v = alpha * acc[r];
        if (beta != 0.0) v += beta * Dm[(int64_t)row * ldd + col];
        C[(int64_t)row * g.ldc + col] = v;
    }
}

template <int BK>
constexpr size_t gemm_lds_bytes() { return (size_t)4 * 64 * (BK + 4) * sizeof(double); }

// Opt in to the > 64 KiB dynamic LDS of the BK = 32 variants (once per process).
inline hipError_t gemm_init() {
    hipError_t e = hipSuccess;
#define GMRF_GEMM_ATTR(AT, BN)                                                                   \
    if (e == hipSuccess)                                                                         \
        e = hipFuncSetAttribute((const void*)gemm_f64_mfma<AT, BN, 32>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes<32>());
    GMRF_GEMM_ATTR(false, false) GMRF_GEMM_ATTR(false, true) GMRF_GEMM_ATTR(true, false) GMRF_GEMM_ATTR(true, true)
#undef GMRF_GEMM_ATTR

    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_f64_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)gemm_big_lds_bytes<false>());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_f64_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)gemm_big_lds_bytes<true>());
    return e;
}

// Kernel choice.  0 (default): whichever kernel the makespan model below predicts faster;
// 1: always the 128 x 128 kernel where it applies; 2: never (tests and tools move it).
inline int& gemm_big_policy() {
    static int v = [] {
        const char* e = getenv("GMRF_GEMM_BIG_POLICY");   // tuning aid
        return e ? atoi(e) : 0;
    }();
    return v;
}

// Predicted duration (us) of a launch: the tiles of one XCD group, in the order gemm_tile_order
// issues them, are list-scheduled on the group's workgroup slots (2 per CU).  A tile costs its
// number of K steps times a per-step time measured on MI355X (tools/gemm_rate.py): 4.0 us per
// 16-wide step of the 128 x 128 kernel with two workgroups per CU, 2.5 us with one; 2.47 / 1.6 us
// per 32-wide step of the 64 x 64 kernel.  What decides is quantisation: 21 lower tiles x 32
// problems are 2 rounds of big tiles (382 us) but 4.9 rounds of small ones (298 us).
inline double gemm_estimate_us(bool big, const GemmArgs& g, int batch) {
    const int BT = big ? GEMM_BIG : GEMM_BM;
    const int BK = big ? GEMM_BIG_BK : (g.K % 32 == 0 ? 32 : 16);
    const int nx = g.N / BT, ny = g.M / BT;
    const bool tri_grid = g.lower_only && g.M == g.N;
    const int tpp = tri_grid ? nx * (nx + 1) / 2 : nx * ny;
    const int groups = (batch % 8 == 0) ? 8 : 1, nzg = batch / groups;
    const bool alone = (int64_t)tpp * batch <= 256;            // one workgroup per CU
    // (64 x 64 kernel re-measured after the round-2 changes to its K loop: 58-60 TF/s in steady state instead of 52-55)
    const double step = big ? (alone ? 2.5 : 4.0) : (BK == 32 ? (alone ? 1.45 : 2.2) : (alone ? 0.85 : 1.25));
    const int slots = 512 / groups;
    std::vector<double> freeat((size_t)slots, 0.0);              // min-heap of slot release times
    auto cmp = [](double a, double b) { return a > b; };
    double makespan = 0.0;
    auto place = [&](int nk) {
        std::pop_heap(freeat.begin(), freeat.end(), cmp);
        const double end = freeat.back() + 1.0 + nk * step;
        freeat.back() = end;
        std::push_heap(freeat.begin(), freeat.end(), cmp);
        makespan = std::max(makespan, end);
    };
    if (tri_grid) {
        for (int q = 0; q < tpp * nzg; ++q) place(g.K / BK);
    } else {
        const bool cls_n = (g.tri & (TRI_B_LOWER | TRI_B_UPPER)) || !(g.tri & (TRI_A_LOWER | TRI_A_UPPER));
        const bool desc = cls_n ? (g.tri & TRI_B_UPPER) != 0 : (g.tri & TRI_A_LOWER) != 0;
        const int ncls = cls_n ? nx : ny, other = cls_n ? ny : nx;
        for (int c0 = 0; c0 < ncls; ++c0) {
            const int c = desc ? ncls - 1 - c0 : c0;
            for (int o = 0; o < other; ++o) {
                const int bn = cls_n ? c : o, bm = cls_n ? o : c;
                int kb = 0, ke = g.K;
                if (g.tri & TRI_A_LOWER) ke = std::min(ke, (bm + 1) * BT);
                if (g.tri & TRI_A_UPPER) kb = std::max(kb, bm * BT);
                if (g.tri & TRI_B_LOWER) kb = std::max(kb, bn * BT);
                if (g.tri & TRI_B_UPPER) ke = std::min(ke, (bn + 1) * BT);
                const int nk = (g.lower_only && bn > bm) ? 0 : std::max(0, ke - kb) / BK;
                for (int z = 0; z < nzg; ++z) place(nk);
            }
        }
    }
    return makespan;
}

inline bool gemm_uses_dma(bool a_t, const GemmArgs& g, int batch);      // (gemm_f64_dma.hpp)

// Which kernel a launch takes (also used for the per-kernel statistics).
inline bool gemm_uses_big(bool a_t, const GemmArgs& g, int batch) {
    if (a_t || g.M % GEMM_BIG || g.N % GEMM_BIG || g.K % GEMM_BIG_BK || (g.tri & ~15) || g.stamps) return false;
    if (g.lower_only && g.M != g.N) return false;
    const int policy = gemm_big_policy();
    if (policy == 1) return true;
    if (policy == 2) return false;
    // The makespan model below was calibrated against the register-staged 64 x 64 kernel; a launch that the LDS-DMA kernel
    // takes is faster there than the model thinks, and the 82 KB of LDS of a 128 x 128 workgroup cost the other streams more
    // than they save (found at batches of 40 / 48 / 56 problems, which the model sent here: 4 x 48 41.3 k solves/s against
    // 44.8 k at 4 x 32 and 47.5 k at 4 x 56)
    if (gemm_uses_dma(a_t, g, batch)) return false;
    if ((int64_t)(g.M / GEMM_BIG) * (g.N / GEMM_BIG) * batch < 64) return false;
    // the choice depends on the shape only: remember it (launchers run on several host threads)
    static std::mutex mu;
    static std::map<std::array<int, 6>, bool> memo;
    const std::array<int, 6> key = {g.M, g.N, g.K, g.tri, g.lower_only, batch};
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(key);
    if (it == memo.end()) it = memo.emplace(key, gemm_estimate_us(true, g, batch) < gemm_estimate_us(false, g, batch)).first;
    return it->second;
}

// The 32 x 32 low-latency kernel: launches with so few 64 x 64 tiles that most CUs would idle.
inline int& gemm_ll_policy() {       // 0: by launch size, 2: never (tests compare the two kernels)
    static int v = 0;
    return v;
}
inline bool gemm_uses_ll(const GemmArgs& g, int batch) {
    if (gemm_ll_policy() == 2 || g.K % 32 || (g.tri & ~15) || g.stamps) return false;
    const int64_t sx = g.N / GEMM_BN, sy = g.M / GEMM_BM;
    const int64_t tiles = ((g.lower_only && g.M == g.N) ? sx * (sx + 1) / 2 : sx * sy) * batch;
    static const int64_t ll_env = [] { const char* e = getenv("GMRF_GEMM_LL_MAX_TILES"); return (int64_t)(e ? atoi(e) : -1); }();   // tuning aid
    if (ll_env >= 0) return tiles <= ll_env;
    // The 32 x 32-tile kernel is for the latency of ONE problem's small products (or a handful of problems).  A launch of a
    // real batch is better served by 64 x 64 tiles even when they fill few CUs: with several handles in flight what counts is
    // what a launch takes from the chip, not how soon it ends (darcy256, 4 x 32: the 128-tile launches on gemm_f64_dma are
    // slower one by one -- 1-stream GEMM time 46.9 -> 48.9 ms -- and the job is faster, 43.7 k -> 44.2 k solves/s).
    // (round 4: a batch of EIGHT counts as a handful -- elliptic512 at 4 x 8: 6.28 k -> 6.46 k solves/s with its 128-tile launches here)
    return tiles <= 128 && batch <= 8;
}

// The LDS-DMA kernels (gemm_f64_dma.hpp) take the launches that qualify; defined there.
inline bool gemm_try_dma(hipStream_t st, bool a_t, bool b_n, const GemmArgs& g, int batch, hipEvent_t ev_start, hipEvent_t ev_stop,
                         hipError_t* err);
inline bool gemm_uses_dma(bool a_t, const GemmArgs& g, int batch);

// Host-side launcher.  tri/lower_only semantics as in GemmArgs.
// ev_start / ev_stop (optional, profiling): updated by the runtime with the dispatch's own begin / end time stamps
// (hipExtLaunchKernelGGL) -- the kernel's duration as a rocprofv3 kernel trace reports it, without the gap an
// event pair recorded around the launch adds.
inline hipError_t launch_gemm(hipStream_t st, bool a_t, bool b_n, const GemmArgs& g, int batch,
                              hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return hipSuccess;
#define GMRF_KLAUNCH(KERNEL, GRID, BLOCK, LDS, ST, ARGS)                                          \
    do {                                                                                           \
        if (ev_start) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, ev_start, ev_stop, 0, ARGS); \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, ARGS);                               \
    } while (0)
    const int64_t sx = g.N / GEMM_BN, sy = g.M / GEMM_BM;
    const bool tri_grid = g.lower_only && g.M == g.N;
    GemmArgs gs = g;
    gs.lower_only = tri_grid ? 1 : (g.lower_only ? 2 : 0);   // 2: rectangular grid, tiles above the diagonal exit
    dim3 grid((unsigned)((tri_grid ? sx * (sx + 1) / 2 : sx * sy) * batch)), block(256);
    if (gemm_uses_big(a_t, g, batch)) {
        const int64_t nx = g.N / GEMM_BIG, ny = g.M / GEMM_BIG;
        dim3 bgrid((unsigned)((g.lower_only ? nx * (nx + 1) / 2 : nx * ny) * batch));
        if (b_n) GMRF_KLAUNCH(gemm_f64_big<true>, bgrid, block, gemm_big_lds_bytes<true>(), st, g);
        else GMRF_KLAUNCH(gemm_f64_big<false>, bgrid, block, gemm_big_lds_bytes<false>(), st, g);
        return hipGetLastError();
    }
    if (!gemm_uses_ll(g, batch)) {
        hipError_t derr = hipSuccess;
        if (gemm_try_dma(st, a_t, b_n, g, batch, ev_start, ev_stop, &derr)) return derr;
    }
    static const bool force_bk16 = getenv("GMRF_GEMM_BK16") != nullptr;     // tuning aid
    // One LDS stage instead of two (two barriers per K step, half the LDS): four workgroups per CU instead of
    // two.  Launches of more than ~1.5 rounds of tiles gain (the fixed part of a tile -- first operand loads,
    // epilogue -- hides behind three neighbours instead of one: 1024^2 x 32 outputs, K = 64 / 128 / 256 / 1024:
    // 26 -> 32, 38 -> 44, 47.6 -> 49.5, 53.7 -> 55.5 TF/s); single-round launches lose 2-8 % to the second barrier.
    static const int single_env = [] { const char* e = getenv("GMRF_GEMM_SINGLE_STAGE"); return e ? atoi(e) : -1; }();   // tuning aid
    const bool single_stage = single_env >= 0 ? single_env != 0 : grid.x > 768;
    const bool wide = (g.K % 32 == 0) && !force_bk16;
    if (gemm_uses_ll(g, batch)) {
        const int64_t lx = g.N / GEMM_LL, ly = g.M / GEMM_LL;
        dim3 lgrid((unsigned)((tri_grid ? lx * (lx + 1) / 2 : lx * ly) * batch));
#define GMRF_GEMM_LL(AT, BN) GMRF_KLAUNCH((gemm_f64_ll<AT, BN>), lgrid, block, 0, st, gs)
        if (!a_t && !b_n) GMRF_GEMM_LL(false, false);
        else if (!a_t && b_n) GMRF_GEMM_LL(false, true);
        else if (a_t && !b_n) GMRF_GEMM_LL(true, false);
        else GMRF_GEMM_LL(true, true);
#undef GMRF_GEMM_LL
        return hipGetLastError();
    }
#define GMRF_GEMM_LAUNCH(AT, BN)                                                                 \
    do {                                                                                         \
        if (single_stage) gs.tri |= 2048;                                                        \
        const size_t ldsdiv = single_stage ? 2 : 1;                                              \
        if (wide) GMRF_KLAUNCH((gemm_f64_mfma<AT, BN, 32>), grid, block, gemm_lds_bytes<32>() / ldsdiv, st, gs); \
        else GMRF_KLAUNCH((gemm_f64_mfma<AT, BN, 16>), grid, block, gemm_lds_bytes<16>() / ldsdiv, st, gs);     \
    } while (0)
    if (!a_t && !b_n) GMRF_GEMM_LAUNCH(false, false);
    else if (!a_t && b_n) GMRF_GEMM_LAUNCH(false, true);
    else if (a_t && !b_n) GMRF_GEMM_LAUNCH(true, false);
    else GMRF_GEMM_LAUNCH(true, true);
#undef GMRF_GEMM_LAUNCH
#undef GMRF_KLAUNCH
    return hipGetLastError();
}

}  // namespace gmrf
