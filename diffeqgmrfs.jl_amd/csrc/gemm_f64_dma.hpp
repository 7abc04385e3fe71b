// fp64 MFMA GEMM whose operand tiles go global -> LDS WITHOUT passing through registers
// (global_load_lds_dwordx4), round 3.  Same contract as gemm_f64_mfma (gemm_f64.hpp):
//
//   C[m][n] = beta * D[m][n] + alpha * sum_k A[m*lda + k] * b(k,n),   b(k,n) = B_N ? B[k*ldb + n] : B[n*ldb + k]
//
// Why: every instruction that RETURNS data into VGPRs beside the fp64 MFMA stream costs matrix-pipe issue time
// (tools/mb5.hip: ds_read_b128 and global_load_dwordx4 -> VGPR alike), and the register-staged kernel pays, per 32
// MFMAs of a wave, 8 global loads + 8 ds_write_b128 + 16 ds_read_b128.  Here the staging costs no VGPR traffic at
// all (the loads write LDS directly, 1 KiB per wave instruction), and the BM x BN workgroup tile is a template
// parameter: 128 x 64 / 64 x 128 (a wave owns 64 x 32 / 32 x 64 = 8 MFMA tiles: 6 fragment reads per 16 MFMAs instead of
// 4 per 8) for launches whose M / N allow it, 64 x 64 otherwise.
//
// LDS images (K step BK = 16, STAGES buffers of (BM + BN) * 128 bytes):
//   [row][16] for an operand stored [row][k]: a row is 128 bytes = 8 chunks of 16 bytes; chunk c of row r sits at
//       chunk position c ^ ((r >> 1) & 7).  A wave's ds_read_b128 of one k pair from 16 consecutive rows then touches
//       every bank once (unswizzled, rows 128 bytes apart put every second row on the same banks: 4-way conflicts).
//       global_load_lds writes LDS linearly (wave-uniform base + lane * 16), so the swizzle is applied on the SOURCE
//       side: lane l of the instruction that fills rows 8q .. 8q+7 fetches chunk (l & 7) ^ ((row >> 1) & 7) of row 8q + (l >> 3).
//   [16][BN] as it lies in memory for B stored [k][n] (a 16-lane group reads 256 contiguous bytes: conflict free).
// Pipeline: STAGES - 1 tiles are in flight (LDS-DMA) while one is multiplied; a wave waits for ITS OWN requests of tile
// kt with a counted s_waitcnt vmcnt(n) that leaves the later tiles in flight, and ONE raw s_barrier per K step both makes
// every wave's part of tile kt visible and frees the buffer of tile kt - 1 for the next request (no __syncthreads: its
// fence would drain the DMA queue).  All LDS is ONE dynamic array (cdna_hip_programming.md: a second __shared__ object makes hipcc
// wait vmcnt(0) before every fragment read).
// Summation order per output element = gemm_f64_mfma's (k ascending in steps of 8, even k then odd k inside a step
// pair ... identical MFMA k-slot assignment), so results are bitwise those of the register-staged kernel.
#pragma once
#include "gemm_f64.hpp"

namespace gmrf {

constexpr int DMA_BK = 16;
constexpr int DMA_ROWS_ASCENDING = 1024;     // GemmArgs::tri bit set by the launcher: triangular grid walked from row tile 0 on
#ifndef GMRF_DMA_READS_FIRST
#define GMRF_DMA_READS_FIRST 0      // 1: a K step issues its fragment reads before the next tile's LDS-DMA requests (measured: see DESIGN.md)
#endif

template <int BM, int BN>
constexpr size_t gemm_dma_lds_bytes(int stages) { return (size_t)stages * (BM + BN) * DMA_BK * sizeof(double); }

#define GMRF_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GMRF_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Tile order over the 1-D grid: XCD-grouped problems (blockIdx % 8 = group), inside a group longest K first.
// lower_only == 1: triangular grid of BM x BN tiles that touch the lower triangle (M == N).
template <int BM, int BN>
__device__ __forceinline__ void gemm_dma_tile_order(const GemmArgs& g, int& bm, int& bn, int& z) {
    const int nx = g.N / BN, ny = g.M / BM;
    int tpp;
    if (g.lower_only == 1) {
        // row tile bm covers rows [bm BM, (bm + 1) BM): column tiles 0 .. ((bm + 1) BM - 1) / BN
        if (BM >= BN) tpp = (BM / BN) * ny * (ny + 1) / 2;
        else tpp = 0;                                           // (not launched: see launch_gemm_dma)
    } else tpp = nx * ny;
    const int nz = (int)gridDim.x / tpp;
    const int groups = (nz % 8 == 0) ? 8 : 1;
    const int xg = (int)blockIdx.x % groups, q = (int)blockIdx.x / groups, nzg = nz / groups;
    int zq;
    if (g.lower_only == 1) {
        constexpr int R = BM / BN > 0 ? BM / BN : 1;            // column tiles per diagonal step
        const int tile = q % tpp;
        zq = q / tpp;
        // tiles of row bm: R (bm + 1); before it: R bm (bm + 1) / 2.  Longest rows first does not matter here (K is
        // the same for all tiles of a rank-k update; G2's staircase bounds grow with bm): rows in descending order.
        // (round 4) with staircase bounds K shrinks as bm grows -- K(bm, bn) = W - kst[bm] for bn <= bm -- so ascending rows ARE
        // longest first, and the descending order used to leave every problem group's longest tile, (0, 0), for the very end
        const int tr = (g.tri & DMA_ROWS_ASCENDING) ? tile : tpp - 1 - tile;
        int b = (int)((sqrtf(8.0f * (float)(tr / R) + 1.0f) - 1.0f) * 0.5f);
        while (R * b * (b + 1) / 2 > tr) --b;
        while (R * (b + 1) * (b + 2) / 2 <= tr) ++b;
        bm = b;
        bn = tr - R * b * (b + 1) / 2;
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

// A_T (round 4): A stored [k][m] -- the product A^T B of two operands that both lie [k][.] in memory (selected inversion:
// M = T1^T C_w, S = X^T Y) without a transposing pass before it.  Its LDS image is the [16][BM] image of a B stored [k][n], its
// fragments pair neighbouring ROW tiles the way that image pairs column tiles: MFMA tile i = 2 ip + q of a wave holds the rows
// wm + 32 ip + 2 li + q (li = MFMA row index), which only moves where the epilogue puts a register.  Same k slots, same
// summation order: bitwise the result of transposing A first.
template <int BM, int BN, bool B_N, int STAGES, bool A_T = false>
__global__ __launch_bounds__(256, (BM + BN > 128) ? 3 : 4) void gemm_f64_dma(GemmArgs g) {
    constexpr int BK = DMA_BK;
    constexpr int MI = BM / 32, NJ = BN / 32;                    // 16 x 16 MFMA tiles of a wave: MI x NJ
    constexpr int NA = BM / 32, NB = BN / 32;                    // LDS-DMA instructions per wave and K step (A, B)
    constexpr int A_ST = BM * BK, B_ST = BN * BK, ST = A_ST + B_ST;   // doubles per stage
    static_assert(NJ % 2 == 0 || !B_N, "the [k][n] image pairs neighbouring column tiles");
    static_assert(MI % 2 == 0 || !A_T, "the [k][m] image pairs neighbouring row tiles");
    int bm, bn, z;
    gemm_dma_tile_order<BM, BN>(g, bm, bn, z);
    const int m0 = bm * BM, n0 = bn * BN;
    if (g.lower_only == 2 && n0 > m0 + BM - 1) return;
    const int zi = z % g.nb1, zp = z / g.nb1;
    const double* __restrict__ A = g.A + (int64_t)zi * g.strideA + (int64_t)zp * g.pA;
    const double* __restrict__ B = g.B + (int64_t)zi * g.strideB + (int64_t)zp * g.pB;
    double* C = g.C + (int64_t)zi * g.strideC + (int64_t)zp * g.pC;
    const double* Dm = g.D ? g.D + (int64_t)zp * g.pD : C;
    const int64_t ldd = g.D ? g.ldd : g.ldc;

    int kb = 0, ke = g.K;
    if (g.tri & TRI_A_LOWER) ke = min(ke, m0 + BM);
    if (g.tri & TRI_A_UPPER) kb = max(kb, m0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, n0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, n0 + BN);
    // staircase bounds are kept per 64-wide tile and are monotone: a wider tile starts at its first part's bound and
    // ends at its last part's (the parts' own zero ranges hold real zeros)
    if (g.kb_m) kb = max(kb, g.kb_m[bm * (BM / 64)]);
    if (g.kb_n) kb = max(kb, g.kb_n[bn * (BN / 64)]);
    if (g.ke_n) ke = min(ke, g.ke_n[bn * (BN / 64) + BN / 64 - 1]);
    const int nkt = (ke > kb) ? (ke - kb) / BK : 0;

    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int t = threadIdx.x;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
    const int li = lane & 15, lq = lane >> 4;
    // waves whose whole quadrant lies strictly above the block diagonal of a lower-only product only help with staging
    // (inside a 64 x 64 tile on the diagonal everything is computed, as gemm_f64_mfma does)
    const bool idle = g.lower_only != 0 && ((n0 + wn) / 64 > (m0 + wm + BM / 2 - 1) / 64);

    // ---- staging plan: per-lane byte offsets (relative to the operand's k0 column / row), one per DMA instruction
    uint32_t offa[NA], offb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = w + 4 * i;
        if (A_T) {
            constexpr int LPR = BM / 2;                          // lanes per k row
            const int k = q * (64 / LPR) + lane / LPR, col = (lane % LPR) * 2;
            offa[i] = (uint32_t)(((int64_t)k * g.lda + m0 + col) * 8);
        } else {
            const int row = 8 * q + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            offa[i] = (uint32_t)(((int64_t)(m0 + row) * g.lda + 2 * ch) * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = w + 4 * i;
        if (B_N) {
            // [k][n]: BN * 8 bytes per k row; an instruction covers 1024 / (BN * 8) rows
            constexpr int LPR = BN / 2;                          // lanes per k row
            const int k = q * (64 / LPR) + lane / LPR, col = (lane % LPR) * 2;
            offb[i] = (uint32_t)(((int64_t)k * g.ldb + n0 + col) * 8);
        } else {
            const int row = 8 * q + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            offb[i] = (uint32_t)(((int64_t)(n0 + row) * g.ldb + 2 * ch) * 8);
        }
    }
    auto issue = [&](int kt, int stage) {
        const int k0 = kb + kt * BK;
        const char* abase = reinterpret_cast<const char*>(A_T ? A + (int64_t)k0 * g.lda : A + k0);
        const char* bbase = reinterpret_cast<const char*>(B_N ? B + (int64_t)k0 * g.ldb : B + k0);
        double* as = gsm + stage * ST;
        double* bs = as + A_ST;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            asm volatile("" : "+v"(offa[i]));                    // keep base + zext(offset) visible to instruction selection
            __builtin_amdgcn_global_load_lds(GMRF_GLB_PTR(abase + offa[i]), GMRF_LDS_PTR(as + (w + 4 * i) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            asm volatile("" : "+v"(offb[i]));
            __builtin_amdgcn_global_load_lds(GMRF_GLB_PTR(bbase + offb[i]), GMRF_LDS_PTR(bs + (w + 4 * i) * 128), 16, 0, 0);
        }
    };

    v4d acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // fragment addresses inside a stage (doubles): A rows wm + 16 i + li, chunk kg * 4 + lq swizzled by the row
    int fa[MI], fb[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm + 16 * i + li;
        fa[i] = A_T ? (2 * lq + (i & 1)) * BM + wm + 32 * (i >> 1) + 2 * li : row * 16 + 2 * (lq ^ ((row >> 1) & 3));
    }
    // (chunk = kg * 4 + lq; swizzle key (row >> 1) & 7 = 4 * s2 + s01: chunk ^ key = (kg ^ s2) * 4 + (lq ^ s01))
    int sa2[MI], sb2[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int row = wm + 16 * i + li; sa2[i] = (!A_T && ((row >> 1) & 4)) ? 8 : 0; }   // doubles: chunk bit 2 = 4 chunks = 8 doubles
    if (!B_N) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wn + 16 * j + li;
            fb[j] = A_ST + row * 16 + 2 * (lq ^ ((row >> 1) & 3));
            sb2[j] = ((row >> 1) & 4) ? 8 : 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { fb[j] = A_ST + (2 * lq + (j & 1)) * BN + wn + 32 * (j >> 1) + 2 * li; sb2[j] = 0; }
        // fb[2 jp + p]: row k = kg * 8 + 2 lq + p, columns (2 li, 2 li + 1) of column group jp
    }

    // Pipeline: DEPTH = STAGES - 1 tiles are in flight while one is multiplied.  Per K step: wait for this wave's own
    // requests of tile kt (counted: the tiles after it stay in flight), barrier (every wave's part of tile kt has landed,
    // and every wave has finished reading tile kt - 1), request tile kt + DEPTH into the buffer tile kt - 1 occupied,
    // multiply tile kt.  ONE barrier per step.
    constexpr int DEPTH = STAGES - 1;
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
        if (s < nkt) issue(s, s);
    for (int kt = 0; kt < nkt; ++kt) {
        const int ahead = min(DEPTH - 1, nkt - 1 - kt);          // requested tiles behind tile kt
        if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!GMRF_DMA_READS_FIRST && kt + DEPTH < nkt) issue(kt + DEPTH, (kt + DEPTH) % STAGES);
        if (!idle) {
            const double* sm = gsm + (kt % STAGES) * ST;
            v2d a[BK / 8][MI], b[BK / 8][NJ];
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kg][i] = *reinterpret_cast<const v2d*>(sm + fa[i] + (A_T ? kg * 8 * BM : ((kg * 8) ^ sa2[i])));
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[kg][j] = *reinterpret_cast<const v2d*>(sm + fb[j] + (B_N ? kg * 8 * BN : ((kg * 8) ^ sb2[j])));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (GMRF_DMA_READS_FIRST && kt + DEPTH < nkt) issue(kt + DEPTH, (kt + DEPTH) % STAGES);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kg = 0; kg < BK / 8; ++kg)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            double av;                             // ([k][m] image: a[2 ip + p] = row k + p, .x / .y = rows 2 li / 2 li + 1 = tiles 2 ip / 2 ip + 1)
                            if (A_T) av = (i & 1) ? a[kg][2 * (i >> 1) + p].y : a[kg][2 * (i >> 1) + p].x;
                            else av = p ? a[kg][i].y : a[kg][i].x;
                            // [n][k] image: tile j = b[j], k slot = .x / .y.  [k][n] image: b[2 jp + p] = row k + p,
                            // .x / .y = even / odd columns = output tiles 2 jp / 2 jp + 1
                            double bv;
                            if (B_N) bv = (j & 1) ? b[kg][2 * (j >> 1) + p].y : b[kg][2 * (j >> 1) + p].x;
                            else bv = p ? b[kg][j].y : b[kg][j].x;
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i][j], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0);
        } else if (GMRF_DMA_READS_FIRST && kt + DEPTH < nkt) issue(kt + DEPTH, (kt + DEPTH) % STAGES);
    }
    if (idle) return;

    // f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.  With an addend (beta != 0) all of its loads are
    // issued before the first is used (one wait instead of one per element).
    const double alpha = g.alpha, beta = g.beta;
    auto orow = [&](int i, int r) -> int64_t {          // row of register r of the wave's MFMA tile i
        return A_T ? m0 + wm + 32 * (i >> 1) + 2 * (lq + 4 * r) + (i & 1) : m0 + wm + i * 16 + lq + 4 * r;
    };
    if (B_N) {
        v2d d[MI][4][NJ / 2 > 0 ? NJ / 2 : 1];
        if (beta != 0.0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jp = 0; jp < NJ / 2; ++jp)
                        d[i][r][jp] = *reinterpret_cast<const v2d*>(Dm + orow(i, r) * ldd + n0 + wn + jp * 32 + 2 * li);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    v2d v = (v2d){alpha * acc[i][2 * jp][r], alpha * acc[i][2 * jp + 1][r]};
                    if (beta != 0.0) { v.x += beta * d[i][r][jp].x; v.y += beta * d[i][r][jp].y; }
                    *reinterpret_cast<v2d*>(C + orow(i, r) * g.ldc + n0 + wn + jp * 32 + 2 * li) = v;
                }
    } else {
        double d[MI][4][NJ];
        if (beta != 0.0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        d[i][r][j] = Dm[orow(i, r) * ldd + n0 + wn + j * 16 + li];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v += beta * d[i][r][j];
                    C[orow(i, r) * g.ldc + n0 + wn + j * 16 + li] = v;
                }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel choice.  GMRF_GEMM_DMA (tuning aid): 0 = never, 1 = only the 64 x 64 tile, 2 (default) = 128 x 64 / 64 x 128 where
// the shape allows and the launch keeps >= 2 workgroups per CU, 64 x 64 otherwise.  GMRF_GEMM_DMA_STAGES: 2 (default) or 3.
inline int& gemm_dma_policy() {
    static int v = [] { const char* e = getenv("GMRF_GEMM_DMA"); return e ? atoi(e) : 2; }();
    return v;
}
inline int& gemm_dma_force() {          // tests: 0 = by policy, 1 / 2 / 3 = this tile shape wherever the sizes divide
    static int v = 0;
    return v;
}
inline int gemm_dma_stages() {
    static const int v = [] { const char* e = getenv("GMRF_GEMM_DMA_STAGES"); const int s = e ? atoi(e) : 2; return s == 3 ? 3 : 2; }();
    return v;
}

// 0: not a DMA launch; 1: 64 x 64; 2: 128 x 64; 3: 64 x 128
inline int gemm_dma_shape(bool a_t, const GemmArgs& g, int batch) {
    const int policy = gemm_dma_force() ? 2 : gemm_dma_policy();
    if (policy == 0 || g.stamps || (g.tri & ~15) || g.K % DMA_BK || g.M % 64 || g.N % 64) return 0;
    if ((g.lda & 1) || (g.ldb & 1) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((g.strideA | g.strideB | g.pA | g.pB) & 1)) return 0;
    // the B [k][n] epilogue loads the addend and stores the result as 16-byte pieces
    if (((uintptr_t)g.C & 15) || (g.ldc & 1) || ((g.strideC | g.pC) & 1)) return 0;
    if (g.D && (((uintptr_t)g.D & 15) || (g.ldd & 1) || (g.pD & 1))) return 0;
    if ((int64_t)g.M * g.ldc * 8 >= ((int64_t)1 << 32) || (g.D && (int64_t)g.M * g.ldd * 8 >= ((int64_t)1 << 32))) return 0;
    // 32-bit byte offsets inside a problem's operand
    if ((int64_t)(a_t ? g.K : g.M) * g.lda * 8 >= ((int64_t)1 << 32) || (int64_t)std::max(g.N, g.K) * g.ldb * 8 >= ((int64_t)1 << 32)) return 0;
    if (a_t) return 1;                                  // A stored [k][m]: the 64 x 64 tile
    const int force = gemm_dma_force();
    if (force == 2 && g.M % 128 == 0) return 2;
    if (force == 3 && g.N % 128 == 0 && !g.lower_only) return 3;
    if (force) return 1;
    if (policy == 1) return 1;
    // The wide tiles pay on full products only (1024^3 x 16: 65 - 69 TF/s against 61 - 64 with 64 x 64 tiles); on the factor's
    // triangular / lower-only / staircase launches their idle quadrants and longer tails lose what the fewer fragment reads gain
    // (tools/gemm_dma_rate.py: G2 304 us against 280 us, rank-256 update of 768^2 101 against 88 us).
    const bool plain = !g.lower_only && !g.tri && !g.kb_m && !g.kb_n && !g.ke_n;
    const int64_t t64 = (int64_t)(g.M / 64) * (g.N / 64) * batch;
    if (policy >= 3 || (plain && t64 >= 2048)) {
        if (g.M % 128 == 0) return 2;
        if (g.N % 128 == 0 && !g.lower_only) return 3;
    }
    return 1;
}
inline bool gemm_uses_dma(bool a_t, const GemmArgs& g, int batch) { return gemm_dma_shape(a_t, g, batch) != 0; }

inline hipError_t gemm_dma_init() {
    hipError_t e = hipSuccess;
#define GMRF_DMA_ATTR(BM, BN, BNAT, ST)                                                                          \
    if (e == hipSuccess)                                                                                          \
        e = hipFuncSetAttribute((const void*)gemm_f64_dma<BM, BN, BNAT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)gemm_dma_lds_bytes<BM, BN>(ST) + 64 * 1024);
    GMRF_DMA_ATTR(64, 64, false, 2) GMRF_DMA_ATTR(64, 64, true, 2) GMRF_DMA_ATTR(64, 64, false, 3) GMRF_DMA_ATTR(64, 64, true, 3)
    GMRF_DMA_ATTR(128, 64, false, 2) GMRF_DMA_ATTR(128, 64, true, 2) GMRF_DMA_ATTR(128, 64, false, 3) GMRF_DMA_ATTR(128, 64, true, 3)
    GMRF_DMA_ATTR(64, 128, false, 2) GMRF_DMA_ATTR(64, 128, true, 2) GMRF_DMA_ATTR(64, 128, false, 3) GMRF_DMA_ATTR(64, 128, true, 3)
#undef GMRF_DMA_ATTR
    return e;
}

// Launches the product on a DMA kernel if it qualifies (returns true), else leaves it to launch_gemm's other kernels.
inline bool gemm_try_dma(hipStream_t st, bool a_t, bool b_n, const GemmArgs& g, int batch, hipEvent_t ev_start, hipEvent_t ev_stop,
                         hipError_t* err) {
    const int shape = gemm_dma_shape(a_t, g, batch);
    if (shape == 0) return false;
    const bool tri_grid = g.lower_only && g.M == g.N;
    GemmArgs gs = g;
    gs.lower_only = tri_grid ? 1 : (g.lower_only ? 2 : 0);
    static const bool asc_off = [] { const char* e = getenv("GMRF_GEMM_G2_ORDER"); return e && atoi(e) == 0; }();   // tuning aid
    if (tri_grid && (g.kb_m || g.kb_n) && !asc_off) gs.tri |= DMA_ROWS_ASCENDING;
    const int stages = gemm_dma_stages();
    const dim3 block(256);
#define GMRF_DMA_GO(BM, BN)                                                                                       \
    do {                                                                                                          \
        const int64_t nx = g.N / BN, ny = g.M / BM;                                                               \
        const int64_t tiles = tri_grid ? (int64_t)(BM / BN > 0 ? BM / BN : 1) * ny * (ny + 1) / 2 : nx * ny;      \
        const dim3 grid((unsigned)(tiles * batch));                                                               \
        static const size_t lds_pad = [] { const char* e = getenv("GMRF_GEMM_DMA_LDS_PAD_KB"); return (size_t)(e ? atoi(e) : 0) * 1024; }();   /* tuning aid */ \
        const size_t lds = gemm_dma_lds_bytes<BM, BN>(stages) + lds_pad;                                          \
        if (a_t && BM == 64 && BN == 64) {                                                                        \
            if (b_n) GMRF_DMA_K((gemm_f64_dma<64, 64, true, 2, true>)); else GMRF_DMA_K((gemm_f64_dma<64, 64, false, 2, true>)); \
        } else if (stages == 3) {                                                                                 \
            if (b_n) GMRF_DMA_K((gemm_f64_dma<BM, BN, true, 3>)); else GMRF_DMA_K((gemm_f64_dma<BM, BN, false, 3>)); \
        } else {                                                                                                  \
            if (b_n) GMRF_DMA_K((gemm_f64_dma<BM, BN, true, 2>)); else GMRF_DMA_K((gemm_f64_dma<BM, BN, false, 2>)); \
        }                                                                                                         \
    } while (0)
#define GMRF_DMA_K(KERNEL)                                                                                        \
    do {                                                                                                          \
        if (ev_start) hipExtLaunchKernelGGL(KERNEL, grid, block, lds, st, ev_start, ev_stop, 0, gs);              \
        else hipLaunchKernelGGL(KERNEL, grid, block, lds, st, gs);                                                \
    } while (0)
    if (shape == 2) GMRF_DMA_GO(128, 64);
    else if (shape == 3) GMRF_DMA_GO(64, 128);
    else GMRF_DMA_GO(64, 64);
#undef GMRF_DMA_K
#undef GMRF_DMA_GO
    *err = hipGetLastError();
    return true;
}

}  // namespace gmrf
