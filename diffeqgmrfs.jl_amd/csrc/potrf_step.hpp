// One 64-column panel step of the dense block Cholesky (dpotrf of one bs x bs Schur block,
// /root/reference/src/tridiagonal_cholesky.jl:67,77) in ONE launch:
//
//   every workgroup   : S_jj -> L_jj, X_jj = L_jj^-1      (tile_potrf_inv, redundantly per WG)
//   workgroup (r, c)  : Lr = S[r,j] X_jj^T, Lc = S[c,j] X_jj^T, S[r,c] -= Lr Lc^T   (j < c <= r)
//   workgroup (r,j+1) : additionally stores Lr as L[r,j];  workgroup 0 stores L_jj and X_jj.
//
// The redundant tile factorisation costs nothing in wall time (the other CUs would idle) and
// removes two dependent launches per panel.
//
// tile_potrf_inv: 64x64 tile in LDS, four 16-column panels, each owned by one wave (see below, round 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"

#ifndef GMRF_TILE_EXACT_DIV
#define GMRF_TILE_EXACT_DIV 0
#endif

namespace gmrf {

#ifndef GMRF_PANEL_FLAT_LDS
#define GMRF_PANEL_FLAT_LDS 0
#endif
#ifndef GMRF_TLD
#define GMRF_TLD 66
#endif
constexpr int TLD = GMRF_TLD;           // LDS row stride (doubles) of a 64x64 tile: 16-byte aligned rows; 66 doubles = 132 banks
                                        // = 4 mod 64: the 16 rows a ds_read_b128 serves together (one row per lane, or MFMA
                                        // operand pairs of 16 rows) start 4 banks apart and cover the 64 banks exactly once
                                        // (68 = 8 mod 64 put rows r and r + 8 on the same banks: SQ_LDS_BANK_CONFLICT was
                                        // 0.34 of the tile kernel's LDS-active cycles)
constexpr int TILE_ELEMS = 64 * TLD;
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to fp64 accuracy.  The hardware seed has 24 good bits (tools/acc.hip); ONE third-order
// (Halley) step  y (1 + e/2 + 3 e^2/8),  e = 1 - p y^2,  leaves a truncation error of 5 e^3 / 16
// ~ 1e-23 and is two dependent fp64 operations shorter than two Newton steps -- this routine sits
// on the per-column dependent chain of the tile factorisation.
__device__ __forceinline__ double rsqrt_nr(double p) {
    const double y = __builtin_amdgcn_rsq(p);
    const double e = fma(-p * y, y, 1.0);
    const double c = fma(0.375, e, 0.5);
    return fma(y * e, c, y);
}

// acc += sign * A(16x16) * B(16x16)^T  with A at a[i*lda + k], B at b[j*ldb + k]   ("NT")
__device__ __forceinline__ v4d mm16_nt(const double* a, int lda, const double* b, int ldb, v4d acc,
                                       bool negate, int li, int lq) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        double av = a[li * lda + 4 * ks + lq];
        const double bv = b[li * ldb + 4 * ks + lq];
        if (negate) av = -av;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// acc += sign * A(16x16) * B(16x16)  with A at a[i*lda + k], B at b[k*ldb + j]   ("NN")
__device__ __forceinline__ v4d mm16_nn(const double* a, int lda, const double* b, int ldb, v4d acc,
                                       bool negate, int li, int lq) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        double av = a[li * lda + 4 * ks + lq];
        const double bv = b[(4 * ks + lq) * ldb + li];
        if (negate) av = -av;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// MFMA f64 C/D layout: element reg of lane (li, lq) is (row lq + 4 reg, col li).
__device__ __forceinline__ v4d load_d16(const double* p, int ld, int li, int lq) {
    v4d v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = p[(lq + 4 * r) * ld + li];
    return v;
}
__device__ __forceinline__ void store_d16(double* p, int ld, const v4d& v, int li, int lq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(lq + 4 * r) * ld + li] = v[r];
}

// acc[Jb] += A(16 x 64 strip) B_Jb(16 x 64)^T for the column blocks Jb < JB_END: `arow` = this lane's row of the strip in an LDS tile
// (+ (16 * wave + li) * TLD), Bs the LDS tile whose rows 16 Jb + li are the other operand; k in the slot order used everywhere
// (two consecutive k per ds_read_b128).  JB_END is a COMPILE-TIME bound: with the run-time test `Jb <= wave` inside the
// unrolled loops hipcc closed every MFMA pair with a branch join -- accumulators copied AGPR -> VGPR behind `s_nop 17` -- and
// the 64 MFMAs of wave 3's diagonal update took 10 500 cycles instead of 4 100 (round 4, tools/persist_stamps.py).
template <int JB_END>
__device__ __forceinline__ void strip_nt(const double* arow, const double* Bs, v4d (&acc)[4], int li, int lq) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(arow + k);
#pragma unroll
        for (int Jb = 0; Jb < JB_END; ++Jb) {
            const v2d bv = *reinterpret_cast<const v2d*>(Bs + (16 * Jb + li) * TLD + k);
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc[Jb], 0, 0, 0);
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc[Jb], 0, 0, 0);
        }
    }
}
// the same with the bound chosen by the (wave-uniform) strip index: column blocks 0 .. w of a diagonal tile
__device__ __forceinline__ void strip_nt_diag(int w, const double* arow, const double* Bs, v4d (&acc)[4], int li, int lq) {
    switch (w) {
        case 0: strip_nt<1>(arow, Bs, acc, li, lq); break;
        case 1: strip_nt<2>(arow, Bs, acc, li, lq); break;
        case 2: strip_nt<3>(arow, Bs, acc, li, lq); break;
        default: strip_nt<4>(arow, Bs, acc, li, lq); break;
    }
}
// acc[Jb] += X_strip(rows 16 w .. of a lower-triangular LDS tile) T(64 x 64, stored [k][n])  for the four column blocks: only
// the k groups 0 .. 2 w + 1 of the strip are non-zero (compile-time bound, as above)
template <int W>
__device__ __forceinline__ void strip_tri_nn(const double* xrow, const double* Tsm, v4d (&acc)[4], int li, int lq) {
#pragma unroll
    for (int kg = 0; kg < 2 * W + 2; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(xrow + k);
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) {
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, Tsm[k * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
            acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, Tsm[(k + 1) * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void strip_tri_nn_w(int w, const double* xrow, const double* Tsm, v4d (&acc)[4], int li, int lq) {
    switch (w) {
        case 0: strip_tri_nn<0>(xrow, Tsm, acc, li, lq); break;
        case 1: strip_tri_nn<1>(xrow, Tsm, acc, li, lq); break;
        case 2: strip_tri_nn<2>(xrow, Tsm, acc, li, lq); break;
        default: strip_tri_nn<3>(xrow, Tsm, acc, li, lq); break;
    }
}

// potrf_diag128: pacc[Jb] += L10_strip L10_Jb^T (Jb <= W) and wv[Jb] += L10_strip X00[:, Jb] (X00[k][c] = 0 for k < c: the k
// groups 2 Jb .. 7), one pass over the strip; W = the strip index, compile-time (see strip_nt)
template <int W>
__device__ __forceinline__ void strip_syrk_and_w(const double* arow, const double* Ls, const double* Xs, v4d (&pacc)[4],
                                                 v4d (&wv)[4], int li, int lq) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(arow + k);
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) {
            if (Jb <= W) {
                const v2d bv = *reinterpret_cast<const v2d*>(Ls + (16 * Jb + li) * TLD + k);
                pacc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, pacc[Jb], 0, 0, 0);
                pacc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, pacc[Jb], 0, 0, 0);
            }
            if (kg >= 2 * Jb) {
                wv[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, Xs[k * TLD + 16 * Jb + li], wv[Jb], 0, 0, 0);
                wv[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, Xs[(k + 1) * TLD + 16 * Jb + li], wv[Jb], 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ void strip_syrk_and_w_w(int w, const double* arow, const double* Ls, const double* Xs, v4d (&pacc)[4],
                                                   v4d (&wv)[4], int li, int lq) {
    switch (w) {
        case 0: strip_syrk_and_w<0>(arow, Ls, Xs, pacc, wv, li, lq); break;
        case 1: strip_syrk_and_w<1>(arow, Ls, Xs, pacc, wv, li, lq); break;
        case 2: strip_syrk_and_w<2>(arow, Ls, Xs, pacc, wv, li, lq); break;
        default: strip_syrk_and_w<3>(arow, Ls, Xs, pacc, wv, li, lq); break;
    }
}

// 1/p to fp64 accuracy: hardware seed (24 bits) + two Newton steps (measured 1 ulp, tools/acc.hip).
__device__ __forceinline__ double rcp_nr(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = fma(-p, y, 1.0);
    y = fma(y, e, y);
    e = fma(-p, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the 64 x 64 tile Cholesky as a STREAMING factorisation -- every wave owns one 16-column panel for the whole tile:
//
//   wave k   waits with its panel as MFMA accumulators: while panel p < k is being factored it adds, four columns at a time as
//            they appear in the panel scratch (pcol_p[jj][row], then the reciprocal pivot rv[column] != 0: a wave's LDS stores land
//            in order), the products L[r][j] L[c][j] of those columns to a sum that starts at ZERO; when panel k - 1 is done it
//            subtracts the sum ONCE from its columns of the tile (one rounding at the magnitude of S for all 16 k earlier columns
//            -- what LAPACK's dot-product forms do, and what round 2 found worth two digits on ill-conditioned Schur complements),
//            turns them into the one-row-per-lane form through its own columns of Ts and LEADS panel k from registers (the column
//            chain of round 2: pivot by v_readlane, rsqrt + one Halley step, in-panel rank-1 updates dealt between the chain's
//            instructions), stores its columns of L (zeros above the diagonal included) and goes on to inverse work.
//
// No barrier and no all-wave sub-tile update between two panels: the hand-over from panel k to panel k + 1 is the follower's
// last rank-4 update and its turn through LDS, ~600 cycles (round 4: per panel 240 cycles of loads + 360 .. 620 of stores + two
// barriers + 850 of sub-tile updates, and a tail of 2 700 -- a third of the tile's 22 k cycles, on the chain of every potrf
// kernel).  The 16 x 16 diagonal inverses follow the panels too (inv16_follow, round 4's form for the last block) where a wave is
// free to watch, the block rows of the 64 x 64 inverse are summed as their inputs arrive; waves meet through words in LDS, and at
// ONE barrier at the end.  Same arithmetic per entry as round 4 up to the grouping of the sums (one subtraction per owner wave
// instead of one per panel): not bitwise the round-4 tile, same error bound, and every route shares this one routine.
//
// Measured on the way (tools/tile_timing.py, cycles per tile; round 4: 22.3 k): followers that apply every column as a rank-1
// update with scalar fma -- four waves x 16-column panels 20.7 k, eight waves x 8-column panels 22.4 k, with the multiplicands
// by ds_read_b128 or by v_readlane alike: a lone wave issues about one instruction per 8 cycles, so a follower's column (16 fma +
// its operand reads) takes ~300 cycles against the leader's ~200 and every hand-over waits 700 - 1 700 cycles for the next leader
// to catch up; and a chain of such fma starting from S rounds at |S| once per column (darcy64's forward error 2.1e-11 against the
// oracle's 7.7e-12).  The rank-4 MFMA follower needs ~10 instructions per four columns: 18.3 k, then the rest below.
// ------------------------------------------------------------------------------------------------------------------

// scratch (doubles): panel p keeps its 16 columns for the rows >= 16 p, column-major, pcol_p[jj * W_p + (row - 16 p)], W_p = 64 - 16 p
__host__ __device__ constexpr int pcol_off(int p) { return p == 0 ? 0 : (p == 1 ? 1024 : (p == 2 ? 1792 : 2304)); }
constexpr int RV_OFF = 2560;            // 64 reciprocal pivots, by column of the tile; 0.0 = column not there yet
constexpr int TFLAG_OFF = 2624;         // 32 ints: words the waves of a tile pass to each other (below)
constexpr int WK_ELEMS = 2640;
typedef __attribute__((address_space(3))) volatile int tile_word;
enum { TF_CS = 0 /* +p: the columns of panel p are in Ts */, TF_I = 4 /* +k: X_kk is in Xs */, TF_X10 = 8, TF_X21 = 9, TF_X20 = 10,
       TF_BAD = 12 /* +wave: a non-positive pivot, or a wait that gave up */, TF_SIDE = 16 /* .. 23: the Side's own words */, TF_WORDS = 24 };
constexpr int TILE_SPIN_LIMIT = 1 << 16;    // polls (>= 64 cycles each) of one wave over one tile: past it the waits fall through, the tile is wrong and flagged `bad`

__device__ __forceinline__ void tile_wait(tile_word* w, int& spins) {
#pragma clang loop unroll(disable)
    while (*w == 0 && spins < TILE_SPIN_LIMIT) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void tile_set(tile_word* w, int lane) {
    asm volatile("" ::: "memory");          // (after this wave's LDS stores in program order; the LDS serves a wave's operations in order)
    if (lane == 0) *w = 1;
}

// The wave that owns panel KB factors it: columns [16 KB, 16 KB + 16), rows >= 16 KB, one row per lane, from registers.
//
// A single wave issues one fp64 VALU instruction every ~8 cycles whatever the dependencies
// (measured, tools/mb2.hip), so this routine is written for instruction count:
//   * square-root-free (LDL^T) elimination, columns scaled by rsqrt(p_j) at the end, one rsqrt
//     per lane instead of one per step;
//   * no row masks inside the panel: a finished row (r <= j) keeps "updating" only entries above its diagonal,
//     which nobody reads (they are stored as zeros);
//   * the pivot and the one multiplicand the next column needs travel lane -> SGPR by
//     v_readlane; the other multiplicands of a step are re-read from the panel scratch
//     as wave-uniform operands (one ds_read_b128 per two rank-1 updates).
template <int KB>
__device__ __forceinline__ void panel_lead(double (&a)[16], double* Ts, double* rinvs, double* Wk, int lane, bool& bad,
                                           unsigned long long* dbg = nullptr) {
    constexpr int PW = 16, c0 = PW * KB, W = 64 - c0;
    typedef __attribute__((address_space(3))) double lds_double;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int r = lane;
    if (r >= c0) {                                         // (lanes above the panel hold nothing of it)
        // Make the LDS bases opaque to the compiler: with a known constant address it materialises
        // every broadcast read address with an s_add + v_mov pair; with a VGPR base the constant part
        // goes into the instruction's offset field.  (LDS pointers, address space 3: left generic, every access became FLAT.)
        lds_double* lrow = (lds_double*)(Wk + pcol_off(KB)) + (r - c0);
        lds_double* lun = (lds_double*)(Wk + pcol_off(KB));
        asm volatile("" : "+v"(lrow));
        asm volatile("" : "+v"(lun));
        // (the reciprocal pivots of this panel, wave-uniform values, addressed off the same opaque base: as a constant address each
        //  store cost an s_add + v_mov on the chain -- 3 236 cycles per panel against 2 960, measured)
        lds_double* rv = lun + (RV_OFF + c0 - pcol_off(KB));
        if (dbg && lane == c0) dbg[0] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(3);          // (the column chain is what everybody waits for; a wave of the same SIMD must not delay its issue)
        double up[PW];                          // multiplicands of the previous step (LDS broadcast); first used in step 1
        double lprev = 0.0;
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
            const int j = c0 + jj;
            // A wave issues in order: work that does not depend on the pivot chain only overlaps the
            // chain's latency (~16-20 cycles per dependent fp64 op, tools/mb4.hip) if it sits BETWEEN the
            // chain's instructions in program order.  The rank-1 updates of the PREVIOUS step (their
            // multiplicands were requested from LDS one step ago) are therefore dealt into four slots
            // between the chain operations; sched_barrier keeps the compiler from regrouping them.
#define GMRF_DELAYED_SLOT(S)                                                                        \
            if (jj > 0) { _Pragma("unroll") for (int cc = jj + 1 + (S); cc < PW; cc += 4) a[cc] = fma(-lprev, up[cc], a[cc]); } \
            __builtin_amdgcn_sched_barrier(0);
            const double p = bcast_lane(a[jj], j);
            if (!(p > 0.0)) bad = true;
            const double y = __builtin_amdgcn_rsq(p);
            __builtin_amdgcn_sched_barrier(0);
            GMRF_DELAYED_SLOT(1)
            const double e = fma(-p * y, y, 1.0);
            __builtin_amdgcn_sched_barrier(0);
            GMRF_DELAYED_SLOT(2)
            const double cf = fma(0.375, e, 0.5);
            const double rinv = fma(y * e, cf, y);
            __builtin_amdgcn_sched_barrier(0);
            GMRF_DELAYED_SLOT(3)
            const double l = a[jj] * rinv;       // lane j: p * rinv = sqrt(p)
            a[jj] = l;
            lrow[jj * W] = l;
            asm volatile("" ::: "memory");       // (the column BEFORE its reciprocal pivot: a wave's LDS stores land in order, and
            rv[jj] = rinv;                       //  the followers take a non-zero rv as "the column is there"); same value from every lane: no branch
            __builtin_amdgcn_sched_barrier(0);
            GMRF_DELAYED_SLOT(0)                 // includes column jj + 1, needed by the update below
            if (jj < PW - 1) {
                const double lc1 = bcast_lane(l, j + 1);
                a[jj + 1] = fma(-l, lc1, a[jj + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // aligned 16-byte pieces only (the scratch is 16-byte aligned, W even): left to itself the
            // compiler merges the reads from cc = jj + 2 on into ds_read_b128 at 8-byte boundaries for odd jj
            // (SQ_LDS_UNALIGNED_STALL was 0.42 of the tile kernel's LDS-active cycles, round 2)
#pragma unroll
            for (int cc = (jj + 2) & ~1; cc < PW; cc += 2) {
                const v2d q = *reinterpret_cast<const lds_v2d*>(lun + jj * W + cc);
                if (cc >= jj + 2) up[cc] = q.x;
                up[cc + 1] = q.y;
            }
            lprev = l;
            __builtin_amdgcn_sched_barrier(0);
#undef GMRF_DELAYED_SLOT
        }
        __builtin_amdgcn_s_setprio(0);
        if (dbg && lane == c0) dbg[1] = __builtin_amdgcn_s_memtime();
    }
    // the panel's columns of L -> Ts, all 64 rows: zeros above the diagonal (the strict upper triangle is part of the output)
#pragma unroll
    for (int c = 0; c < PW; c += 2) {
        v2d v;
        v.x = (r >= c0 + c) ? a[c] : 0.0;
        v.y = (r >= c0 + c + 1) ? a[c + 1] : 0.0;
        *reinterpret_cast<v2d*>(Ts + r * TLD + c0 + c) = v;
    }
    if (lane < PW) rinvs[c0 + lane] = Wk[RV_OFF + c0 + lane];
    tile_word* fl = (tile_word*)reinterpret_cast<int*>(Wk + TFLAG_OFF);
    tile_set(fl + TF_CS + KB, lane);
    if (dbg && lane == c0) dbg[2] = __builtin_amdgcn_s_memtime();
}

// A waiting wave keeps what the earlier panels owe its panel [CW, CW + 16) as MFMA accumulators, one 16 x 16 block per row block
// I >= CW / 16 (rows above the panel's diagonal block are never read): acc[I] += L[16 I + i][j] L[CW + c][j] over the columns j of
// the leading panel, four at a time.  Per four columns: one look at the pivot word of the fourth, one operand read per block and
// one for the panel's own rows, one MFMA per block.  `idle` is called once per group of four columns, after its MFMAs are issued:
// the Side's chance to do a bounded piece of work (the chain workgroup of potrf_persist stores and prefetches tiles there).
template <int KB, int CW, bool NEXT, class Idle>
__device__ __forceinline__ void panel_follow_mfma(v4d (&acc)[4], double* Wk, int li, int lq, int& spins, Idle&& idle) {
    constexpr int c0 = 16 * KB, W = 64 - c0, IW = CW / 16;
    typedef __attribute__((address_space(3))) const volatile double lds_cvd;
    lds_cvd* rv = (lds_cvd*)(Wk + RV_OFF + c0);
    lds_cvd* pc = (lds_cvd*)(Wk + pcol_off(KB)) + lq * W + li;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double rk = rv[4 * g + 3];
#pragma clang loop unroll(disable)
        while (rk == 0.0 && spins < TILE_SPIN_LIMIT) {
            if (!NEXT) __builtin_amdgcn_s_sleep(1);     // (the wave that leads the next panel does not sleep: its last wait here is the hand-over of the chain)
            ++spins;
            rk = rv[4 * g + 3];
        }
        const double b = pc[4 * g * W + (CW - c0)];
#pragma unroll
        for (int I = IW; I < 4; ++I) {
            const double a = pc[4 * g * W + (16 * I - c0)];
            acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[I], 0, 0, 0);
        }
        if (!(NEXT && g == 3)) idle(4 * KB + g);
    }
}
// wave `CW / 16` follows the panels 0 .. CW / 16 - 1 one after the other
template <int CW, int KB = 0, class Idle>
__device__ __forceinline__ void panels_follow_mfma(v4d (&acc)[4], double* Wk, int li, int lq, int& spins, Idle&& idle) {
    if constexpr (16 * KB < CW) {
        panel_follow_mfma<KB, CW, CW == 16 * KB + 16>(acc, Wk, li, lq, spins, idle);
        panels_follow_mfma<CW, KB + 1>(acc, Wk, li, lq, spins, idle);
    }
}
// tile - sum (ONE subtraction) -> this wave's columns of Ts -> one row per lane
template <int CW>
__device__ __forceinline__ void panel_to_rows(const v4d (&s0)[4], const v4d (&acc)[4], double (&a)[16], double* Ts, int lane, int li, int lq) {
#pragma unroll
    for (int I = CW / 16; I < 4; ++I) store_d16(Ts + (16 * I) * TLD + CW, TLD, s0[I] - acc[I], li, lq);
    asm volatile("" ::: "memory");          // (the same wave reads them back: its LDS operations are served in order)
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const v2d v = *reinterpret_cast<const v2d*>(Ts + lane * TLD + CW + c);
        a[c] = v.x; a[c + 1] = v.y;
    }
}

// value of quad lane J (lanes 4g .. 4g+3 form a quad) in every lane of the quad: two DPP moves, no LDS
template <int J>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);          // quad_perm:[J,J,J,J]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// One wave: X_kk = L_kk^-1 for the 16x16 diagonal block at c0.  Column-oriented substitution -- once x[k] is final
// every later row gets its update at once -- over all 64 lanes: lane 4 c + q owns rows q, q+4, q+8, q+12 of column c,
// so a step is one multiply in the owner lane, a quad broadcast of x[k] (DPP) and at most four fma, with every
// entry of L and every 1/l_kk in registers before the chain starts: ~60 cycles per step.  (16 lanes with one
// column each, L re-read from LDS inside the chain: 2 900 cycles per call, measured.)
__device__ __forceinline__ void inv16(const double* Ts, const double* rinvs, double* Xs, int c0, int lane) {
    const int c = lane >> 2, q = lane & 3;
    double lv[4][16];                      // lv[i][k] = L[4 i + q][k] for 4 i + q > k, else 0
    double rv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        rv[k] = rinvs[c0 + k];
#pragma unroll
        for (int i = k / 4; i < 4; ++i) {
            const int r = 4 * i + q;
            const double v = Ts[(c0 + r) * TLD + c0 + k];
            lv[i][k] = (r > k) ? v : 0.0;
        }
    }
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (4 * i + q == c) ? 1.0 : 0.0;
#define GMRF_INV16_STEP(K)                                                                        \
    {                                                                                              \
        constexpr int ik = (K) / 4, qk = (K) % 4;                                                  \
        const double xs = x[ik] * rv[K];                                                           \
        if (q == qk) x[ik] = xs;                      /* rows above the column's diagonal stay 0 */ \
        const double xk = quad_bcast<qk>(x[ik]);                                                   \
        _Pragma("unroll") for (int i = ik; i < 4; ++i) x[i] = fma(-lv[i][K], xk, x[i]);           \
    }
    GMRF_INV16_STEP(0) GMRF_INV16_STEP(1) GMRF_INV16_STEP(2) GMRF_INV16_STEP(3)
    GMRF_INV16_STEP(4) GMRF_INV16_STEP(5) GMRF_INV16_STEP(6) GMRF_INV16_STEP(7)
    GMRF_INV16_STEP(8) GMRF_INV16_STEP(9) GMRF_INV16_STEP(10) GMRF_INV16_STEP(11)
    GMRF_INV16_STEP(12) GMRF_INV16_STEP(13) GMRF_INV16_STEP(14) GMRF_INV16_STEP(15)
#undef GMRF_INV16_STEP
#pragma unroll
    for (int i = 0; i < 4; ++i) Xs[(c0 + 4 * i + q) * TLD + c0 + c] = x[i];
}

// Y = L_kk^-1 R for NR right-hand sides R (16 x 16 each) of the diagonal block KB (columns 16 KB ..), run by another wave WHILE
// those columns are being factored: step k needs column k of the block and its reciprocal pivot, which panel_lead leaves in the
// panel scratch as it goes (pcol[k][row], then rv[column]); this wave looks at the pivot word -- and, behind it in the LDS queue,
// at the column -- until the word is non-zero, and does the step of the column-oriented substitution of inv16 (lane 4 c + q owns
// rows q, q + 4, q + 8, q + 12 of column c of every right-hand side).  R = I gives X_kk with the operations of inv16 on the same
// values, bitwise.  (NR > 1, or R = -S_3J to get the last block row of the tile inverse without waiting for X_33: built and
// measured -- a step with two right-hand sides is ~40 instructions, ~320 cycles for a lone wave against the leader's ~190 per
// column, and the tile got 800 cycles LONGER; the last block row stays a product with X_33, one per wave.)
template <int KB, int NR>
__device__ __forceinline__ void solve16_follow(double* Wk, double (&x)[NR][4], int lane, int& spins) {
    constexpr int c0 = 16 * KB;
    typedef __attribute__((address_space(3))) const volatile double lds_cvd;
    lds_cvd* wk = (lds_cvd*)Wk;
    lds_cvd* rvp = (lds_cvd*)(Wk + RV_OFF + c0);
    const int q = lane & 3;
#define GMRF_SOLVE16F_STEP(K)                                                                     \
    {                                                                                              \
        constexpr int ik = (K) / 4, qk = (K) % 4;                                                  \
        constexpr int base = pcol_off(KB) + (K) * (64 - c0);                                       \
        double rk = rvp[K];                                                                        \
        double lvk[4];                                                                             \
        _Pragma("unroll") for (int i = ik; i < 4; ++i) lvk[i] = wk[base + 4 * i + q];             \
        _Pragma("clang loop unroll(disable)")                                                      \
        while (rk == 0.0 && spins < TILE_SPIN_LIMIT) {                                             \
            ++spins;                                                                               \
            rk = rvp[K];                                                                           \
            _Pragma("unroll") for (int i = ik; i < 4; ++i) lvk[i] = wk[base + 4 * i + q];         \
        }                                                                                          \
        lvk[ik] = (4 * ik + q > (K)) ? lvk[ik] : 0.0;   /* (the rows of the later groups are all below row K) */ \
        _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                           \
            const double xs = x[r][ik] * rk;                                                       \
            if (q == qk) x[r][ik] = xs;               /* rows above the step's diagonal keep their values */ \
            const double xk = quad_bcast<qk>(x[r][ik]);                                            \
            _Pragma("unroll") for (int i = ik; i < 4; ++i) x[r][i] = fma(-lvk[i], xk, x[r][i]);   \
        }                                                                                          \
    }
    GMRF_SOLVE16F_STEP(0) GMRF_SOLVE16F_STEP(1) GMRF_SOLVE16F_STEP(2) GMRF_SOLVE16F_STEP(3)
    GMRF_SOLVE16F_STEP(4) GMRF_SOLVE16F_STEP(5) GMRF_SOLVE16F_STEP(6) GMRF_SOLVE16F_STEP(7)
    GMRF_SOLVE16F_STEP(8) GMRF_SOLVE16F_STEP(9) GMRF_SOLVE16F_STEP(10) GMRF_SOLVE16F_STEP(11)
    GMRF_SOLVE16F_STEP(12) GMRF_SOLVE16F_STEP(13) GMRF_SOLVE16F_STEP(14) GMRF_SOLVE16F_STEP(15)
#undef GMRF_SOLVE16F_STEP
}
// right-hand side / result of solve16_follow <-> a 16 x 16 block in LDS (row stride ld): lane 4 c + q, register i <-> (4 i + q, c)
__device__ __forceinline__ void solve16_rhs_identity(double (&x)[4], int lane) {
    const int c = lane >> 2, q = lane & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (4 * i + q == c) ? 1.0 : 0.0;
}
__device__ __forceinline__ void solve16_rhs_neg(double (&x)[4], const double* blk, int ld, int lane) {
    const int c = lane >> 2, q = lane & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = -blk[(4 * i + q) * ld + c];
}
__device__ __forceinline__ void solve16_store(const double (&x)[4], double* blk, int ld, int lane) {
    const int c = lane >> 2, q = lane & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) blk[(4 * i + q) * ld + c] = x[i];
}

// What the waiting waves of a tile factorisation can do for their caller (tile_potrf_inv<Side>):
//   idle(tick, wave, lane)   waves 1 - 3, once per four columns of a panel they follow (tick = 4 panel + group, 0 .. 11; wave w
//                            is called for the ticks < 4 w except its last): a bounded piece of work, never a wait.
//   extra(w, lane, fl)       waves 4 + w of a workgroup with more than four waves, between the routine's two barriers: anything that
//                            ends by itself (fl: the routine's hand-over words, e.g. fl[TF_I + 3] != 0: the tile is done).
//   load_panel(w, s0, ...)   waves 1 - 3, behind the routine's first barrier: the wave's 16-column panel of the tile, one 16 x 16 block
//                            per row block I >= w in the MFMA C/D layout.  Default: it is in Ts.  (The chain workgroup of
//                            potrf_persist has only the first panel of the updated tile in Ts when the routine starts, and forms
//                            the others here, beside wave 0's first panel.)
struct NoSide {
    unsigned long long* stamps = nullptr;     // diagnostic: s_memtime at phase boundaries (tests only)
    __device__ __forceinline__ void idle(int, int, int) {}
    __device__ __forceinline__ void extra(int, int, tile_word*) {}
    __device__ __forceinline__ void load_panel(int w, v4d (&s0)[4], const double* Ts, tile_word*, int li, int lq, int&) {
#pragma unroll
        for (int I = 0; I < 4; ++I) s0[I] = load_d16(Ts + (16 * I) * TLD + 16 * w, TLD, li, lq);
    }
};

#define TILE_STAMP(i) do { if (side.stamps) side.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)

// Ts: SPD tile (lower triangle valid) -> L (strict upper zero).  Xs -> L^-1 (strict upper zero).
// Wk: WK_ELEMS doubles of scratch.  All threads of the workgroup must call this (waves 0 - 3 factor; more waves run Side::extra);
// Ts must be in LDS for everybody (a barrier behind its last store) and nobody may still read Xs, Wk or rinvs.
// the words of a call of tile_potrf_inv: reciprocal pivots (0 = not there) and hand-over flags (the Side's too: its hooks run
// behind the barrier that follows).  Inside the routine, or -- PREPARED -- by the caller before ITS last barrier ahead of the call.
__device__ __forceinline__ void tile_potrf_prepare(double* Wk, int tid) {
    tile_word* fl = (tile_word*)reinterpret_cast<int*>(Wk + TFLAG_OFF);
    if (tid < 64) Wk[RV_OFF + tid] = 0.0;
    else if (tid < 64 + TF_WORDS) fl[tid - 64] = 0;
}
// PREPARED: the caller has run tile_potrf_prepare and then a barrier behind which Ts is complete too (the chain workgroup of
// potrf_persist: one barrier of eight waves per tile less).
template <bool PREPARED = false, class Side>
__device__ __forceinline__ void tile_potrf_inv(double* Ts, double* Xs, double* Wk, double* rinvs, int tid, bool& bad, Side& side) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    tile_word* fl = (tile_word*)reinterpret_cast<int*>(Wk + TFLAG_OFF);
    if (tid == 0) TILE_STAMP(0);
    if (!PREPARED) tile_potrf_prepare(Wk, tid);
    double a[16];
    v4d s0[4], acc[4];                                      // waves 1 - 3: the panel as it came, and what the earlier panels owe it
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const v2d v = *reinterpret_cast<const v2d*>(Ts + lane * TLD + c);
            a[c] = v.x; a[c + 1] = v.y;
        }
    }
    if (!PREPARED) __syncthreads();
    int spins = 0;
    if (wave >= 1 && wave < 4) {
        side.load_panel(wave, s0, Ts, fl, li, lq, spins);
#pragma unroll
        for (int I = 0; I < 4; ++I) acc[I] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    // --- inverse of the tile by block forward substitution over the four 16-row blocks:
    //        X[I][J] = -X[I][I] * sum_{K=J..I-1} L[I][K] X[K][J],      I = 1, 2, 3,  J < I.
    // Only the 16x16 diagonal inverses (computed by substitution) multiply, so L X = I holds to
    // eps * cond(16x16 block); the cheaper recursive doubling X21 = -X22 (L21 X11) multiplies two
    // computed inverses and was measured 20-100x less accurate for cond(tile) >= 1e5.
    // A sum is formed in the block of Xs it belongs to and replaced there by its product with the diagonal inverse (one wave: its
    // LDS operations are served in order, and the product's operands are in registers before its result is stored).
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    auto sum_ij = [&](int I, int J) {                    // Xs[I][J] <- sum_{K=J..I-1} L[I][K] X[K][J]
        v4d t = zero;
        for (int K = J; K < I; ++K)
            t = mm16_nn(Ts + (16 * I) * TLD + 16 * K, TLD, Xs + (16 * K) * TLD + 16 * J, TLD, t, false, li, lq);
        store_d16(Xs + (16 * I) * TLD + 16 * J, TLD, t, li, lq);
    };
    auto finish_ij = [&](int I, int J) {                 // Xs[I][J] <- -X[I][I] * Xs[I][J]
        const v4d x = mm16_nn(Xs + (16 * I) * TLD + 16 * I, TLD, Xs + (16 * I) * TLD + 16 * J, TLD, zero, true, li, lq);
        store_d16(Xs + (16 * I) * TLD + 16 * J, TLD, x, li, lq);
    };
    auto need = [&](int w) { tile_wait(fl + w, spins); };
    auto need_cols = [&](int c_end) {                    // the columns [0, c_end) of L are in Ts
        for (int p = 0; p < c_end / 16; ++p) tile_wait(fl + TF_CS + p, spins);
    };
    auto zero_x_upper = [&]() {                          // the six 16x16 blocks above the block diagonal of X (nobody reads them in here)
        for (int i = lane; i < 6 * 256; i += 64) {
            const int b = i >> 8, e = i & 255;
            const int I = (b < 3) ? 0 : ((b < 5) ? 1 : 2);
            const int J = (b < 3) ? b + 1 : ((b < 5) ? b - 1 : 3);
            Xs[(16 * I + (e >> 4)) * TLD + 16 * J + (e & 15)] = 0.0;
        }
    };
    auto dbgp = [&](int p) -> unsigned long long* { return side.stamps ? side.stamps + 32 + 3 * p : nullptr; };
    auto idle = [&](int tick) { side.idle(tick, wave, lane); };
    // accumulate one more product into a sum that waits in its block of Xs:  Xs[I][J] += L[I][K] X[K][J]
    auto add_ij = [&](int I, int J, int K) {
        v4d t = load_d16(Xs + (16 * I) * TLD + 16 * J, TLD, li, lq);
        t = mm16_nn(Ts + (16 * I) * TLD + 16 * K, TLD, Xs + (16 * K) * TLD + 16 * J, TLD, t, false, li, lq);
        store_d16(Xs + (16 * I) * TLD + 16 * J, TLD, t, li, lq);
    };
    auto sum_upto = [&](int I, int J, int Kend) {        // Xs[I][J] <- sum_{K=J..Kend-1} L[I][K] X[K][J]
        v4d t = zero;
        for (int K = J; K < Kend; ++K)
            t = mm16_nn(Ts + (16 * I) * TLD + 16 * K, TLD, Xs + (16 * K) * TLD + 16 * J, TLD, t, false, li, lq);
        store_d16(Xs + (16 * I) * TLD + 16 * J, TLD, t, li, lq);
    };
    if (wave == 0) {
        panel_lead<0>(a, Ts, rinvs, Wk, lane, bad, dbgp(0));
        if (lane == 0) TILE_STAMP(1);
        zero_x_upper();
        inv16(Ts, rinvs, Xs, 0, lane);                       // (its own stores of panel 0: in order)
        tile_set(fl + TF_I + 0, lane);
        {   // X_22 beside wave 2's panel
            double x[1][4];
            solve16_rhs_identity(x[0], lane);
            solve16_follow<2, 1>(Wk, x, lane, spins);
            solve16_store(x[0], Xs + 32 * TLD + 32, TLD, lane);
        }
        tile_set(fl + TF_I + 2, lane);
        if (lane == 0) TILE_STAMP(7);
        need_cols(48);
        sum_ij(3, 2);                                        // S_32 = L_32 X_22
        {   // X_33 beside wave 3's panel
            double x[1][4];
            solve16_rhs_identity(x[0], lane);
            solve16_follow<3, 1>(Wk, x, lane, spins);
            solve16_store(x[0], Xs + 48 * TLD + 48, TLD, lane);
        }
        tile_set(fl + TF_I + 3, lane);
        if (lane == 0) TILE_STAMP(8);
        finish_ij(3, 2);                                     // (the last block row: one product with X_33 per wave)
    } else if (wave == 1) {
        panels_follow_mfma<16>(acc, Wk, li, lq, spins, idle);
        panel_to_rows<16>(s0, acc, a, Ts, lane, li, lq);
        if (lane == 63) TILE_STAMP(57);
        panel_lead<1>(a, Ts, rinvs, Wk, lane, bad, dbgp(1));
        if (lane == 16) TILE_STAMP(2);
        inv16(Ts, rinvs, Xs, 16, lane);
        tile_set(fl + TF_I + 1, lane);
        need(TF_I + 0); need_cols(16);
        sum_ij(1, 0); finish_ij(1, 0);
        tile_set(fl + TF_X10, lane);
        // everything of block rows 2 and 3 that needs neither X_22 nor panel 2's columns, while panel 2 is being factored
        sum_ij(2, 1);                                        // L_21 X_11
        sum_ij(2, 0);                                        // L_20 X_00 + L_21 X_10
        sum_upto(3, 1, 2);                                   // L_31 X_11
        sum_upto(3, 0, 2);                                   // L_30 X_00 + L_31 X_10
        need(TF_I + 2);
        finish_ij(2, 1);
        tile_set(fl + TF_X21, lane);
        finish_ij(2, 0);
        tile_set(fl + TF_X20, lane);
        need_cols(48);
        add_ij(3, 1, 2);                                     // + L_32 X_21
        need(TF_I + 3);
        finish_ij(3, 1);
    } else if (wave == 2) {
        panels_follow_mfma<32>(acc, Wk, li, lq, spins, idle);
        panel_to_rows<32>(s0, acc, a, Ts, lane, li, lq);
        if (lane == 63) TILE_STAMP(58);
        panel_lead<2>(a, Ts, rinvs, Wk, lane, bad, dbgp(2));
        if (lane == 32) TILE_STAMP(3);
        need(TF_X20);                                        // (wave 1 formed the sums of block (3, 0) up to K = 1 before it)
        add_ij(3, 0, 2);                                     // + L_32 X_20
        need(TF_I + 3);
        finish_ij(3, 0);
    } else if (wave == 3) {
        panels_follow_mfma<48>(acc, Wk, li, lq, spins, idle);
        panel_to_rows<48>(s0, acc, a, Ts, lane, li, lq);
        if (lane == 63) TILE_STAMP(59);
        panel_lead<3>(a, Ts, rinvs, Wk, lane, bad, dbgp(3));
        if (lane == 48) TILE_STAMP(4);
    } else {
        side.extra(wave - 4, lane, fl);
    }
    if (tid == 0) TILE_STAMP(13);
    // a non-positive pivot (seen by the lanes of its panel) or a wait that gave up, in any wave, is everybody's verdict
    if (wave < 4 && (__builtin_amdgcn_ballot_w64(bad) != 0ull || spins >= TILE_SPIN_LIMIT) && lane == 0) fl[TF_BAD + wave] = 1;
    __syncthreads();
    bad = fl[TF_BAD + 0] != 0 || fl[TF_BAD + 1] != 0 || fl[TF_BAD + 2] != 0 || fl[TF_BAD + 3] != 0;
    if (tid == 0) TILE_STAMP(14);
}

// 64x64 tile: global (row stride ld) -> LDS (row stride TLD), 256 threads.
__device__ __forceinline__ void tile_g2s(const double* __restrict__ g, int64_t ld, double* s, int tid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256;            // 2048 v2d
        const int r = idx >> 5, c = (idx & 31) * 2;
        *reinterpret_cast<v2d*>(s + r * TLD + c) = *reinterpret_cast<const v2d*>(g + (int64_t)r * ld + c);
    }
}
__device__ __forceinline__ void tile_s2g(const double* s, double* __restrict__ g, int64_t ld, int tid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 5, c = (idx & 31) * 2;
        *reinterpret_cast<v2d*>(g + (int64_t)r * ld + c) = *reinterpret_cast<const v2d*>(s + r * TLD + c);
    }
}

struct StepArgs {
    double* S;            // bsp x bsp Schur block (lower tiles valid); trailing tiles updated in place
    double* L;            // factor block
    double* X;            // inverse block (diagonal tile written here)
    int64_t ld;
    int j;                // panel index
    int nt;               // tiles per dimension
    int* info;
    int blk;              // block id reported on a non-positive pivot (1-based)
    int64_t pS, pL, pX;   // per-problem strides of S, L and X (blockIdx.y); L may be a one-block work buffer
    int blk_per_problem;  // reported id = blk + blockIdx.y * blk_per_problem
    int cend;             // potrf_update: column tiles j+1 .. cend-1 only (nt: the whole trailing block)
    unsigned long long* dbg;   // diagnostic stamps of workgroup 1 of step 0 (tests), else nullptr
    // One problem: the inverse Linv is assembled ROW BY ROW beside the panel steps instead of by recursive
    // doubling afterwards (8 dependent launches per block off the chain): workgroups xrow_first .. of the
    // launch of step j compute the tiles X[xrow, c], c < xrow, of row xrow = j - 1, whose inputs (L[xrow, :],
    // the rows of X above, X[xrow, xrow]) are final by then.  xrow < 0: none.
    int xrow = -1;
    int xrow_first = 0;        // first blockIdx.x of the X-row workgroups (4 per tile: 16-column strips)
    // One problem, look-ahead form of the fused step: the diagonal tile j was factored by the PREVIOUS launch (its
    // L_jj, X_jj are in global memory), nobody re-factors it; workgroup 0 owns tile (j+1, j+1): it forms L[j+1,j],
    // updates its tile in LDS and factors it for the next launch, while the other workgroups do panel + update of
    // theirs.  The chain per step is  update (j+1,j+1) -> factor  instead of  factor -> update everything.
    int lookahead = 0;
};

// X[r, c][:, strip] = -X_rr * sum_{p = c}^{r-1} L[r, p] X[p, c][:, strip]   (block forward substitution: only the
// diagonal inverses multiply, like inside the tile).  One workgroup per 64 x 16 output strip: wave w forms rows
// 16 w .. of the sum (operands straight from L2), the four strips meet in LDS for the product with X_rr.
__device__ __forceinline__ void xrow_strip(const StepArgs& sa, int q, double* sm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld;
    const int r = sa.xrow, c = q >> 2, cs = q & 3;
    const int64_t orow = (int64_t)r * 64, ocol = (int64_t)c * 64 + 16 * cs;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ arow = sa.L + (orow + 16 * wave + li) * ld;
    for (int p = c; p < r; ++p) {
        const int64_t op = (int64_t)p * 64;
        v2d a[8];
        double b0[8], b1[8];
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            a[kg] = *reinterpret_cast<const v2d*>(arow + op + 8 * kg + 2 * lq);
            b0[kg] = sa.X[(op + 8 * kg + 2 * lq) * ld + ocol + li];
            b1[kg] = sa.X[(op + 8 * kg + 2 * lq + 1) * ld + ocol + li];
        }
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].x, b0[kg], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].y, b1[kg], acc, 0, 0, 0);
        }
    }
    constexpr int XL = 18;                                   // LDS image T[64][16] with row stride 18
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) sm[(16 * wave + lq + 4 * qq) * XL + li] = acc[qq];
    __syncthreads();
    // rows 16 w .. of  -X_rr T ; X_rr lower triangular: k groups 0 .. 2 w + 1
    v4d res = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ xrr = sa.X + (orow + 16 * wave + li) * ld + orow;
    for (int kg = 0; kg < 2 * wave + 2; ++kg) {
        const v2d xv = *reinterpret_cast<const v2d*>(xrr + 8 * kg + 2 * lq);
        res = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.x, sm[(8 * kg + 2 * lq) * XL + li], res, 0, 0, 0);
        res = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.y, sm[(8 * kg + 2 * lq + 1) * XL + li], res, 0, 0, 0);
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) sa.X[(orow + 16 * wave + lq + 4 * qq) * ld + ocol + li] = -res[qq];
}


// Fused panel step: grid.x = 1 + m (m + 1) / 2, m = nt - j - 1  (cend < nt: 1 + the tiles of columns
// j+1 .. cend-1 only; the rest of the trailing block is updated per panel by the GEMM kernel).  Workgroup 0 factors and inverts
// the diagonal tile and writes it; every other workgroup does the same factorisation for itself
// (idle CUs otherwise), then forms its two panel tiles and updates its trailing tile.  With
// grid.x = 1 it is the tile kernel of the split form (batches, large blocks).  The template
// parameter only keeps the kernel's symbol (profiles, tests); it is always false.
template <bool UNUSED>
__global__ __launch_bounds__(256, 2) void potrf_step(StepArgs sa) {
    sa.S += (int64_t)blockIdx.y * sa.pS;
    sa.L += (int64_t)blockIdx.y * sa.pL;
    sa.X += (int64_t)blockIdx.y * sa.pX;
    sa.blk += (int)blockIdx.y * sa.blk_per_problem;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (sa.xrow >= 0 && (int)blockIdx.x >= sa.xrow_first) {          // X-row role (one problem, fused steps)
        xrow_strip(sa, (int)blockIdx.x - sa.xrow_first, smem);
        return;
    }
    double* Ts = smem;
    double* Xs = Ts + TILE_ELEMS;
    // the tile-only launch (grid.x = 1) allocates up to here only (POTRF_TILE_LDS): with 75 KB instead
    // of 144 KB a GEMM workgroup of another stream still fits on the CU beside it
    double* Wk = Xs + TILE_ELEMS;               // WK_ELEMS
    double* rinvs = Wk + WK_ELEMS;           // 64
    double* As = rinvs + 64;                    // fused form only: the two panel tiles
    double* Bs = As + TILE_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld;
    const int64_t oj = (int64_t)sa.j * 64;

    int r = 0, c = 0;
    const int w = (int)blockIdx.x;
    const bool la = sa.lookahead != 0;
    if (la) {                                    // every workgroup owns a tile of the trailing block; 0 -> (j+1, j+1)
        int t = w, rr = 0;
        while ((rr + 1) * (rr + 2) / 2 <= t) ++rr;
        r = sa.j + 1 + rr;
        c = sa.j + 1 + (t - rr * (rr + 1) / 2);
    } else if (w > 0) {
        if (sa.cend >= sa.nt) {                  // the whole trailing block, row by row
            int t = w - 1, rr = 0;
            while ((rr + 1) * (rr + 2) / 2 <= t) ++rr;
            r = sa.j + 1 + rr;
            c = sa.j + 1 + (t - rr * (rr + 1) / 2);
        } else {                                 // columns j+1 .. cend-1 only (two-level form), column by column
            int t = w - 1;
            c = sa.j + 1;
            while (t >= sa.nt - c) { t -= sa.nt - c; ++c; }
            r = c + t;
        }
    }
    // the C tile this wave will update (rows 16*wave.., MFMA C/D layout), fetched now, used last
    double* Sg = sa.S + (int64_t)r * 64 * ld + (int64_t)c * 64;
    v4d cpre[4];
    if (w > 0 || la) {
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cpre[Jb][q] = Sg[(int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li];
    }
    bool bad = false;
    const bool stamp = !la && sa.dbg && sa.j == 0 && w == 1 && tid == 0 && blockIdx.y == 0;
    if (la) {
        // X_jj of the previous launch and this workgroup's panel tiles: three independent tile loads
        tile_g2s(sa.X + oj * ld + oj, ld, Xs, tid);
        tile_g2s(sa.S + (int64_t)r * 64 * ld + oj, ld, As, tid);
        if (c != r) tile_g2s(sa.S + (int64_t)c * 64 * ld + oj, ld, Bs, tid);
        __syncthreads();
    } else {
        tile_g2s(sa.S + oj * ld + oj, ld, Ts, tid);
        if (w > 0) {                                       // the two panel tiles (round 5: no longer staged beside the first panel -- every wave is busy there)
            tile_g2s(sa.S + (int64_t)r * 64 * ld + oj, ld, As, tid);
            if (c != r) tile_g2s(sa.S + (int64_t)c * 64 * ld + oj, ld, Bs, tid);
        }
        NoSide side;
        __syncthreads();
        if (stamp) sa.dbg[0] = __builtin_amdgcn_s_memtime();
        tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, side);
        if (stamp) sa.dbg[1] = __builtin_amdgcn_s_memtime();
        if (w == 0) {
            if (bad && tid == 0) atomicCAS(sa.info, 0, sa.blk);
            tile_s2g(Ts, sa.L + oj * ld + oj, ld, tid);
            tile_s2g(Xs, sa.X + oj * ld + oj, ld, tid);
            return;
        }
    }
    // ---- panel rows: Lr = As Xs^T, Lc = Bs Xs^T.  k runs outermost in groups of 8: a lane fetches
    // two consecutive k of its operand row with ONE ds_read_b128 (every LDS instruction issued
    // beside fp64 MFMAs costs ~26 cycles of MFMA issue, tools/mb3.hip) and feeds them to two MFMAs
    // (k-slot permutation as in gemm_f64.hpp).  X is lower triangular: column block Jb needs the
    // k groups 0 .. 2 Jb + 1 only.
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    v4d lr[4], lc[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) { lr[Jb] = zero; lc[Jb] = zero; }
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
        v2d bv = av;
        if (c != r) bv = *reinterpret_cast<const v2d*>(Bs + (16 * wave + li) * TLD + k);
#pragma unroll
        for (int Jb = kg / 2; Jb < 4; ++Jb) {
            const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, lr[Jb], 0, 0, 0);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, lr[Jb], 0, 0, 0);
            if (c != r) {
                lc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.x, xv.x, lc[Jb], 0, 0, 0);
                lc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.y, xv.y, lc[Jb], 0, 0, 0);
            }
        }
    }
    if (stamp) sa.dbg[2] = __builtin_amdgcn_s_memtime();
    __syncthreads();                                   // every wave is done reading As / Bs
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, lr[Jb], li, lq);
        if (c != r) store_d16(Bs + (16 * wave) * TLD + 16 * Jb, TLD, lc[Jb], li, lq);
    }
    __syncthreads();
    if (stamp) sa.dbg[3] = __builtin_amdgcn_s_memtime();
    if (c == sa.j + 1) tile_s2g(As, sa.L + (int64_t)r * 64 * ld + oj, ld, tid);
    // ---- S[r,c] -= Lr Lc^T ; wave owns rows 16*wave.., on a diagonal tile only Jb <= wave matters
    const double* Lcs = (c != r) ? Bs : As;
    const int jb_end = (c == r) ? wave + 1 : 4;
    // The product is accumulated from zero and subtracted ONCE (one rounding at the magnitude of
    // S) -- an fma chain that starts from S rounds 64 times at that magnitude, which is what
    // LAPACK's dsyrk does not do and what cost two digits on ill-conditioned Schur complements.
    v4d pacc[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) pacc[Jb] = zero;
    if (c != r) strip_nt<4>(As + (16 * wave + li) * TLD, Lcs, pacc, li, lq);
    else strip_nt_diag(wave, As + (16 * wave + li) * TLD, Lcs, pacc, li, lq);
    if (stamp) sa.dbg[4] = __builtin_amdgcn_s_memtime();
    if (la && w == 0) {
        // look-ahead: the updated tile (j+1, j+1) stays in LDS and is factored here, for the next launch
        __syncthreads();                                   // (Ts is free: nothing of this launch lives there)
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) {
            if (Jb < jb_end) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    Ts[(16 * wave + lq + 4 * q) * TLD + 16 * Jb + li] = cpre[Jb][q] - pacc[Jb][q];
            }
        }
        __syncthreads();
        NoSide none;
        tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, none);
        if (bad && tid == 0) atomicCAS(sa.info, 0, sa.blk);
        const int64_t o1 = oj + 64;
        tile_s2g(Ts, sa.L + o1 * ld + o1, ld, tid);
        tile_s2g(Xs, sa.X + o1 * ld + o1, ld, tid);
        return;
    }
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        if (Jb < jb_end) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Sg[(int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li] = cpre[Jb][q] - pacc[Jb][q];
        }
    }
    if (stamp) sa.dbg[5] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------------------------------
// Split form of the panel step for batches of problems (throughput regime): after the tile launch
// (potrf_step<false> with grid.x = 1) the panel and the trailing update run as two small kernels
// that feed the MFMAs straight from global memory / L2 -- no LDS, no barriers, ~100 VGPRs, so
// several workgroups are resident per CU and one's loads overlap another's MFMAs.
//   potrf_panel : L[r,j] = S[r,j] X_jj^T                 grid (m, B)
//   potrf_update: S[r,c] -= L[r,j] L[c,j]^T  (j < c <= r, c < cend)  grid (sum_c (nt - c), B)
// A lane fetches two consecutive k of its operand row (16 B) and feeds them to two MFMAs (k-slot
// permutation as in gemm_f64.hpp); each wave owns a 16-row strip of the 64x64 tile.
__global__ __launch_bounds__(256, 2) void potrf_panel(StepArgs sa) {
    sa.S += (int64_t)blockIdx.y * sa.pS;
    sa.L += (int64_t)blockIdx.y * sa.pL;
    sa.X += (int64_t)blockIdx.y * sa.pX;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld, oj = (int64_t)sa.j * 64;
    const int64_t R0 = (int64_t)(sa.j + 1 + blockIdx.x) * 64 + 16 * wave;
    v2d a[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
        a[kg] = *reinterpret_cast<const v2d*>(sa.S + (R0 + li) * ld + oj + 8 * kg + 2 * lq);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kg = 0; kg < 2 * Jb + 2; ++kg) {          // X lower triangular
            const v2d x = *reinterpret_cast<const v2d*>(sa.X + (oj + 16 * Jb + li) * ld + oj + 8 * kg + 2 * lq);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].y, x.y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sa.L[(R0 + lq + 4 * q) * ld + oj + 16 * Jb + li] = acc[q];
    }
}

__global__ __launch_bounds__(256, 2) void potrf_update(StepArgs sa) {
    sa.S += (int64_t)blockIdx.y * sa.pS;
    sa.L += (int64_t)blockIdx.y * sa.pL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld, oj = (int64_t)sa.j * 64;
    int t = blockIdx.x, c = sa.j + 1;            // column by column: column c has nt - c row tiles
    while (t >= sa.nt - c) { t -= sa.nt - c; ++c; }
    const int r = c + t;
    const int64_t R0 = (int64_t)r * 64 + 16 * wave, C0 = (int64_t)c * 64;
    const int jb_end = (c == r) ? wave + 1 : 4;
    v2d a[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
        a[kg] = *reinterpret_cast<const v2d*>(sa.L + (R0 + li) * ld + oj + 8 * kg + 2 * lq);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        if (Jb < jb_end) {
            double* cg = sa.S + R0 * ld + C0 + 16 * Jb;
            v4d cur;
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = cg[(int64_t)(lq + 4 * q) * ld + li];
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};         // product from zero, one subtraction (rounding)
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const v2d b = *reinterpret_cast<const v2d*>(sa.L + (C0 + 16 * Jb + li) * ld + oj + 8 * kg + 2 * lq);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kg].y, b.y, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cg[(int64_t)(lq + 4 * q) * ld + li] = cur[q] - acc[q];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Batches, round 3: the in-block Cholesky advances in 128-column DIAGONAL BLOCKS; everything below a diagonal block
// is level-3 work on the GEMM kernel (K = 128 / 256) instead of rank-64 steps fed from HBM:
//
//   potrf_diag128(j)  one workgroup per problem: the 128 x 128 diagonal block at tile j (j even), in LDS --
//                       L00 = chol(S00), X00 = L00^-1          (tile_potrf_inv)
//                       L10 = S10 X00^T                          S11 -= L10 L10^T   (product from zero, one subtraction)
//                       L11 = chol(S11), X11 = L11^-1
//                       X10 = -X11 (L10 X00)                     (the 128 x 128 inverse X_A = [X00 0; X10 X11])
//   GEMM              L[j+2.., j..j+1] = S[j+2.., j..j+1] X_A^T                          (K = 128, X_A lower triangular)
//   GEMM              first half of a 256-column panel: its second half S[j+2.., j+2..j+3] -= L[j+2.., j..j+1] L[j+2..j+3, j..j+1]^T
//                     second half: the rank-256 update of everything right of the panel (as before)
//
// 22 launches per 1024-block instead of 51 (16 x (tile, potrf_panel, potrf_update) + 3), 8 dependent tile kernels per
// block instead of 16, and the level-64 doubling products of the block inverse come out of this kernel.  potrf_panel /
// potrf_update moved 3.1 TB/s of S through HBM at 6 - 13 TF/s for 25 ms of every batch-32 factorisation.
// The rows below a diagonal block now meet the 128 x 128 inverse (a product of two computed tile inverses, like every
// level of the block inverse Linv_i itself) instead of two 64 x 64 ones: the factor's error carries cond(128-block) eps
// where it carried cond(64-tile) eps (BASELINE workloads: cond(L_i) <= 840 for the whole 1024-block).
// LDS: three tiles + scratch = 111 KB (a GEMM workgroup of another stream still fits beside it).
__global__ __launch_bounds__(256, 1) void potrf_diag128(StepArgs sa) {
    sa.S += (int64_t)blockIdx.y * sa.pS;
    sa.L += (int64_t)blockIdx.y * sa.pL;
    sa.X += (int64_t)blockIdx.y * sa.pX;
    sa.blk += (int)blockIdx.y * sa.blk_per_problem;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ts = smem;
    double* Xs = Ts + TILE_ELEMS;
    double* As = Xs + TILE_ELEMS;
    double* Wk = As + TILE_ELEMS;               // WK_ELEMS
    double* rinvs = Wk + WK_ELEMS;           // 64
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld;
    const int64_t o0 = (int64_t)sa.j * 64, o1 = o0 + 64;
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    // S11 (this wave's 16-row strip, MFMA C/D layout; lower blocks only): requested now, used after the first tile
    v4d cpre[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            cpre[Jb][q] = (Jb <= wave) ? sa.S[(o1 + 16 * wave + lq + 4 * q) * ld + o1 + 16 * Jb + li] : 0.0;
    bool bad = false;
    tile_g2s(sa.S + o0 * ld + o0, ld, Ts, tid);
    tile_g2s(sa.S + o1 * ld + o0, ld, As, tid);           // S10
    NoSide side;
    __syncthreads();
    tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, side);
    tile_s2g(Ts, sa.L + o0 * ld + o0, ld, tid);
    tile_s2g(Xs, sa.X + o0 * ld + o0, ld, tid);
    // ---- L10 = S10 X00^T (X00 lower triangular: column block Jb needs the k groups 0 .. 2 Jb + 1)
    v4d lr[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) lr[Jb] = zero;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
#pragma unroll
        for (int Jb = kg / 2; Jb < 4; ++Jb) {
            const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, lr[Jb], 0, 0, 0);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, lr[Jb], 0, 0, 0);
        }
    }
    __syncthreads();                                   // every wave is done reading S10
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, lr[Jb], li, lq);
    __syncthreads();
    tile_s2g(As, sa.L + o1 * ld + o0, ld, tid);
    // ---- S11 - L10 L10^T (lower 16 x 16 blocks) and W = L10 X00 (X00[k][c] = 0 for k < c: k groups 2 Jb .. 7)
    v4d pacc[4], wv[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) { pacc[Jb] = zero; wv[Jb] = zero; }
    strip_syrk_and_w_w(wave, As + (16 * wave + li) * TLD, As, Xs, pacc, wv, li, lq);
    __syncthreads();                                   // L10 (As) and X00 (Xs) have been read by everyone
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, wv[Jb], li, lq);            // W replaces L10
        if (Jb <= wave) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Ts[(16 * wave + lq + 4 * q) * TLD + 16 * Jb + li] = cpre[Jb][q] - pacc[Jb][q];
        }
    }
    __syncthreads();
    NoSide none;
    tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, none);
    if (bad && tid == 0) atomicCAS(sa.info, 0, sa.blk);
    tile_s2g(Ts, sa.L + o1 * ld + o1, ld, tid);
    tile_s2g(Xs, sa.X + o1 * ld + o1, ld, tid);
    // ---- X10 = -X11 W (X11 lower triangular: the rows of wave w need the k groups 0 .. 2 w + 1)
    v4d xr[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) xr[Jb] = zero;
    strip_tri_nn_w(wave, Xs + (16 * wave + li) * TLD, As, xr, li, lq);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) sa.X[(o1 + 16 * wave + lq + 4 * q) * ld + o0 + 16 * Jb + li] = -xr[Jb][q];
}
constexpr size_t POTRF_DIAG128_LDS = (3 * TILE_ELEMS + WK_ELEMS + 64) * sizeof(double);

// The same block with TWO LDS tiles (77 KB: two GEMM workgroups of another stream fit beside it instead of one): S10 goes
// straight from L2 into registers as the MFMA A operand of L10 = S10 X00^T (as potrf_panel did), L10 takes the place of
// L00 once that is stored, and W = L10 X00 waits in registers (16 doubles per thread) across the second tile
// factorisation before it takes the place of L11.  Same operations in the same order as potrf_diag128: bitwise equal.
__global__ __launch_bounds__(256, 1) void potrf_diag128_slim(StepArgs sa) {
    sa.S += (int64_t)blockIdx.y * sa.pS;
    sa.L += (int64_t)blockIdx.y * sa.pL;
    sa.X += (int64_t)blockIdx.y * sa.pX;
    sa.blk += (int)blockIdx.y * sa.blk_per_problem;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ts = smem;
    double* Xs = Ts + TILE_ELEMS;
    double* Wk = Xs + TILE_ELEMS;               // WK_ELEMS
    double* rinvs = Wk + WK_ELEMS;           // 64
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = sa.ld;
    const int64_t o0 = (int64_t)sa.j * 64, o1 = o0 + 64;
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    v4d cpre[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            cpre[Jb][q] = (Jb <= wave) ? sa.S[(o1 + 16 * wave + lq + 4 * q) * ld + o1 + 16 * Jb + li] : 0.0;
    // S10 as the A operand of this wave's 16-row strip: two consecutive k per 8-wide k group
    v2d s10[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
        s10[kg] = *reinterpret_cast<const v2d*>(sa.S + (o1 + 16 * wave + li) * ld + o0 + 8 * kg + 2 * lq);
    bool bad = false;
    tile_g2s(sa.S + o0 * ld + o0, ld, Ts, tid);
    NoSide none;
    __syncthreads();
    tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, none);
    tile_s2g(Ts, sa.L + o0 * ld + o0, ld, tid);
    tile_s2g(Xs, sa.X + o0 * ld + o0, ld, tid);
    // ---- L10 = S10 X00^T
    v4d lr[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) lr[Jb] = zero;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
#pragma unroll
        for (int Jb = kg / 2; Jb < 4; ++Jb) {
            const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(s10[kg].x, xv.x, lr[Jb], 0, 0, 0);
            lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(s10[kg].y, xv.y, lr[Jb], 0, 0, 0);
        }
    }
    __syncthreads();                                   // L00 has left Ts (tile_s2g reads complete before the stores below)
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) store_d16(Ts + (16 * wave) * TLD + 16 * Jb, TLD, lr[Jb], li, lq);
    __syncthreads();
    tile_s2g(Ts, sa.L + o1 * ld + o0, ld, tid);
    // ---- S11 - L10 L10^T (lower 16 x 16 blocks) and W = L10 X00
    v4d pacc[4], wv[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) { pacc[Jb] = zero; wv[Jb] = zero; }
    strip_syrk_and_w_w(wave, Ts + (16 * wave + li) * TLD, Ts, Xs, pacc, wv, li, lq);
    __syncthreads();                                   // L10 (Ts) and X00 (Xs) have been read by everyone
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) {
        if (Jb <= wave) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Ts[(16 * wave + lq + 4 * q) * TLD + 16 * Jb + li] = cpre[Jb][q] - pacc[Jb][q];
        }
    }
    __syncthreads();
    tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, none);
    if (bad && tid == 0) atomicCAS(sa.info, 0, sa.blk);
    tile_s2g(Ts, sa.L + o1 * ld + o1, ld, tid);
    tile_s2g(Xs, sa.X + o1 * ld + o1, ld, tid);
    __syncthreads();                                   // L11 has left Ts
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) store_d16(Ts + (16 * wave) * TLD + 16 * Jb, TLD, wv[Jb], li, lq);     // W takes its place
    __syncthreads();
    // ---- X10 = -X11 W
    v4d xr[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) xr[Jb] = zero;
    strip_tri_nn_w(wave, Xs + (16 * wave + li) * TLD, Ts, xr, li, lq);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) sa.X[(o1 + 16 * wave + lq + 4 * q) * ld + o0 + 16 * Jb + li] = -xr[Jb][q];
}
constexpr size_t POTRF_DIAG128_SLIM_LDS = (2 * TILE_ELEMS + WK_ELEMS + 64) * sizeof(double);


constexpr size_t POTRF_STEP_LDS = (4 * TILE_ELEMS + WK_ELEMS + 64) * sizeof(double);

// Stand-alone tile kernel (tests): S (ld 64) -> L, X.
__global__ __launch_bounds__(256, 2) void potrf_tile_kernel(const double* S, double* L, double* X, int* info,
                                                            unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ts = smem;
    double* Xs = Ts + TILE_ELEMS;
    double* Wk = Xs + TILE_ELEMS;
    double* rinvs = Wk + WK_ELEMS;
    const int tid = threadIdx.x;
    tile_g2s(S, 64, Ts, tid);
    __syncthreads();
    bool bad = false;
    NoSide side;
    side.stamps = stamps;
    if (stamps && tid == 0) stamps[15] = __builtin_amdgcn_s_memtime();
    tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, side);
    if (bad && tid == 0) atomicCAS(info, 0, 1);
    tile_s2g(Ts, L, 64, tid);
    tile_s2g(Xs, X, 64, tid);
    if (stamps && tid == 0) stamps[16] = __builtin_amdgcn_s_memtime();
}
constexpr size_t POTRF_TILE_LDS = (2 * TILE_ELEMS + WK_ELEMS + 64) * sizeof(double);

}  // namespace gmrf
