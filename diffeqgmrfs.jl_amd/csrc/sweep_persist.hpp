// One problem: a whole block-bidiagonal sweep
//   forward_solve  (/root/reference/src/tridiagonal_cholesky.jl:43-52)   y_i = L_i^-1 (b_i - C_{i-1} y_{i-1})
//   backward_solve (/root/reference/src/tridiagonal_cholesky.jl:24-33)   x_i = L_i^-T (y_i - C_i^T x_{i+1})
// as ONE persistent launch instead of two dependent launches per block.  Round 5.
//
// The arithmetic is that of the launch-per-product kernels (sweep.hpp), body for body -- the same workgroup decomposition, the same
// summation order, results bitwise equal -- run by `nw` resident workgroups over "virtual" block indices.  What is gone is the
// launch boundary between two dependent products; what takes its place is NOT a grid barrier (an all-to-all barrier through
// memory is three fabric hops -- drained stores, the flag, the loads behind it -- and measured SLOWER than the 1.7 us boundary:
// burgers512x64 5.6 us per product against 4.3, darcy256 k = 64 14 us against 11) but a DATA FLOW: every panel element is
// written exactly once per launch by one 8-byte `sc1` store, the panels a product reads from other workgroups (T: the
// right-hand side minus the coupling product, Y: the results) are filled with a sentinel before the launch, and a body simply
// repeats the `sc1` loads of an input chunk until none of its values is the sentinel (SweepVec::ldw, sweep.hpp) -- one fabric hop
// per product, the guide's data-tagged "allgather" edge with the fp64 value as its own tag (MI355X_MICROARCH.md, price list).
// A workgroup runs ahead as far as its inputs allow, and a body requests its own piece of the matrix BEFORE it waits for its
// input (k = 1: the whole piece, into registers), so that the matrix's trip overlaps the input's.  The in-place update of the launch-per-product form (P_i becomes
// P_i - C y) cannot stay in place here (the old value is no sentinel): it goes to the panel T, the rows outside the coupling
// window are copied along.
//
// Every workgroup must be resident (the host claims the whole chip for the handle: persist_plan); every repetition is bounded
// (`spin_limit` ticks of the 100 MHz clock): a wave that gives up sets the abort words (device + mapped host memory), every
// other wait ends on the device word, the launch drains with garbage in the panels, the host sees its word at its next
// synchronisation and repeats the solve with a launch per product, which the handle keeps from then on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sweep.hpp"

namespace gmrf {

// Block sizes up to this: with 1024 a product is at most ONE virtual block per workgroup (256 workgroups: 1024 rows / 4, 64 x 4 tiles)
// and the persistent form wins (darcy256, elliptic512, burgers512x64).  Blocks of 4096 were measured with the limit at 4096
// (burgers4096x512: four virtual blocks per workgroup and product, each with its own gather): mean 48.6 -> 63.7 ms, 64 samples
// 124.6 -> 138.4 ms -- there a launch per product, whose boundary is small beside its 32 - 128 MB products, stays.
constexpr int SWEEP_PERSIST_XMAX = 1024;

struct SweepPersistArgs {
    const double* C; const double* Linv;     // the factor: coupling windows, explicit inverses of the diagonal blocks
    const double* Pin; double* T; double* Yout;   // right-hand panel (read only), intermediate panel, result panel: [kp][npad]
    int N, bsp, cm, rm, kp, backward, nw;
    int64_t npad, ldc, cstride, bstride;
    const int* kst; const int* mend;
    unsigned* abort_w;                       // device word
    unsigned* host_abort;                    // mapped host word (the host reads it without a copy)
    unsigned spin_limit;
    int pause;                               // s_sleep units (64 clocks) between two looks at an input panel
    int dbg;
};

// the panels other workgroups' results are read from, before the launch: every element = the sentinel
__global__ __launch_bounds__(256) void sweep_fill_sentinel(double* a, double* b, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        const sw_v4u v = (sw_v4u){SWEEP_SENT_LO, SWEEP_SENT_HI, SWEEP_SENT_LO, SWEEP_SENT_HI};
        *reinterpret_cast<sw_v4u*>(a + i) = v;
        *reinterpret_cast<sw_v4u*>(b + i) = v;
    }
}

typedef __attribute__((address_space(3))) const double sp_lds_double;

// ---- k = 1: the sums of sweep_gemv_n / sweep_gemv_t<., 8> (sweep.hpp), term for term in the same order, as a DATA-FLOW body:
// the block's own piece of the matrix is requested into registers FIRST (it waits for nobody), then `gather()` brings the
// input vector into LDS (the workgroup polls it there: SweepVec::ldw; ends with a barrier), then the sums run from registers
// and LDS.  The matrix's trip through the memory system thus overlaps the input's -- the one thing a launch per product cannot do.

// A lone workgroup per CU issues about one instruction per 8 cycles and wave: the bodies below count instructions.  Which
// pieces of a thread's sequence exist is decided by SCALAR comparisons (whole pieces: the same for every lane of the wave /
// thread of the workgroup; only the last, partial piece carries a per-lane predicate), and the matrix is read by buffer
// loads whose per-piece stride sits in the scalar offset (no 64-bit address arithmetic per piece).

// forward: out[row] = sum_k Mat[row][k] x[k] -- one wave per row; per lane the even / odd elements of k = kb + 2 lane + 128 u in two
// sums, ascending u; the odd last element on lane 0; the wave's xor-shuffle reduction
template <bool TRI, class Gather>
__device__ __forceinline__ void sweep_gemv_n_flow(const SweepArgs& s, int vb, sp_lds_double* xs, double* res, Gather&& gather) {
    constexpr int NM = SWEEP_PERSIST_XMAX / 128;      // 16-byte pieces per lane: a whole row
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row = vb * 4 + wv;                       // (wave-uniform)
    const bool valid = row < s.rows;
    const int ke = valid ? (TRI ? (row + 1) : s.kdim) : 0;
    const int kb = (!TRI && s.kst && valid) ? __builtin_amdgcn_readfirstlane(s.kst[row >> 6]) : 0;
    const int ke2 = ke & ~1;
    const int span = max(ke2 - kb, 0), nfull = span >> 7, rem = span & 127;      // whole pieces of 128 elements; the partial one
    const bool tail = 2 * lane < rem;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(s.Mat), 0, 0x7fffffff, 0x00020000);
    const int voff = (int)(((int64_t)row * s.ld + kb + 2 * lane) * 8);
    // (the four rows' results leave in ONE store instruction, 32 contiguous bytes from lanes 0 - 3 of wave 0: a 128-byte line of
    //  the panel then sees 4 write-through stores per product instead of 16)
    const SweepVec<true> Bv(s.Bin, s), Ov(s.Out, s);
    const int trow = vb * 4 + (int)threadIdx.x;
    double bin = 0.0;
    if (s.sub && threadIdx.x < 4 && trow < s.rows) bin = Bv.ld(trow);
    sw_v4u mv[NM];
#pragma unroll
    for (int u = 0; u < NM; ++u) {
        if (s.dbg & 1) mv[u] = (sw_v4u){0u, 0x3ff00000u, 0u, 0x3ff00000u};       // (tuning aid: no matrix loads, ones instead)
        else if (u < nfull) mv[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, voff, u * 1024, 0);
        else if (u == nfull && tail) mv[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, voff, u * 1024, 0);
    }
    gather();
    double sum0 = 0.0, sum1 = 0.0;
    sp_lds_double* xl = xs + kb + 2 * lane;
#pragma unroll
    for (int u = 0; u < NM; ++u) {
        if (u < nfull || (u == nfull && tail)) {
            const v2d xv = *reinterpret_cast<__attribute__((address_space(3))) const v2d*>(xl + 128 * u);
            sum0 = fma(__hiloint2double((int)mv[u].y, (int)mv[u].x), xv.x, sum0);
            sum1 = fma(__hiloint2double((int)mv[u].w, (int)mv[u].z), xv.y, sum1);
        }
    }
    if ((ke & 1) && lane == 0) sum0 = fma(s.Mat[(int64_t)row * s.ld + ke - 1], xs[ke - 1], sum0);
    double sum = sum0 + sum1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) res[wv] = sum;
    __syncthreads();
    if (threadIdx.x < 4 && trow < s.rows) Ov.st(trow, s.sub ? bin - res[threadIdx.x] : res[threadIdx.x]);
}

// backward: out[c] = sum_k Mat[k][c] x[k] for the CW columns of block `cb` -- CW / 2 threads x 16 bytes per matrix row, 512 / CW row
// groups; per thread the rows k = kb + group + RG u, ascending u; the fixed-order LDS reduction over the row groups.  (The
// launch-per-product kernel pairs block j with block ncb - 1 - j in one workgroup to balance a LAUNCH; here every block is a
// virtual block of its own -- the chip has the workgroups -- which changes nothing in a block's sums.)
template <bool TRI, int CW, class Gather>
__device__ __forceinline__ void sweep_gemv_t_flow(const SweepArgs& s, int cb, sp_lds_double* xs, double (*red)[CW + 1], Gather&& gather) {
    constexpr int TPR = CW / 2, RG = 256 / TPR;
    constexpr int NM = SWEEP_PERSIST_XMAX / RG;                 // rows per thread: a whole block
    const int t = threadIdx.x;
    const int c2 = (t % TPR) * 2, gidx = t / TPR;
    const int col0 = cb * CW;
    const int kb = TRI ? col0 : 0;
    const int ke = (!TRI && s.mend) ? __builtin_amdgcn_readfirstlane(s.mend[col0 >> 6]) : s.kdim;
    const int span = max(ke - kb, 0), nfull = span / RG, rem = span - nfull * RG;      // whole pieces of RG rows; the partial one
    const bool tail = gidx < rem;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(s.Mat), 0, 0x7fffffff, 0x00020000);
    const int voff = (int)(((int64_t)(kb + gidx) * s.ld + col0 + c2) * 8);
    const int step = (int)(RG * s.ld * 8);
    const SweepVec<true> Bv(s.Bin, s), Ov(s.Out, s);
    double bin = 0.0;
    if (s.sub && t < CW) bin = Bv.ld(col0 + t);        // (before the wait, not behind the reduction)
    sw_v4u mv[NM];
#pragma unroll
    for (int u = 0; u < NM; ++u) {
        if (s.dbg & 1) mv[u] = (sw_v4u){0u, 0x3ff00000u, 0u, 0x3ff00000u};       // (tuning aid: no matrix loads, ones instead)
        else if (u < nfull) mv[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, voff, u * step, 0);
        else if (u == nfull && tail) mv[u] = __builtin_amdgcn_raw_buffer_load_b128(rm, voff, u * step, 0);
    }
    gather();
    double s0 = 0.0, s1 = 0.0;
    sp_lds_double* xl = xs + kb + gidx;
#pragma unroll
    for (int u = 0; u < NM; ++u) {
        if (u < nfull || (u == nfull && tail)) {
            const double xv = xl[u * RG];
            s0 = fma(__hiloint2double((int)mv[u].y, (int)mv[u].x), xv, s0);
            s1 = fma(__hiloint2double((int)mv[u].w, (int)mv[u].z), xv, s1);
        }
    }
    red[gidx][c2] = s0; red[gidx][c2 + 1] = s1;
    __syncthreads();
    if (t < CW) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < RG; ++i) tot += red[i][t];
        Ov.st(col0 + t, s.sub ? bin - tot : tot);
    }
}

template <bool KP1>
__global__ __launch_bounds__(256) void sweep_persist(SweepPersistArgs a) {
    constexpr int CWB = 8;                              // column block of the backward k = 1 products (launch_sweep: `narrow`)
    __shared__ __attribute__((aligned(16))) double xs[KP1 ? SWEEP_PERSIST_XMAX : 2];
    __shared__ double red_t[KP1 ? 512 / CWB : 1][CWB + 1];
    __shared__ double res_n[4];
    __shared__ int stair[2][SWEEP_PERSIST_XMAX / 64];    // kst | mend: looked up in every product, so not from global memory each time
    const int w = blockIdx.x, tid = threadIdx.x, nw = a.nw;
    const int bsp = a.bsp, cm = a.cm, rm = a.rm, wc = bsp - cm;
    if (tid < bsp / 64) { stair[0][tid] = a.kst[tid]; stair[1][tid] = a.mend[tid]; }
    __syncthreads();
    const bool bw = a.backward != 0;
    // product p = 0 .. 2 N - 2:  p = 0 is the first block's X product; then (C product, X product) per block
    auto product = [&](int p, SweepArgs& s) -> int {
        const int step = (p + 1) >> 1, part = (p + 1) & 1;
        const int i = bw ? (a.N - 1 - step) : step;
        s.pMat = s.pXin = s.pBin = s.pOut = 0;
        s.ldx = s.ldb = s.ldo = a.npad;
        s.abort_w = a.abort_w; s.host_abort = a.host_abort; s.spin_limit = a.spin_limit; s.dbg = a.dbg; s.pause = a.pause;
        if (part == 0) {
            // forward: T_i = P_i - C_{i-1} y_{i-1};  backward: T_i = P_i - C_i^T x_{i+1}
            const int ci = bw ? i : (i - 1), prev = bw ? (i + 1) : (i - 1);
            s.Mat = a.C + (int64_t)ci * a.cstride; s.ld = a.ldc;
            s.Xin = a.Yout + (int64_t)prev * bsp + (bw ? 0 : cm);
            s.Bin = a.Pin + (int64_t)i * bsp + (bw ? cm : 0);
            s.Out = a.T + (int64_t)i * bsp + (bw ? cm : 0);
            s.rows = bw ? wc : rm; s.kdim = bw ? rm : wc; s.sub = 1;
            s.kst = stair[0]; s.mend = stair[1];
        } else {
            s.Mat = a.Linv + (int64_t)i * a.bstride; s.ld = bsp;
            s.Xin = (step == 0 ? a.Pin : a.T) + (int64_t)i * bsp;        // (the first block has no coupling product)
            s.Bin = nullptr; s.ldb = 0; s.Out = a.Yout + (int64_t)i * bsp;
            s.rows = bsp; s.kdim = bsp; s.sub = 0;
            s.kst = nullptr; s.mend = nullptr;
        }
        return part;
    };
    const int np = 2 * a.N - 1;
    for (int p = 0; p < np; ++p) {
        SweepArgs s;
        const int part = product(p, s);
        if (KP1) {
            // One right-hand side: every wave of the chip reads (nearly) the whole input vector.  The workgroup gathers the range
            // its virtual block reads ONCE -- thread t the 16-byte pieces t, t + 256, ..., each repeated until it holds no
            // sentinel -- into LDS, behind the requests for its piece of the matrix (sweep_gemv_*_flow).
            sp_lds_double* xl = (sp_lds_double*)xs;
            const int ncb = s.rows / CWB;
            const int nvb = !bw ? (s.rows + 3) / 4 : ncb;
            for (int vb = w; vb < nvb; vb += nw) {
                // backward: the two column blocks that share the 128-byte lines of the matrix rows go to workgroups 8 apart
                // (one XCD under round-robin placement: speed only) -- w = x + 8 q  ->  block 2 (8 (q >> 1) + x) + (q & 1)
                int cb = vb;
                if (bw && ncb % 16 == 0) { const int q = vb >> 3, x = vb & 7; cb = 2 * (8 * (q >> 1) + x) + (q & 1); }
                int xlo, xhi;                      // the range of the input the block reads (even bounds)
                if (!bw) {
                    const int row0 = vb * 4;
                    xlo = (part == 0 && s.kst) ? s.kst[row0 >> 6] & ~1 : 0;
                    xhi = part == 0 ? s.kdim : min(row0 + 4, s.kdim);
                } else {
                    const int col0 = cb * CWB;
                    xlo = part == 0 ? 0 : col0 & ~1;
                    xhi = (part == 0 && s.mend) ? s.mend[col0 >> 6] : s.kdim;
                }
                xhi = min((xhi + 1) & ~1, (s.kdim + 1) & ~1);
                auto gather = [&]() {
                    const SweepVec<true> X(s.Xin, s);
                    // (piece by piece: looking at a thread's pieces together, or a pause before the first look, measured slower --
                    //  0.59 -> 0.60 ms and +0.28 us per product per 0.43 us of pause on darcy256's forward sweep)
                    for (int k = xlo + 2 * tid; k < xhi; k += 512) {
                        v2d v[1];
                        X.ldw2(v, k, 0);
                        *reinterpret_cast<v2d*>(xs + k) = v[0];
                    }
                    __syncthreads();
                };
                if (!bw) { if (part == 0) sweep_gemv_n_flow<false>(s, vb, xl, res_n, gather); else sweep_gemv_n_flow<true>(s, vb, xl, res_n, gather); }
                else { if (part == 0) sweep_gemv_t_flow<false, CWB>(s, cb, xl, red_t, gather); else sweep_gemv_t_flow<true, CWB>(s, cb, xl, red_t, gather); }
                __syncthreads();                   // (the next block's vector / partial sums go to the same LDS words)
            }
        } else {
            const int gx = s.rows / 16, nvb = gx * (a.kp / 16);
            for (int vb = w; vb < nvb; vb += nw) {
                const int bx = vb % gx, by = vb / gx;
                if (!bw) { if (part == 0) sweep_mm_body<false, false, true>(s, bx, by); else sweep_mm_body<false, true, true>(s, bx, by); }
                else { if (part == 0) sweep_mm_body<true, false, true>(s, bx, by); else sweep_mm_body<true, true, true>(s, bx, by); }
                __syncthreads();               // (`red` is rewritten by the next block)
            }
        }
        if (part == 0) {
            // the rows of T_i outside the coupling window (forward: rows >= rm; backward: rows < cm) are those of P_i
            const int u0 = bw ? 0 : rm, un = bw ? cm : bsp - rm;
            if (un > 0) {
                const int step = (p + 1) >> 1, i = bw ? (a.N - 1 - step) : step;
                const SweepVec<true> src(a.Pin + (int64_t)i * bsp + u0, s), dst(a.T + (int64_t)i * bsp + u0, s);
                for (int e = w * 256 + tid; e < un * a.kp; e += nw * 256) {
                    const int r = e / un, c = e - r * un;
                    dst.st((int64_t)r * a.npad + c, src.ld((int64_t)r * a.npad + c));
                }
            }
        }
    }
}

}  // namespace gmrf
