// Block-bidiagonal sweep kernels: one block step of
//   forward_solve  (/root/reference/src/tridiagonal_cholesky.jl:43-52)   y_i = L_i^-1 (b_i - C_{i-1} y_{i-1})
//   backward_solve (/root/reference/src/tridiagonal_cholesky.jl:24-33)   x_i = L_i^-T (y_i - C_i^T x_{i+1})
// on the device-resident factor (dense row-major C_i and explicit lower-triangular
// inverse Linv_i, so that a triangular solve is a matrix product and a block step is two
// dependent launches instead of bs/64 of them).
//
// Right-hand sides live in a panel P[rhs][n_pad] (each right-hand side contiguous, the
// column-major n x k layout of a Julia Matrix).  Two kernel families:
//   * sweep_mm   : kp = multiple of 16 right-hand sides, v_mfma_f64_16x16x4_f64; one
//                  16(rhs) x 16(rows) output tile per workgroup, its 4 waves split K, fixed
//                  order LDS reduction (deterministic);
//   * sweep_gemv : one right-hand side (posterior mean), pure HBM streaming of the block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_f64.hpp"

namespace gmrf {

struct SweepArgs {
    const double* Mat; int64_t ld;      // row-major block (C_i, Linv_i) or a sub-block of it
    const double* Xin; int64_t ldx;     // input vectors, one per rhs, stride ldx between rhs
    const double* Bin; int64_t ldb;     // optional addend (Out = Bin - Mat*X when sub != 0)
    double* Out; int64_t ldo;           // Out may be Bin (in place): an output reads only its own addend
    int rows;                           // outputs per right-hand side
    int kdim;                           // length of the sums (TRI: rows == kdim)
    int sub;
    int64_t pMat, pXin, pBin, pOut;     // per-problem strides (blockIdx.z)
    // staircase of a coupling block (non-triangular kernels only; nullptr: dense).  Per 64-wide tile:
    // TRANS = false: row tile t sums over k >= kst[t];  TRANS = true: output tile u sums over k < mend[u].
    const int* kst = nullptr;
    const int* mend = nullptr;
    // sweep_persist.hpp only (the bodies' COH form): words a bounded wait for the input panel gives up on
    unsigned* abort_w = nullptr;
    unsigned* host_abort = nullptr;
    unsigned spin_limit = 0;
    int narrow = 0;                     // k = 1, transposed: 8-column blocks (one problem or a few, blocks of 512 .. 1024: launch_sweep)
    int pause = 8;                      // clocks / 64 between two looks at an input that is not there yet (multiples of 8; sweep_persist sets it)
    int dbg = 0;                        // tuning aid (GMRF_SWEEP_DBG): 1 = the k = 1 flow bodies skip their matrix loads (garbage results: the pure hand-off time)
};

__device__ __forceinline__ void sweep_select_problem(SweepArgs& s, int p) {
    s.Mat += (int64_t)p * s.pMat;
    s.Xin += (int64_t)p * s.pXin;
    if (s.Bin) s.Bin += (int64_t)p * s.pBin;
    s.Out += (int64_t)p * s.pOut;
}


typedef unsigned sw_v4u __attribute__((ext_vector_type(4)));
typedef unsigned sw_v2u __attribute__((ext_vector_type(2)));

// The right-hand panel as a sweep body reads and writes it.  COH = false (one launch per product): plain loads and stores.
// COH = true (sweep_persist.hpp: the products of a whole sweep inside ONE launch, workgroups on different XCDs handing the
// panel on to each other as a DATA FLOW, no flags): every store is an 8-byte `sc1` write-through store, every load an `sc1`
// buffer load (L1 bypassed; the per-XCD L2s are not coherent), and the INPUT vectors are read by the `ldw` forms, which
// repeat a chunk's loads until none of its values is the sentinel the panel was filled with before the launch (an element is
// written once per launch, by one store instruction: a value that is not the sentinel is final -- the guide's data-tagged
// granule, MI355X_MICROARCH.md "allgather", with the fp64 value as its own tag).  Every repetition is bounded: after
// `spin_limit` ticks of the 100 MHz clock (or when another wave has given up) the abort words are set and the body goes on
// with what it has -- the launch always drains, the host repeats the solve.  Offsets are 32-bit byte offsets from the base.
constexpr unsigned SWEEP_SENT_HI = 0x7ffbadc0u, SWEEP_SENT_LO = 0xdec0de5au;      // a quiet NaN no arithmetic produces

template <bool COH>
struct SweepVec {
    const double* p;
    __amdgpu_buffer_rsrc_t rs;
    const SweepArgs* sa;
    __device__ __forceinline__ SweepVec(const double* q, const SweepArgs& s) : p(q), sa(&s) {
        if (COH) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(q), 0, 0x7fffffff, 0x00020000);
    }
    __device__ __forceinline__ double ld(int64_t i) const {
        if (COH) {
            const sw_v2u u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(i * 8), 0, 16);
            return __hiloint2double((int)u.y, (int)u.x);
        }
        return p[i];
    }
    __device__ __forceinline__ v2d ld2(int64_t i) const {
        if (COH) {
            const sw_v4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 8), 0, 16);
            return (v2d){__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z)};
        }
        return *reinterpret_cast<const v2d*>(p + i);
    }
    __device__ __forceinline__ void st(int64_t i, double v) const {
        if (COH) {
            sw_v2u u;
            u.x = (unsigned)__double2loint(v); u.y = (unsigned)__double2hiint(v);
            __builtin_amdgcn_raw_buffer_store_b64(u, rs, (int)(i * 8), 0, 16);
        } else {
            const_cast<double*>(p)[i] = v;
        }
    }
    static __device__ __forceinline__ bool sent(double v) { return (unsigned)__double2hiint(v) == SWEEP_SENT_HI; }
    // one more look has failed: false when the wait is over for good (abort).  The clock and the abort word are looked at every
    // 64th time only (either is a trip of its own through the scalar cache / the fabric, and nearly every wait fails once or twice)
    __device__ __forceinline__ bool again(unsigned& n, unsigned long long& t0) const {
        for (int z = sa->pause; z > 0; z -= 8) __builtin_amdgcn_s_sleep(8);        // (s_sleep counts 64 clocks per unit, 8 units at a time)
        if ((++n & 63u) == 0u || sa->spin_limit == 0u) {           // (limit 0, tests: the first look that fails gives up)
            if (__hip_atomic_load(sa->abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (n <= 64u) t0 = now;
            if (now - t0 >= (unsigned long long)sa->spin_limit) {
                __hip_atomic_store(sa->abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sa->host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
        return true;
    }
    // NV 16-byte pieces of the input, `stride` elements apart: v[u] = in[i0 + u stride .. +1]
    template <int NV>
    __device__ __forceinline__ void ldw2(v2d (&v)[NV], int64_t i0, int64_t stride) const {
        unsigned n = 0; unsigned long long t0 = 0;
#pragma clang loop unroll(disable)
        for (;;) {
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u] = ld2(i0 + u * stride);
            if (!COH) return;
            bool bad = false;
#pragma unroll
            for (int u = 0; u < NV; ++u) bad = bad || sent(v[u].x) || sent(v[u].y);
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
            if (!again(n, t0)) return;
        }
    }
    // NV single elements of the input, `stride` apart
    template <int NV>
    __device__ __forceinline__ void ldw1(double (&v)[NV], int64_t i0, int64_t stride) const {
        unsigned n = 0; unsigned long long t0 = 0;
#pragma clang loop unroll(disable)
        for (;;) {
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u] = ld(i0 + u * stride);
            if (!COH) return;
            bool bad = false;
#pragma unroll
            for (int u = 0; u < NV; ++u) bad = bad || sent(v[u]);
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;
            if (!again(n, t0)) return;
        }
    }
};

// TRANS = false: out[m] = sum_k Mat[m][k] x[k]   (TRI: Mat lower triangular, k <= m)
// TRANS = true : out[m] = sum_k Mat[k][m] x[k]   (TRI: Mat lower triangular, k >= m)
template <bool TRANS, bool TRI, bool COH>
__device__ __forceinline__ void sweep_mm_body(const SweepArgs& s, int bx, int by) {
    const int m0 = bx * 16, r0 = by * 16;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    int kb = 0, ke = s.kdim;
    if (TRI) {
        if (!TRANS) ke = m0 + 16; else kb = m0;
    } else {
        if (!TRANS && s.kst) kb = s.kst[m0 >> 6];
        if (TRANS && s.mend) ke = s.mend[m0 >> 6];
    }
    // split [kb, ke) over the 4 waves in multiples of 8
    const int total = ke - kb;
    const int chunk = ((total / 4 + 7) / 8) * 8;
    const int k_lo = kb + w * chunk;
    const int k_hi = min(ke, k_lo + chunk);

    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const SweepVec<COH> X(s.Xin, s), Bv(s.Bin, s), Ov(s.Out, s);
    const int64_t xrow = (int64_t)(r0 + li) * s.ldx;
    // (sweep_persist: the addend is requested before the wait for the input -- behind the reduction it is a round trip on the critical path)
    const int rhs_t = ((t & 63) >> 4) + 4 * (t >> 6), m_t = t & 15;
    double bin_early = 0.0;
    if (COH && s.sub) bin_early = Bv.ld((int64_t)(r0 + rhs_t) * s.ldb + m0 + m_t);
    // Operands come straight from global memory (each wave has its own K range, nothing to share
    // through LDS).  Loads are issued a whole chunk of k-groups ahead of the MFMAs that consume
    // them, otherwise every pair of MFMAs waits a full memory latency: chunks of 8 groups, then single groups -- and inside
    // sweep_persist (COH), where a lone workgroup per CU has the registers for it and nothing else hides the latency, chunks
    // of 16 first and of 4 and 2 behind the 8s.  The MFMAs run in ascending k whatever the chunking: same sums, bit for bit.
    int k = k_lo;
    if (!TRANS) {
        const double* __restrict__ mrow = s.Mat + (int64_t)(m0 + li) * s.ld;
        auto run = [&](auto chc) {
            constexpr int CH = decltype(chc)::value;
            for (; k + 8 * CH <= k_hi; k += 8 * CH) {
                v2d a[CH], b[CH];
                if (COH) {                              // (the block's piece first: it does not wait for anybody)
#pragma unroll
                    for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const v2d*>(mrow + k + 8 * u + 2 * lq);
                    X.ldw2(a, xrow + k + 2 * lq, 8);
                } else {
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        a[u] = X.ld2(xrow + k + 8 * u + 2 * lq);
                        b[u] = *reinterpret_cast<const v2d*>(mrow + k + 8 * u + 2 * lq);
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b[u].y, acc, 0, 0, 0);
                }
            }
        };
        if (COH) run(std::integral_constant<int, 16>{});
        run(std::integral_constant<int, 8>{});
        if (COH) { run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 2>{}); }
        run(std::integral_constant<int, 1>{});
    } else {
        const double* __restrict__ mcol = s.Mat + m0 + li;
        auto run = [&](auto chc) {
            constexpr int CH = decltype(chc)::value;
            for (; k + 8 * CH <= k_hi; k += 8 * CH) {
                v2d a[CH];
                double b0[CH], b1[CH];
                if (COH) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        b0[u] = mcol[(int64_t)(k + 8 * u + 2 * lq) * s.ld];
                        b1[u] = mcol[(int64_t)(k + 8 * u + 2 * lq + 1) * s.ld];
                    }
                    X.ldw2(a, xrow + k + 2 * lq, 8);
                } else {
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        a[u] = X.ld2(xrow + k + 8 * u + 2 * lq);
                        b0[u] = mcol[(int64_t)(k + 8 * u + 2 * lq) * s.ld];
                        b1[u] = mcol[(int64_t)(k + 8 * u + 2 * lq + 1) * s.ld];
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b0[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b1[u], acc, 0, 0, 0);
                }
            }
        };
        if (COH) run(std::integral_constant<int, 16>{});
        run(std::integral_constant<int, 8>{});
        if (COH) { run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 2>{}); }
        run(std::integral_constant<int, 1>{});
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][r * 64 + lane] = acc[r];
    __syncthreads();
    // element t of the 16x16 tile: reg = t >> 6, lane' = t & 63 -> rhs = (lane' >> 4) + 4 reg, m = lane' & 15
    const double sum = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    const int rhs = ((t & 63) >> 4) + 4 * (t >> 6);
    const int m = t & 15;
    double v = sum;
    if (s.sub) v = (COH ? bin_early : Bv.ld((int64_t)(r0 + rhs) * s.ldb + m0 + m)) - sum;
    Ov.st((int64_t)(r0 + rhs) * s.ldo + m0 + m, v);
}
template <bool TRANS, bool TRI>
__global__ __launch_bounds__(256, 2) void sweep_mm(SweepArgs s) {
    sweep_select_problem(s, blockIdx.z);
    sweep_mm_body<TRANS, TRI, false>(s, blockIdx.x, blockIdx.y);
}

// One right-hand side, non-transposed: one wave per row, 16-byte loads along the row.
template <bool TRI>
__global__ __launch_bounds__(256) void sweep_gemv_n(SweepArgs s) {
    sweep_select_problem(s, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= s.rows) return;
    const int ke = TRI ? (row + 1) : s.kdim;
    const int kb = (!TRI && s.kst) ? s.kst[row >> 6] : 0;      // staircase: the row is zero left of kb
    const double* __restrict__ mrow = s.Mat + (int64_t)row * s.ld;
    const double* __restrict__ x = s.Xin;
    double sum0 = 0.0, sum1 = 0.0;
    const int ke2 = ke & ~1;
    // all loads of a chunk of 8 strides are issued before the first fma consumes one
    int k = kb + lane * 2;
    for (; k + 7 * 128 < ke2; k += 8 * 128) {
        v2d mv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            mv[u] = *reinterpret_cast<const v2d*>(mrow + k + 128 * u);
            xv[u] = *reinterpret_cast<const v2d*>(x + k + 128 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sum0 = fma(mv[u].x, xv[u].x, sum0);
            sum1 = fma(mv[u].y, xv[u].y, sum1);
        }
    }
    // rows of 512 .. 1023 elements (a 768-wide coupling window, the 768-row part of a split block inverse): four loads in flight
    for (; k + 3 * 128 < ke2; k += 4 * 128) {
        v2d mv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mv[u] = *reinterpret_cast<const v2d*>(mrow + k + 128 * u);
            xv[u] = *reinterpret_cast<const v2d*>(x + k + 128 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sum0 = fma(mv[u].x, xv[u].x, sum0);
            sum1 = fma(mv[u].y, xv[u].y, sum1);
        }
    }
    for (; k < ke2; k += 128) {
        const v2d mv = *reinterpret_cast<const v2d*>(mrow + k);
        const v2d xv = *reinterpret_cast<const v2d*>(x + k);
        sum0 = fma(mv.x, xv.x, sum0);
        sum1 = fma(mv.y, xv.y, sum1);
    }
    if ((ke & 1) && lane == 0) sum0 = fma(mrow[ke - 1], x[ke - 1], sum0);
    double sum = sum0 + sum1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) {
        double v = sum;
        if (s.sub) v = s.Bin[row] - sum;
        s.Out[row] = v;
    }
}

// One right-hand side, non-transposed, SHORT rows (kdim <= 256: the first block column of a split block inverse, see
// gmrf_handle::xsplit): a wave owns four consecutive rows, the vector piece is loaded once, the eight 16-byte row pieces
// are in flight together (one row per wave leaves a lane with two loads and a reduction: latency, not bandwidth).
template <bool TRI>
__global__ __launch_bounds__(256) void sweep_gemv_n_short(SweepArgs s) {
    sweep_select_problem(s, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= s.rows) return;
    const double* __restrict__ x = s.Xin;
    const int k0 = lane * 2, k1 = lane * 2 + 128;
    const int kd = s.kdim;
    const v2d z = (v2d){0.0, 0.0};
    // (TRI: row r sums k <= r; the stored zeros right of the diagonal make whole 16-byte pieces safe to read up to the
    //  next even k, and pieces entirely right of it are skipped)
    v2d m[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ke = TRI ? (row0 + r + 2) & ~1 : kd;
        const double* mrow = s.Mat + (int64_t)(row0 + r) * s.ld;
        m[r][0] = (k0 < ke) ? *reinterpret_cast<const v2d*>(mrow + k0) : z;
        m[r][1] = (k1 < ke) ? *reinterpret_cast<const v2d*>(mrow + k1) : z;
    }
    const int kx = TRI ? (row0 + 5) & ~1 : kd;
    const v2d x0 = (k0 < kx && k0 < kd) ? *reinterpret_cast<const v2d*>(x + k0) : z;
    const v2d x1 = (k1 < kx && k1 < kd) ? *reinterpret_cast<const v2d*>(x + k1) : z;
    double sum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // the order of sweep_gemv_n: even elements into one sum, odd into the other, ascending k, then the wave reduction
        double a0 = fma(m[r][0].x, x0.x, 0.0), a1 = fma(m[r][0].y, x0.y, 0.0);
        a0 = fma(m[r][1].x, x1.x, a0); a1 = fma(m[r][1].y, x1.y, a1);
        sum[r] = a0 + a1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], off, 64);
    }
    if (lane < 4) {
        const int row = row0 + lane;
        const double v0 = lane == 0 ? sum[0] : (lane == 1 ? sum[1] : (lane == 2 ? sum[2] : sum[3]));
        s.Out[row] = s.sub ? s.Bin[row] - v0 : v0;
    }
}

// One right-hand side, non-transposed, rows of moderate length (a batch's coupling window inside its staircase, the parts of
// a split block inverse: 256 .. 768 elements): a wave owns FOUR consecutive rows (rows % 4 == 0; they share the staircase
// start, 4 | 64), reads the vector piece once per 128 elements and keeps eight 16-byte row pieces in flight -- one row per
// wave (sweep_gemv_n) gets there only for rows of 1024 elements and more.  Per row the summation order of sweep_gemv_n.
template <bool TRI>
__global__ __launch_bounds__(256) void sweep_gemv_n4(SweepArgs s) {
    sweep_select_problem(s, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= s.rows) return;
    const double* __restrict__ x = s.Xin;
    const int kb = (!TRI && s.kst) ? s.kst[row0 >> 6] : 0;
    // TRI: row r ends at r + 1; pieces are read up to the next even index (stored zeros right of the diagonal)
    const int ke_max = TRI ? (row0 + 5) & ~1 : s.kdim & ~1;
    const double* __restrict__ m0 = s.Mat + (int64_t)row0 * s.ld;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    const v2d z = (v2d){0.0, 0.0};
    int k = kb + lane * 2;
    for (; k + 128 < ke_max; k += 256) {
        v2d mv[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ke = TRI ? (row0 + r + 2) & ~1 : ke_max;
            mv[r][0] = (k < ke) ? *reinterpret_cast<const v2d*>(m0 + (int64_t)r * s.ld + k) : z;
            mv[r][1] = (k + 128 < ke) ? *reinterpret_cast<const v2d*>(m0 + (int64_t)r * s.ld + k + 128) : z;
        }
        const v2d x0 = *reinterpret_cast<const v2d*>(x + k), x1 = *reinterpret_cast<const v2d*>(x + k + 128);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0[r] = fma(mv[r][0].x, x0.x, a0[r]); a1[r] = fma(mv[r][0].y, x0.y, a1[r]);
            a0[r] = fma(mv[r][1].x, x1.x, a0[r]); a1[r] = fma(mv[r][1].y, x1.y, a1[r]);
        }
    }
    for (; k < ke_max; k += 128) {
        const v2d x0 = *reinterpret_cast<const v2d*>(x + k);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ke = TRI ? (row0 + r + 2) & ~1 : ke_max;
            const v2d mv = (k < ke) ? *reinterpret_cast<const v2d*>(m0 + (int64_t)r * s.ld + k) : z;
            a0[r] = fma(mv.x, x0.x, a0[r]); a1[r] = fma(mv.y, x0.y, a1[r]);
        }
    }
    double sum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] = a0[r] + a1[r];
    if (!TRI && (s.kdim & 1) && lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = fma(m0[(int64_t)r * s.ld + s.kdim - 1], x[s.kdim - 1], sum[r]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], off, 64);
    }
    if (lane < 4) {
        const int row = row0 + lane;
        const double v0 = lane == 0 ? sum[0] : (lane == 1 ? sum[1] : (lane == 2 ? sum[2] : sum[3]));
        s.Out[row] = s.sub ? s.Bin[row] - v0 : v0;
    }
}

// One right-hand side, transposed: out[c] = sum_k Mat[k][c] x[k].  A workgroup owns CW adjacent output
// columns (CW / 2 threads x 16 bytes per matrix row, 512 / CW row groups); fixed-order LDS reduction
// over the row groups.  TRI (Linv^T, rows >= column): column block j is paired with block ncb - 1 - j in
// the same workgroup so that every workgroup streams the same number of rows -- with one block each
// the first workgroup reads 64 times what the last one does, and once the short ones have left, the
// long ones run on a nearly empty chip (3.0 TB/s on darcy256 / batch 32 against 4.6 for the other
// sweep kernels).  Non-TRI (C^T inside its staircase): rows [0, mend[column tile]).
template <bool TRI, int CW>
__global__ __launch_bounds__(256) void sweep_gemv_t(SweepArgs s) {
    sweep_select_problem(s, blockIdx.z);
    constexpr int TPR = CW / 2, RG = 256 / TPR;
    const int t = threadIdx.x;
    const int c2 = (t % TPR) * 2, gidx = t / TPR;
    const double* __restrict__ x = s.Xin;
    __shared__ double red[RG][CW + 1];
    const int ncb = s.rows / CW;
    const int npass = TRI ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int cb = TRI ? (pass == 0 ? (int)blockIdx.x : ncb - 1 - (int)blockIdx.x) : (int)blockIdx.x;
        if (TRI && pass == 1 && cb == (int)blockIdx.x) break;          // odd block count: the middle block once
        const int col0 = cb * CW;
        const int kb = TRI ? col0 : 0;          // lower triangular: rows >= column (zeros above the diagonal are stored)
        const int ke = (!TRI && s.mend) ? s.mend[col0 >> 6] : s.kdim;
        const double* __restrict__ mp = s.Mat + col0 + c2;
        double s0 = 0.0, s1 = 0.0;
        int k = kb + gidx;
        for (; k + 7 * RG < ke; k += 8 * RG) {                          // eight independent 16-byte loads in flight
            v2d mv[8];
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                mv[u] = *reinterpret_cast<const v2d*>(mp + (int64_t)(k + u * RG) * s.ld);
                xv[u] = x[k + u * RG];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 = fma(mv[u].x, xv[u], s0); s1 = fma(mv[u].y, xv[u], s1); }
        }
        for (; k < ke; k += RG) {
            const v2d mv = *reinterpret_cast<const v2d*>(mp + (int64_t)k * s.ld);
            const double xv = x[k];
            s0 = fma(mv.x, xv, s0); s1 = fma(mv.y, xv, s1);
        }
        if (pass == 1) __syncthreads();                                  // the first block's reduction has read `red`
        red[gidx][c2] = s0; red[gidx][c2 + 1] = s1;
        __syncthreads();
        if (t < CW) {
            double tot = 0.0;
#pragma unroll
            for (int i = 0; i < RG; ++i) tot += red[i][t];
            double v = tot;
            if (s.sub) v = s.Bin[col0 + t] - tot;
            s.Out[col0 + t] = v;
        }
    }
}

inline hipError_t launch_sweep(hipStream_t st, bool trans, bool tri, int kp, const SweepArgs& s, int nprob) {
    if (kp == 1) {
        if (!trans) {
            if (s.kdim <= 256 && s.rows % 16 == 0 && s.kdim % 2 == 0 && !s.kst) {
                dim3 grid(s.rows / 16, 1, nprob), block(256);
                if (tri) hipLaunchKernelGGL((sweep_gemv_n_short<true>), grid, block, 0, st, s);
                else hipLaunchKernelGGL((sweep_gemv_n_short<false>), grid, block, 0, st, s);
                return hipGetLastError();
            }
            if (nprob >= 8 && s.rows % 16 == 0 && s.kdim <= 768 && s.kdim % 2 == 0) {
                dim3 grid(s.rows / 16, 1, nprob), block(256);
                if (tri) hipLaunchKernelGGL((sweep_gemv_n4<true>), grid, block, 0, st, s);
                else hipLaunchKernelGGL((sweep_gemv_n4<false>), grid, block, 0, st, s);
                return hipGetLastError();
            }
            dim3 grid((s.rows + 3) / 4, 1, nprob), block(256);
            if (tri) hipLaunchKernelGGL((sweep_gemv_n<true>), grid, block, 0, st, s);
            else hipLaunchKernelGGL((sweep_gemv_n<false>), grid, block, 0, st, s);
        } else {
            // wide column blocks (256-byte row pieces) when the batch supplies enough workgroups that way (two per CU),
            // narrow ones otherwise (the short products of a split block inverse); one problem (or a few) with blocks of 512 and more:
            // 8 columns (64-byte row pieces, 64 row groups) -- its chain of dependent products is bound by what ONE CU pulls in,
            // and twice the workgroups pull twice as much (round 5; sweep_persist's k = 1 bodies do the same sums)
            const int ncb32 = s.rows / 32;
            const bool wide = nprob >= 8 && s.rows % 32 == 0 && (int64_t)(tri ? (ncb32 + 1) / 2 : ncb32) * nprob >= 512;
            // (a few problems: the same sums as one alone.  Up to 1024 rows: darcy256's backward sweep 1.10 -> 0.91 ms; blocks of 4096
            //  have the workgroups anyway and lose on the 64-byte row pieces -- burgers4096x512: 26 -> 46 us per coupling product)
            const bool narrow = s.narrow != 0 && s.rows % 8 == 0;      // (the host decides by the BLOCK size: sweep_launches)
            const int cw = wide ? 32 : (narrow ? 8 : 16), ncb = s.rows / cw;
            dim3 grid(tri ? (ncb + 1) / 2 : ncb, 1, nprob), block(256);
            if (tri && wide) hipLaunchKernelGGL((sweep_gemv_t<true, 32>), grid, block, 0, st, s);
            else if (tri && narrow) hipLaunchKernelGGL((sweep_gemv_t<true, 8>), grid, block, 0, st, s);
            else if (tri) hipLaunchKernelGGL((sweep_gemv_t<true, 16>), grid, block, 0, st, s);
            else if (wide) hipLaunchKernelGGL((sweep_gemv_t<false, 32>), grid, block, 0, st, s);
            else if (narrow) hipLaunchKernelGGL((sweep_gemv_t<false, 8>), grid, block, 0, st, s);
            else hipLaunchKernelGGL((sweep_gemv_t<false, 16>), grid, block, 0, st, s);
        }
    } else {
        dim3 grid(s.rows / 16, kp / 16, nprob), block(256);
        if (!trans && !tri) hipLaunchKernelGGL((sweep_mm<false, false>), grid, block, 0, st, s);
        else if (!trans && tri) hipLaunchKernelGGL((sweep_mm<false, true>), grid, block, 0, st, s);
        else if (trans && !tri) hipLaunchKernelGGL((sweep_mm<true, false>), grid, block, 0, st, s);
        else hipLaunchKernelGGL((sweep_mm<true, true>), grid, block, 0, st, s);
    }
    return hipGetLastError();
}

}  // namespace gmrf
