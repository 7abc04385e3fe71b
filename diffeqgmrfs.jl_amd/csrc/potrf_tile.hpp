// 64x64 diagonal-tile Cholesky + triangular inverse in one wave.
//
// Replaces the unblocked dpotf2 / dtrti2 leaves of LAPACK's dpotrf that Julia's
// `cholesky(Array(...))` runs for /root/reference/src/tridiagonal_cholesky.jl:67,77.
//
// Lane r owns row r of the tile in registers (a[c], c <= r, fully unrolled so every index
// is a compile-time constant).  Column step j: the pivot comes from lane j by v_readlane,
// every lane scales its own entry l = a[j] * rsqrt(p), column j of L goes to LDS and is
// read back as a wave-uniform (broadcast) operand for the rank-1 update of the trailing
// columns.  The inverse X = L^-1 is accumulated in the same sweep, row-oriented:
//   X[r][:] = rinv_r * (e_r - sum_{k<r} L[r][k] X[k][:]),
// lane r keeps acc[c] = -sum_k L[r][k] X[k][c] and row j of X is final (and published
// through LDS) exactly when column j of L is.
// A non-positive or NaN pivot records `blk` in *info (first failure wins) and the tile is
// completed with NaNs rather than faulting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to fp64 accuracy: hardware seed + two Newton steps.
__device__ __forceinline__ double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-p * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-p * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

constexpr int PT = 64;
constexpr int PT_LD = 66;   // LDS row stride of the staged tile (16-byte aligned rows)

struct TileArgs {
    const double* S; int64_t lds;       // input SPD tile (lower triangle read)
    double* L; int64_t ldl;             // output L tile (strict upper written as zero)
    double* X; int64_t ldx;             // output inverse tile (strict upper zero)
    int* info; int blk;                 // failure report
};

__global__ __launch_bounds__(64) void potrf_tile64_inv(TileArgs ta) {
    __shared__ __attribute__((aligned(16))) double Ts[PT * PT_LD];   // staged input, then X rows
    __shared__ __attribute__((aligned(16))) double Lc[PT * PT];      // Lc[j*64 + r] = L[r][j]
    __shared__ double rinvs[PT];
    const int lane = threadIdx.x;

    for (int idx = lane; idx < PT * PT / 2; idx += 64) {
        const int r = idx >> 5, c = (idx & 31) * 2;
        *reinterpret_cast<double2*>(&Ts[r * PT_LD + c]) =
            *reinterpret_cast<const double2*>(&ta.S[(int64_t)r * ta.lds + c]);
    }
    __syncthreads();
    double a[PT], acc[PT];
#pragma unroll
    for (int c = 0; c < PT; ++c) {
        a[c] = Ts[lane * PT_LD + c];
        acc[c] = 0.0;
    }
    __syncthreads();
    bool bad = false;

#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const double p = bcast_lane(a[j], j);
        if (!(p > 0.0)) bad = true;
        const double rinv = rsqrt_nr(p);
        const double l = (lane >= j) ? a[j] * rinv : 0.0;     // lane j: p * rinv = sqrt(p)
        a[j] = l;
        Lc[j * PT + lane] = l;
        // publish the unscaled row j of X: Y[j][c] = acc[c] (c < j), Y[j][j] = 1
        if (lane == j) {
#pragma unroll
            for (int c = 0; c < j; ++c) Ts[j * PT_LD + c] = acc[c];
            Ts[j * PT_LD + j] = 1.0;
            rinvs[j] = rinv;
        }
        __syncthreads();
        // trailing update of the factor: a[c] -= L[r][j] * L[c][j]
#pragma unroll
        for (int c = j + 1; c < PT; ++c) a[c] = fma(-l, Lc[j * PT + c], a[c]);
        // inverse accumulation for rows below j: acc[c] -= L[r][j] * rinv_j * Y[j][c]
        const double ls = (lane > j) ? l * rinv : 0.0;
#pragma unroll
        for (int c = 0; c <= j; ++c) acc[c] = fma(-ls, Ts[j * PT_LD + c], acc[c]);
    }
    __syncthreads();
    if (bad && lane == 0) atomicCAS(ta.info, 0, ta.blk);

    // write-out, coalesced along rows: L[r][c] = Lc[c][r] (c <= r), X[r][c] = rinv_r * Y[r][c]
    for (int idx = lane; idx < PT * PT; idx += 64) {
        const int r = idx >> 6, c = idx & 63;
        const double lv = (c <= r) ? Lc[c * PT + r] : 0.0;
        const double xv = (c <= r) ? rinvs[r] * Ts[r * PT_LD + c] : 0.0;
        ta.L[(int64_t)r * ta.ldl + c] = lv;
        ta.X[(int64_t)r * ta.ldx + c] = xv;
    }
}

}  // namespace gmrf
