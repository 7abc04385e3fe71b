// Micro-benchmarks used to pin the roofline denominators on the box the bench runs on
// (fp64 MFMA / VALU rate, shader clock under light and heavy load, launch latency).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

typedef double mbv4 __attribute__((ext_vector_type(4)));

// out[0] = shader cycles, out[1] = 100 MHz ticks for the loop (block 0, lane 0)
template <int NACC>
__global__ __launch_bounds__(256, 2) void mb_mfma_f64(unsigned long long* out, double* sink, int iters) {
    mbv4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (mbv4){0, 0, 0, 0};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void mb_valu_f64(unsigned long long* out, double* sink, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
        a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// Shader-clock probe (round 4): one wave per block (8 blocks: one per XCD under the observed round-robin placement) stamps
// {s_memtime, s_memrealtime} every ~`sleeps` x 3.4 us for n samples while OTHER streams run the load under test; the clock between
// two samples is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).  Bounded: n samples, then it ends.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int n, int sleeps) {
    for (int i = 0; i < n; ++i) {
        if (threadIdx.x == 0) {
            out[((size_t)blockIdx.x * n + i) * 2] = __builtin_amdgcn_s_memtime();
            out[((size_t)blockIdx.x * n + i) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        for (int s = 0; s < sleeps; ++s) __builtin_amdgcn_s_sleep(127);
    }
}

__global__ void mb_null(int* p) {
    if (p && threadIdx.x == 1000) p[0] = 1;
}

}  // namespace gmrf
