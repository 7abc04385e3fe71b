// latency of the per-column dependent chain of the tile factorisation, piece by piece
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <int V>
__global__ void k(unsigned long long* out, double* sink, int iters) {
    double a = 2.0 + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double p = a;
        if (V & 1) p = bcast_lane(a, it & 63);                 // readlane of the pivot
        double y = p;
        if (V & 2) y = __builtin_amdgcn_rsq(p);                // transcendental seed
        if (V & 4) { const double e = fma(-p * y, y, 1.0); const double c = fma(0.375, e, 0.5); y = fma(y * e, c, y); }  // Halley (4 dep ops)
        double l = b * y;                                       // multiplier
        if (V & 8) { const double lc = bcast_lane(l, (it + 1) & 63); a = fma(-l, lc, a + 3.0); }   // urgent update
        else a = fma(-l, 0.5, a + 3.0);
        b = l + 1.0;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (a == 12345.678) sink[0] = a + b;
    if (threadIdx.x == 0) out[0] = c1 - c0;
}
int main() {
    unsigned long long* d_t; double* d_s; hipMalloc(&d_t, 64); hipMalloc(&d_s, 64);
    unsigned long long ht; const int it = 20000;
#define RUN(V, name) { hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, d_t, d_s, it); hipDeviceSynchronize(); hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, d_t, d_s, it); hipDeviceSynchronize(); \
    hipMemcpy(&ht, d_t, 8, hipMemcpyDeviceToHost); printf("%-50s %.1f cycles/iter\n", name, (double)ht / it); }
    RUN(0, "mul + fma + add only");
    RUN(1, "+ readlane pivot");
    RUN(2, "+ v_rsq_f64");
    RUN(6, "+ v_rsq_f64 + Halley");
    RUN(7, "+ readlane + rsq + Halley");
    RUN(15, "full chain (2 readlanes, rsq, Halley, mul, fma)");
    return 0;
}
