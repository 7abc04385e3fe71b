#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k16(unsigned long long* out, double* sink, int iters) {
    v4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4){0,0,0,0};
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = c1 - c0;
}
template <int NACC>
__global__ void k4(unsigned long long* out, double* sink, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[j], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = c1 - c0;
}
// VALU fma with many chains
template <int NCH>
__global__ void kv(unsigned long long* out, double* sink, int iters) {
    double a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = i + threadIdx.x;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) a[j] = fma(a[j], x, y);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NCH; ++i) s += a[i];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = c1 - c0;
}
template <typename F> void run(const char* name, F launch, double flops_per_thread_block_iter_total) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s %8.1f TF/s  (%.3f ms)\n", name, flops_per_thread_block_iter_total / (ms * 1e-3) / 1e12, ms);
}
int main() {
    unsigned long long* d_t; double* d_s; hipMalloc(&d_t, 64); hipMalloc(&d_s, 64);
    unsigned long long ht;
    const int it = 20000;
#define MF(NACC, WPS) { char nm[64]; snprintf(nm, 64, "mfma16x16x4 acc=%d waves/SIMD=%d", NACC, WPS); \
    run(nm, [&]{ hipLaunchKernelGGL((k16<NACC>), dim3(256*WPS), dim3(256), 0, 0, d_t, d_s, it); }, 256.0*WPS*4*it*NACC*2048.0); \
    hipMemcpy(&ht, d_t, 8, hipMemcpyDeviceToHost); printf("     cycles/MFMA/wave %.1f\n", (double)ht/(it*NACC)); }
    MF(2,1) MF(4,1) MF(8,1) MF(4,2) MF(4,3) MF(4,4) MF(2,8) MF(1,8)
#define M4(NACC, WPS) { char nm[64]; snprintf(nm, 64, "mfma4x4x4 acc=%d waves/SIMD=%d", NACC, WPS); \
    run(nm, [&]{ hipLaunchKernelGGL((k4<NACC>), dim3(256*WPS), dim3(256), 0, 0, d_t, d_s, it); }, 256.0*WPS*4*it*NACC*512.0); \
    hipMemcpy(&ht, d_t, 8, hipMemcpyDeviceToHost); printf("     cycles/MFMA/wave %.1f\n", (double)ht/(it*NACC)); }
    M4(4,1) M4(8,1) M4(8,2) M4(8,4)
#define VV(NCH, WPS) { char nm[64]; snprintf(nm, 64, "valu fma chains=%d waves/SIMD=%d", NCH, WPS); \
    run(nm, [&]{ hipLaunchKernelGGL((kv<NCH>), dim3(256*WPS), dim3(256), 0, 0, d_t, d_s, it); }, 256.0*WPS*256*it*NCH*2.0); \
    hipMemcpy(&ht, d_t, 8, hipMemcpyDeviceToHost); printf("     cycles/FMA/wave %.2f\n", (double)ht/(it*NCH)); }
    VV(8,1) VV(16,1) VV(16,2) VV(16,4) VV(8,8)
    return 0;
}
