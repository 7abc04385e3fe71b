// What does an instruction issued beside the fp64 MFMA stream cost?  (round 3: decides the GEMM staging design)
// Per loop iteration 8 independent-accumulator MFMAs (512 cycles of one SIMD's matrix pipe) plus NX instructions of
// kind V; workgroups of 256 threads, `wgs` of them per CU (dynamic LDS padding), grid = 256 * wgs.
//   V 0: nothing            1: ds_read_b128 (prefetched fragments)      2: ds_read_b64
//     3: global_load_dwordx4 into VGPRs (L2-resident source)           4: global_load_lds_dwordx4 (no VGPR destination)
//     5: ds_write_b128       6: global_load_dwordx4 + ds_write_b128 (register staging)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mb5.hip -o tools/bin/mb5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4 __attribute__((ext_vector_type(4)));
typedef double v2 __attribute__((ext_vector_type(2)));

template <int V, int NX>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, double* sink, const double* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int i = threadIdx.x; i < 4864; i += 256) lds[i] = src[i % 4096];
    __syncthreads();
    v4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    v2 f[4] = {{src[lane], src[lane + 64]}, {src[lane + 128], src[lane + 192]}, {src[lane + 256], src[lane + 320]}, {src[lane + 384], src[lane + 448]}};
    v2 g[NX > 0 ? NX : 1];
    for (int i = 0; i < (NX > 0 ? NX : 1); ++i) g[i] = (v2){0.0, 0.0};
    const double* gp = src + (size_t)(blockIdx.x % 64) * 8192 + threadIdx.x * 2;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int kk = (it & 3) * 8 + 2 * lq;
        if (V == 1) {
#pragma unroll
            for (int x = 0; x < NX; ++x) g[x] = *reinterpret_cast<const v2*>(lds + ((x & 3) * 16 + li) * 36 + kk + (x >> 2) * 2304);
        } else if (V == 2) {
#pragma unroll
            for (int x = 0; x < NX; ++x) g[x].x = lds[((x & 3) * 16 + li) * 37 + kk + (x >> 2) * 2400];
        } else if (V == 3 || V == 6) {
#pragma unroll
            for (int x = 0; x < NX; ++x) g[x] = *reinterpret_cast<const v2*>(gp + ((it * NX + x) & 7) * 512);
        } else if (V == 4) {
#pragma unroll
            for (int x = 0; x < NX; ++x)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * NX + x) & 7) * 512),
                                                 (__attribute__((address_space(3))) void*)(lds + w * 128 + (x & 3) * 512), 16, 0, 0);
        } else if (V == 5) {
#pragma unroll
            for (int x = 0; x < NX; ++x) *reinterpret_cast<v2*>(lds + threadIdx.x * 2 + (x & 3) * 512) = f[x & 3];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].x, f[2].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].x, f[3].x, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1].x, f[2].x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1].x, f[3].x, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].y, f[2].y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].y, f[3].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1].y, f[2].y, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1].y, f[3].y, acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (V == 1 || V == 2 || V == 3) {
            // keep what was loaded live until after the MFMAs (no VALU work: the wait for it sits behind the MFMA issue)
#pragma unroll
            for (int x = 0; x < NX; ++x) { asm volatile("" ::"v"(g[x].x)); if (V != 2) asm volatile("" ::"v"(g[x].y)); }
        }
        if (V == 6) {
#pragma unroll
            for (int x = 0; x < NX; ++x) *reinterpret_cast<v2*>(lds + threadIdx.x * 2 + (x & 3) * 512) = g[x];
        }
    }
    if (V == 4) __builtin_amdgcn_s_waitcnt(0);
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += lds[threadIdx.x];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = c1 - c0;
}

static unsigned long long* d_t; static double* d_s; static double* d_src;

template <int V, int NX>
void run(const char* name, int wgs) {
    const int it = 3000;
    const size_t lds = (size_t)160 * 1024 / wgs - 1024;         // forces `wgs` workgroups per CU
    hipFuncSetAttribute((const void*)k<V, NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<V, NX>), dim3(256 * wgs), dim3(256), lds, 0, d_t, d_s, d_src, it);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V, NX>), dim3(256 * wgs), dim3(256), lds, 0, d_t, d_s, d_src, it);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long ht; hipMemcpy(&ht, d_t, 8, hipMemcpyDeviceToHost);
    const double tf = 256.0 * wgs * 4 * it * 8 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-44s NX=%d wgs/CU=%d: %7.1f wave-cycles per MFMA, %6.1f TF/s\n", name, NX, wgs, (double)ht / (it * 8.0), tf);
    fflush(stdout);
}

int main() {
    hipMalloc(&d_t, 64); hipMalloc(&d_s, 64); hipMalloc(&d_src, 64 * 8192 * 8 + 65536);
    double* h = (double*)malloc(64 * 8192 * 8 + 65536);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < 64 * 8192 + 8192; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
    hipMemcpy(d_src, h, 64 * 8192 * 8 + 65536, hipMemcpyHostToDevice);
    for (int wgs : {1, 2, 4}) {
        run<0, 0>("MFMA only", wgs);
        run<1, 2>("ds_read_b128", wgs);
        run<1, 4>("ds_read_b128", wgs);
        run<1, 8>("ds_read_b128", wgs);
        run<2, 4>("ds_read_b64", wgs);
        run<2, 8>("ds_read_b64", wgs);
        run<3, 1>("global_load_dwordx4 -> VGPR", wgs);
        run<3, 2>("global_load_dwordx4 -> VGPR", wgs);
        run<4, 1>("global_load_lds_dwordx4", wgs);
        run<4, 2>("global_load_lds_dwordx4", wgs);
        run<5, 1>("ds_write_b128", wgs);
        run<5, 2>("ds_write_b128", wgs);
        run<6, 1>("global_load_dwordx4 + ds_write_b128", wgs);
        run<6, 2>("global_load_dwordx4 + ds_write_b128", wgs);
    }
    return 0;
}
