// accuracy of v_mfma_f64_16x16x4_f64 when a large accumulator receives small products
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D, double* Dv, int K) {
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    v4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(lq + 4 * r) * 16 + li];
    for (int ks = 0; ks < K / 4; ++ks) {
        const double a = -A[li * K + 4 * ks + lq];
        const double b = B[li * K + 4 * ks + lq];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(lq + 4 * r) * 16 + li] = acc[r];
    // VALU reference: same order with fma
    for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r, col = li;
        double s = C[row * 16 + col];
        for (int kk = 0; kk < K; ++kk) s = fma(-A[row * K + kk], B[col * K + kk], s);
        Dv[row * 16 + col] = s;
    }
}
int main() {
    for (double scale : {1.0, 1e-3, 1e-6}) {
        const int K = 64;
        std::vector<double> A(16 * K), B(16 * K), C(256), D(256), Dv(256);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5; };
        for (auto& x : A) x = rnd() * scale; for (auto& x : B) x = rnd(); for (auto& x : C) x = rnd() * 10;
        double *dA, *dB, *dC, *dD, *dV; hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048); hipMalloc(&dV, 2048);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, dV, K);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(Dv.data(), dV, 2048, hipMemcpyDeviceToHost);
        double em = 0, ev = 0, dm = 0;
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            long double t = C[r * 16 + c];
            for (int kk = 0; kk < K; ++kk) t -= (long double)A[r * K + kk] * (long double)B[c * K + kk];
            em = fmax(em, fabs((double)((D[r * 16 + c] - t) / t))); ev = fmax(ev, fabs((double)((Dv[r * 16 + c] - t) / t)));
            dm = fmax(dm, fabs(D[r * 16 + c] - Dv[r * 16 + c]));
        }
        printf("product scale %.0e: max rel err MFMA %.3e | VALU fma chain %.3e | max |MFMA - VALU| %.3e\n", scale, em, ev, dm);
    }
    return 0;
}
