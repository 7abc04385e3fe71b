// accuracy of v_rcp_f64 / v_rsq_f64 seeds and Newton steps on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const double* p, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = p[i];
    double y0 = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y0, 1.0); double y1 = fma(y0, e, y0);
    e = fma(-x, y1, 1.0); double y2 = fma(y1, e, y1);
    double r0 = __builtin_amdgcn_rsq(x);
    double f = fma(-x * r0, r0, 1.0); double r1 = fma(0.5 * r0, f, r0);
    f = fma(-x * r1, r1, 1.0); double r2 = fma(0.5 * r1, f, r1);
    out[i * 6 + 0] = y0; out[i * 6 + 1] = y1; out[i * 6 + 2] = y2;
    { double e = fma(-x * r0, r0, 1.0); double c = fma(0.375, e, 0.5); r1 = fma(r0 * e, c, r0); }   // Halley
    out[i * 6 + 3] = r0; out[i * 6 + 4] = r1; out[i * 6 + 5] = r2;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> p(n), o(6 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); p[i] = exp((u - 0.5) * 60.0); }
    double *dp, *dout; hipMalloc(&dp, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    long double sg[6] = {0, 0, 0, 0, 0, 0}, sq2 = 0, sq2b = 0;
    for (int i = 0; i < n; ++i) {
        long double ry = 1.0L / (long double)p[i], rr = 1.0L / sqrtl((long double)p[i]);
        for (int j = 0; j < 3; ++j) { long double es = ((long double)o[i * 6 + j] - ry) / ry; sg[j] += es; double e = fabs((double)es); if (e > m[j]) m[j] = e; }
        for (int j = 3; j < 6; ++j) { long double es = ((long double)o[i * 6 + j] - rr) / rr; sg[j] += es; double e = fabs((double)es); if (e > m[j]) m[j] = e; }
        { long double s0 = (long double)(p[i] * o[i * 6 + 5]); long double st = sqrtl((long double)p[i]); sq2 += (s0 - st) / st; long double sl = (long double)sqrt(p[i]); sq2b += (sl - st) / st; }
    }
    printf("rcp: seed %.3e  1NR %.3e  2NR %.3e   (eps = %.3e)\n", m[0], m[1], m[2], 2.22e-16);
    printf("rsq: seed %.3e  1 Halley %.3e  2NR %.3e\n", m[3], m[4], m[5]);
    printf("mean signed rel err: rcp 2NR %.3Le | rsq 2NR %.3Le | p*rsq(p) as sqrt %.3Le | host sqrt %.3Le\n", sg[2] / n, sg[5] / n, sq2 / n, sq2b / n);
    return 0;
}
