// Calibration of rocprofv3 FETCH_SIZE on gfx950 by access width (MI355X_MICROARCH.md: FETCH_SIZE reports HALF the bytes of a
// wide coalesced 16-B-per-lane streaming read; "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern").  Each kernel streams the same 1 GiB buffer once, with 4 / 8 / 16 bytes per lane, plus the access
// shape of spmm_bxt_tiles' gather (8 bytes per lane, 16 rows a stride apart per thread).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/bin/fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ p, size_t n, double* sink) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        s += (double)reinterpret_cast<const unsigned char*>(&v)[0];
    }
    if (s == 12345.678) sink[0] = s;
}
// thread t of a workgroup reads column (256 * blockIdx.y + t) of 16 consecutive rows (row stride ld doubles)
__global__ __launch_bounds__(256) void gather_rows(const double* __restrict__ X, int ld, double* sink) {
    const double* p = X + (size_t)blockIdx.x * 16 * ld + blockIdx.y * 256 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += p[(size_t)r * ld];
    if (s == 12345.678) sink[0] = s;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf; double* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_read<float>, dim3(8192), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(stream_read<double>, dim3(8192), dim3(256), 0, 0, (const double*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(stream_read<double2>, dim3(8192), dim3(256), 0, 0, (const double2*)buf, bytes / 16, sink);
    // 1 GiB as a matrix of 8192-double rows: 16384 rows -> grid (1024 row groups, 32 column groups)
    hipLaunchKernelGGL(gather_rows, dim3(1024, 32), dim3(256), 0, 0, (const double*)buf, 8192, sink);
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes\n", bytes);
    return 0;
}
