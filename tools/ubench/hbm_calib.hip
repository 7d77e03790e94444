// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the sampler's access pattern (MI355X guide: "calibrate on a known byte count in
// your own access pattern before trusting an absolute"): every half-wave streams whole ROWS of 352 doubles, lane l taking elements
// l, l + 32, ... (8 bytes per lane, 256 contiguous bytes per half-wave and instruction) -- once over a buffer far beyond L2 and the
// Infinity Cache, so that every byte comes from / goes to HBM exactly once.  Kernels: calib_read (bytes read = the buffer),
// calib_write (bytes written = the buffer), calib_copy (both).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (separate passes); the program prints the byte counts to compare with.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_calib hbm_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int ROW = 352;          // doubles per row (32 * 11: the headline kernel's state rows)
__global__ __launch_bounds__(512) void calib_read(const double *buf, long rows_per_hw, double *sink)
{
    const int hw = (blockIdx.x * 512 + threadIdx.x) >> 5, l = threadIdx.x & 31;
    const double *p = buf + (size_t)hw * rows_per_hw * ROW;
    double acc = 0.0;
    for (long r = 0; r < rows_per_hw; ++r) {
        double v[11];
#pragma unroll
        for (int m = 0; m < 11; ++m) v[m] = p[r * ROW + l + 32 * m];
#pragma unroll
        for (int m = 0; m < 11; ++m) acc += v[m];
    }
    if (acc == 12345.678) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(512) void calib_write(double *buf, long rows_per_hw)
{
    const int hw = (blockIdx.x * 512 + threadIdx.x) >> 5, l = threadIdx.x & 31;
    double *p = buf + (size_t)hw * rows_per_hw * ROW;
    for (long r = 0; r < rows_per_hw; ++r)
#pragma unroll
        for (int m = 0; m < 11; ++m) p[r * ROW + l + 32 * m] = (double)(r + m);
}
__global__ __launch_bounds__(512) void calib_copy(const double *src, double *dst, long rows_per_hw)
{
    const int hw = (blockIdx.x * 512 + threadIdx.x) >> 5, l = threadIdx.x & 31;
    const double *p = src + (size_t)hw * rows_per_hw * ROW;
    double *q = dst + (size_t)hw * rows_per_hw * ROW;
    for (long r = 0; r < rows_per_hw; ++r) {
        double v[11];
#pragma unroll
        for (int m = 0; m < 11; ++m) v[m] = p[r * ROW + l + 32 * m];
#pragma unroll
        for (int m = 0; m < 11; ++m) q[r * ROW + l + 32 * m] = v[m] + 1.0;
    }
}
int main(int argc, char **argv)
{
    const int nwg = 256;
    const long rows = argc > 1 ? atol(argv[1]) : 256;            // rows per half-wave: 256 -> 4096 half-waves x 256 x 2816 B = 2.95 GB
    const size_t n = (size_t)nwg * 16 * rows * ROW;
    double *a, *b, *sink;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&sink, 4096);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(calib_read, dim3(nwg), dim3(512), 0, 0, a, rows, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("calib_read : %.6e bytes read, 0 written, %.3f ms (%.2f TB/s)\n", (double)n * 8, ms, n * 8 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(calib_write, dim3(nwg), dim3(512), 0, 0, b, rows); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("calib_write: 0 bytes read, %.6e written, %.3f ms (%.2f TB/s)\n", (double)n * 8, ms, n * 8 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(calib_copy, dim3(nwg), dim3(512), 0, 0, a, b, rows); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("calib_copy : %.6e bytes read, %.6e written, %.3f ms (%.2f TB/s)\n", (double)n * 8, (double)n * 8, ms, 2.0 * n * 8 / ms / 1e9);
    }
    return 0;
}
