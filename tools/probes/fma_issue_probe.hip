// fp64 FMA issue rate of ONE wave against two waves on a SIMD (gfx950): cycles per v_fma_f64 seen by a wave that issues
// independent FMAs (8 accumulators) back to back.  Build: hipcc -O3 --offload-arch=gfx950 fma_issue_probe.hip -o fma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MOVS>
__global__ __launch_bounds__(64) void probe(double *out, long long *cyc, int iters, double a, double b)
{
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
    double x = a, y = b;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], x, y);
            if (MOVS) {
#pragma unroll
                for (int i = 0; i < MOVS; ++i) __asm__ volatile("v_mov_b32 %0, %0" : "+v"(x));
            }
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    int ncu = 256;
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 4096 * 64 * sizeof(double)); hipMalloc(&cyc, 8 * 4096 * sizeof(long long));
    const int iters = 200;
    for (int per_cu : {1, 4, 8, 16}) {
        const int n = ncu * per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            // dynamic LDS keeps the waves of a CU at `per_cu` (160 KB / per_cu each)
            const size_t lds = (size_t)(160 * 1024 / per_cu / 1280) * 1280 - 128;
            hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(probe<0>, dim3(n), dim3(64), lds > 65536 ? 65536 : lds, 0, out, cyc, iters, 1.0000001, 1e-9);
            hipDeviceSynchronize();
        }
        std::vector<long long> h(n);
        hipMemcpy(h.data(), cyc, n * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("%2d waves per CU: %.2f cycles per v_fma_f64 per wave (128 FMAs per iteration, %d iterations)\n", per_cu, s / n / (iters * 128.0), iters);
    }
    return 0;
}
