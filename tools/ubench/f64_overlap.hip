// fp64 issue rates on one SIMD, in shader cycles (s_memtime): does fp64 MFMA of one wave overlap fp64 VALU of the other wave?
// 512 threads = two waves per SIMD: waves 0-3 run `ma`, waves 4-7 run `mb` (0 idle, 1 MFMA f64 16x16x4, 2 v_fma_f64 x 16
// independent accumulators, 3 v_add_u32 x 16).  Build: hipcc --offload-arch=gfx950 -O3 -o f64_overlap f64_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double work(int m, int n)
{
    double r = 0;
    if (m == 1) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + x;
        for (int i = 0; i < n; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (m == 2) {
        double v[16], b = 1.0000001 + threadIdx.x * 1e-12, c = 1e-9 * threadIdx.x;
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = threadIdx.x + j;
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 16; j++) v[j] = __builtin_fma(v[j], b, c);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) r += v[j];
    } else if (m == 3) {
        unsigned v[16], b = threadIdx.x | 1;
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = threadIdx.x + j;
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 16; j++) v[j] = (v[j] ^ b) + (v[j] >> 3);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) r += v[j];
    }
    return r;
}
__global__ __launch_bounds__(512) void k(int ma, int mb, int n, double *out, long long *cyc)
{
    const int w = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = clock64();
    const double r = work(w < 4 ? ma : mb, n);
    const long long t1 = clock64();
    if (r == 12345.678) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 4096); hipMalloc(&c, 64);
    const int n = 20000;
    const char *nm[] = {"idle", "mfma_f64_16x16x4 (4 per iter)", "v_fma_f64 (64 per iter)", "u32 xor/shift/add (192 per iter)"};
    const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (auto &cs : cases) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, cs[0], cs[1], n, o, c);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("waves 0-3: %-34s waves 4-7: %-34s cycles per iteration: %8.1f | %8.1f\n", nm[cs[0]], nm[cs[1]], (double)h[0] / n, (double)h[4] / n);
    }
    return 0;
}
