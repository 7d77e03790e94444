// Cycles of the S1 tile's general Toeplitz-table GEMMs (bdrt_tile_s1.h::toep_gemm_gen) per wave over a list of shapes, beside the MFMA
// count of the busiest SIMD (64 cycles each: the floor), and of the default shapes' routine at 81 x 161.
// Build (from tools/ubench): hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I../../bayes_drt_amd/csrc -I../../include
//   -o toep_gen_probe toep_gen_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../bayes_drt_amd/csrc/bdrt_tile_s1.h"
using namespace bdrt;
__global__ __launch_bounds__(512) void k(const double *in, double *out, long long *cyc, int nf, int K, int tlen, int reps, int gen)
{
    extern __shared__ double smem[];
    for (int e = threadIdx.x; e < 18000; e += 512) smem[e] = in[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *Xs = smem, *Zh = smem + 208 * 16, *Tt = smem + (208 + 256) * 16 + 6656;
    long long t0 = 0, t1 = 0, t2 = 0, f = 0, b = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        t0 = clock64();
        if (gen) toep_gemm_gen<true>(nf, K, tlen, Tt, Xs, Zh, wave, lane); else toep_gemm<true>(nf, K, tlen, Tt, Xs, Zh, wave, lane);
        t1 = clock64();
        __syncthreads();
        long long t1b = clock64();
        if (gen) toep_gemm_gen<false>(nf, K, tlen, Tt, Zh, Xs, wave, lane); else toep_gemm<false>(nf, K, tlen, Tt, Zh, Xs, wave, lane);
        t2 = clock64();
        if (r) { f += t1 - t0; b += t2 - t1b; }
    }
    out[threadIdx.x] = Xs[threadIdx.x] + Zh[threadIdx.x];
    if (blockIdx.x == 0 && lane == 0) { cyc[wave] = f / (reps - 1); cyc[8 + wave] = b / (reps - 1); }
}
static int tiles(int m) { return (m & 15) <= 2 ? (m >> 4) : ((m + 15) >> 4); }
int main(int argc, char **argv)
{
    double *in, *out; long long *cyc;
    const size_t lds = 150 * 1024;
    hipMalloc(&in, 18000 * 8); hipMemset(in, 0, 18000 * 8); hipMalloc(&out, 4096); hipMalloc(&cyc, 128);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int shapes[][3] = {{81, 161, 0}, {81, 161, 1}, {81, 101, 1}, {81, 81, 1}, {41, 51, 1}, {53, 81, 1}, {106, 101, 1}, {128, 192, 1}};
    for (auto &s : shapes) {
        const int nf = s[0], K = s[1], gen = s[2];
        const int tlen = gen ? (16 + nf + K - 1 + 16 + 1) & ~1 : (8 + nf + K - 1 + 8 + 1) & ~1;
        {
            static double host[18000];
            for (double &v : host) v = 0.0;
            unsigned *st = (unsigned *)(host + (208 + 256) * 16 + 6656 + 2 * tlen);
            if (gen)
                for (int dir = 0; dir < 2; ++dir)
                    for (int w = 0; w < 8; ++w)
                        if (toep_gen_steps(dir == 0, nf, K, tlen, w, st + (dir * 8 + w) * 2 * TOEP_STEPS) < 0) { printf("step list does not fit\n"); return 1; }
            hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
        }
        hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, in, out, cyc, nf, K, tlen, 11, gen);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        const int Tf = 2 * tiles(nf), Tb = tiles(K), cf = (K + 3) / 4, cb = 2 * ((nf + 3) / 4);
        printf("%3d x %3d %s  fwd %2d tiles x %2d MFMA (ideal %5.0f cyc/SIMD):", nf, K, gen ? "gen" : "def", Tf, cf, Tf * cf * 64 / 4.0);
        for (int w = 0; w < 8; ++w) printf(" %5lld", h[w]);
        printf("\n                bwd %2d tiles x %2d MFMA (ideal %5.0f cyc/SIMD):", Tb, cb, Tb * cb * 64 / 4.0);
        for (int w = 0; w < 8; ++w) printf(" %5lld", h[8 + w]);
        printf("\n");
    }
    return 0;
}
