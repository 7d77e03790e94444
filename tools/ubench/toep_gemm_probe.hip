// Cycle stamps inside the S1 tile's Toeplitz-table GEMM (bdrt_tile_s1.h::toep_gemm) at the 81 x 161 shape, with all eight waves, with
// waves 0 and 4 (one SIMD) and with wave 0 alone.  Build (from tools/ubench): hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast
//   -I../../bayes_drt_amd/csrc -I../../include -o toep_gemm_probe toep_gemm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ long long g_st[8][2][16];
#define BDRT_TOEP_STAMP(slot) do { if ((lane) == 0 && blockIdx.x == 0) g_st[wave][FWD ? 0 : 1][slot] = clock64(); } while (0)
#include "../../bayes_drt_amd/csrc/bdrt_tile_s1.h"
using namespace bdrt;
__global__ __launch_bounds__(512) void k(const double *in, double *out, long long *cyc, int nf, int K, int tlen, int reps, int wmask)
{
    extern __shared__ double smem[];
    for (int e = threadIdx.x; e < 18000; e += 512) smem[e] = in[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *Xs = smem, *Zh = smem + 176 * 16, *Tt = smem + 2 * 176 * 16 + 6528;
    long long t0 = 0, t1 = 0, t2 = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        if (lane == 0) t0 = clock64();
        if ((wmask >> wave) & 1) toep_gemm<true>(nf, K, tlen, Tt, Xs, Zh, wave, lane);
        if (lane == 0) t1 = clock64();
        __syncthreads();
        if ((wmask >> wave) & 1) toep_gemm<false>(nf, K, tlen, Tt, Zh, Xs, wave, lane);
        if (lane == 0) t2 = clock64();
    }
    out[threadIdx.x] = Xs[threadIdx.x] + Zh[threadIdx.x];
    if (blockIdx.x == 0 && lane == 0) { cyc[wave] = t1 - t0; cyc[8 + wave] = t2 - t1; }
}
int main()
{
    double *in, *out; long long *cyc;
    hipMalloc(&in, 18000 * 8); hipMemset(in, 0, 18000 * 8); hipMalloc(&out, 4096); hipMalloc(&cyc, 128);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 18000 * 8);
    const int masks[] = {0xff, 0x11, 0x01};
    for (int m : masks) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 18000 * 8, 0, in, out, cyc, 81, 161, 258, 10, m);
        hipDeviceSynchronize();
        long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        printf("waves %02x fwd:", m); for (int w = 0; w < 8; ++w) printf(" %5lld", h[w]); printf("  bwd:"); for (int w = 0; w < 8; ++w) printf(" %5lld", h[8 + w]); printf("\n");
        long long st[8][2][16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    for (int d = 0; d < 2; ++d) for (int w : {0, 4}) { printf("%s wave %d stamps (entry, opened, [mfma done, next opened, stored] x pieces):", d ? "bwd" : "fwd", w); for (int k2 = 1; k2 < 8; ++k2) printf(" %lld", st[w][d][k2] - st[w][d][0]); printf("\n"); }
    }
    return 0;
}
