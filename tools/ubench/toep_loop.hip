// MFMA f64 16x16x4 fed from LDS: cycles per MFMA and SIMD for variants of the operand loop (two waves per SIMD, as in the S1 tile).
// Build: hipcc --offload-arch=gfx950 -O3 -o toep_loop toep_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) double *lp;
__device__ __forceinline__ d4 mf(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
#define SB __builtin_amdgcn_sched_barrier(0)
template <int MODE>
__global__ __launch_bounds__(512) void k(const double *in, double *out, long long *cyc, int n)
{
    extern __shared__ double smem[];
    for (int e = threadIdx.x; e < 16384; e += 512) smem[e] = in[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    lp a = (lp)(smem + 12000 + i - kq + wave * 16);
    lp b0 = (lp)(smem + (kq) * 16 + (i ^ kq)), b1 = (lp)(smem + (4 + kq) * 16 + (i ^ (4 + kq))), b2 = (lp)(smem + (8 + kq) * 16 + (i ^ (8 + kq))),
       b3 = (lp)(smem + (12 + kq) * 16 + (i ^ (12 + kq)));
    __asm__ volatile("" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    int zero = 0; __asm__ volatile("" : "+v"(zero));
    double A0[4], B0[4], A1[4], B1[4];
#define LD(A, B, r) do { _Pragma("unroll") for (int j = 0; j < 4; ++j) A[j] = a[44 - 16 * r - 4 * j]; B[0] = b0[256 * r]; B[1] = b1[256 * r]; B[2] = b2[256 * r]; B[3] = b3[256 * r]; } while (0)
    for (int j = 0; j < 4; ++j) { A0[j] = A1[j] = 1.0 + lane; B0[j] = B1[j] = 0.5; }
    if (MODE != 0) LD(A0, B0, 0);
    __syncthreads();
    const long long t0 = clock64();
    for (int p = 0; p < n; ++p) {
        if (MODE == 0) {          // no loads
            c0 = mf(A0[0], B0[0], c0); c1 = mf(A0[1], B0[1], c1); c0 = mf(A0[2], B0[2], c0); c1 = mf(A0[3], B0[3], c1);
            c0 = mf(A1[0], B1[0], c0); c1 = mf(A1[1], B1[1], c1); c0 = mf(A1[2], B1[2], c0); c1 = mf(A1[3], B1[3], c1);
        } else if (MODE == 1 || MODE == 3) {   // loads behind the first MFMA of each quad (1: two accumulators, 3: four)
            c0 = mf(A0[0], B0[0], c0); SB; LD(A1, B1, 1); SB;
            c1 = mf(A0[1], B0[1], c1);
            if (MODE == 3) { c2 = mf(A0[2], B0[2], c2); c3 = mf(A0[3], B0[3], c3); } else { c0 = mf(A0[2], B0[2], c0); c1 = mf(A0[3], B0[3], c1); }
            SB;
            c0 = mf(A1[0], B1[0], c0); SB; LD(A0, B0, 2); SB;
            c1 = mf(A1[1], B1[1], c1);
            if (MODE == 3) { c2 = mf(A1[2], B1[2], c2); c3 = mf(A1[3], B1[3], c3); } else { c0 = mf(A1[2], B1[2], c0); c1 = mf(A1[3], B1[3], c1); }
            SB;
        } else if (MODE == 4) {   // mode 1 + five pointer increments per two quads
            c0 = mf(A0[0], B0[0], c0); SB; LD(A1, B1, 1); SB;
            c1 = mf(A0[1], B0[1], c1); c0 = mf(A0[2], B0[2], c0); c1 = mf(A0[3], B0[3], c1);
            SB;
            c0 = mf(A1[0], B1[0], c0); SB; LD(A0, B0, 2); a += zero; b0 += zero; b1 += zero; b2 += zero; b3 += zero; SB;
            c1 = mf(A1[1], B1[1], c1); c0 = mf(A1[2], B1[2], c0); c1 = mf(A1[3], B1[3], c1);
            SB;
        } else if (MODE == 5) {   // mode 1 without scheduling barriers: the compiler's own order and wait counts
            c0 = mf(A0[0], B0[0], c0); LD(A1, B1, 1);
            c1 = mf(A0[1], B0[1], c1); c0 = mf(A0[2], B0[2], c0); c1 = mf(A0[3], B0[3], c1);
            c0 = mf(A1[0], B1[0], c0); LD(A0, B0, 2);
            c1 = mf(A1[1], B1[1], c1); c0 = mf(A1[2], B1[2], c0); c1 = mf(A1[3], B1[3], c1);
        } else if (MODE == 6) {   // mode 1 with an s_nop between the MFMAs of a quad
            c0 = mf(A0[0], B0[0], c0); SB; LD(A1, B1, 1); SB;
            c1 = mf(A0[1], B0[1], c1); SB; __asm__ volatile("s_nop 0"); SB; c0 = mf(A0[2], B0[2], c0); SB; __asm__ volatile("s_nop 0"); SB; c1 = mf(A0[3], B0[3], c1);
            SB;
            c0 = mf(A1[0], B1[0], c0); SB; LD(A0, B0, 2); SB;
            c1 = mf(A1[1], B1[1], c1); SB; __asm__ volatile("s_nop 0"); SB; c0 = mf(A1[2], B1[2], c0); SB; __asm__ volatile("s_nop 0"); SB; c1 = mf(A1[3], B1[3], c1);
            SB;
        } else if (MODE == 2) {   // loads in front of the quad before
            SB; LD(A1, B1, 1); SB;
            c0 = mf(A0[0], B0[0], c0); c1 = mf(A0[1], B0[1], c1); c0 = mf(A0[2], B0[2], c0); c1 = mf(A0[3], B0[3], c1);
            SB; LD(A0, B0, 2); SB;
            c0 = mf(A1[0], B1[0], c0); c1 = mf(A1[1], B1[1], c1); c0 = mf(A1[2], B1[2], c0); c1 = mf(A1[3], B1[3], c1);
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
template <int MODE> void run(const char *name, double *in, double *out, long long *cyc)
{
    const int n = 4000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 16384 * 8, 0, in, out, cyc, n);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-52s cycles per MFMA and SIMD: %.1f (wave 0), %.1f (wave 4)\n", name, h[0] / (16.0 * n), h[4] / (16.0 * n));
}
int main()
{
    double *in, *out; long long *cyc;
    hipMalloc(&in, 16384 * 8); hipMemset(in, 0, 16384 * 8); hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run<0>("operands in registers", in, out, cyc);
    run<1>("LDS operands, loads behind the quad's first MFMA", in, out, cyc);
    run<3>("  ... with four accumulators", in, out, cyc);
    run<2>("LDS operands, loads in front of the quad", in, out, cyc);
    run<4>("mode 1 + five pointer increments per two quads", in, out, cyc);
    run<5>("mode 1, compiler-scheduled (progressive waits)", in, out, cyc);
    run<6>("mode 1 + s_nop between the MFMAs of a quad", in, out, cyc);
    return 0;
}
