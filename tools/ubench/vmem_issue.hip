// What a vector-memory INSTRUCTION costs on one CU, in shader cycles (s_memtime): a 512-thread workgroup (two waves per SIMD, as the
// sampler kernel) where every wave issues batches of NB independent global loads / stores of 8 or 16 bytes per lane on rows that
// stay in L2 (each wave walks its own 64 KB), `active` of the eight waves at a time.  One workgroup per CU on every CU (the
// sampler's situation), the figures are those of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 -o vmem_issue vmem_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NB, int BYTES, bool STORE>
__global__ __launch_bounds__(512) void k(double *buf, int iters, int active, long long *cyc, double *sink)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *base = buf + ((size_t)blockIdx.x * 8 + w) * 8192;          // 64 KB per wave
    double acc = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    if (w < active) {
        for (int it = 0; it < iters; ++it) {
            const int off = (it & 3) * 2048;
            if (BYTES == 8) {
                double v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    double *p = base + off + i * 64 + lane;
                    if (STORE) *p = acc + i; else v[i] = *p;
                }
                if (!STORE) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) acc += v[i];
                }
            } else {
                d2 v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    d2 *p = reinterpret_cast<d2 *>(base + off) + i * 64 + lane;
                    if (STORE) *p = d2{acc + i, acc}; else v[i] = *p;
                }
                if (!STORE) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) acc += v[i].x + v[i].y;
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = clock64();
    if (acc == 12345.678) sink[threadIdx.x] = acc;
    if (blockIdx.x == 0 && lane == 0) cyc[w] = t1 - t0;
}
template <int NB, int BYTES, bool STORE>
void run(double *buf, long long *c, double *sink, int nwg, const char *name)
{
    const int iters = 2000;
    for (int active : {1, 2, 4, 8}) {
        hipLaunchKernelGGL((k<NB, BYTES, STORE>), dim3(nwg), dim3(512), 0, 0, buf, iters, active, c, sink);
        hipLaunchKernelGGL((k<NB, BYTES, STORE>), dim3(nwg), dim3(512), 0, 0, buf, iters, active, c, sink);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < active; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-28s batch %2d  waves %d  %7.1f cycles per instruction and wave, %6.1f per instruction on the CU\n", name, NB, active,
               (double)mx / (iters * NB), (double)mx / (iters * NB * active));
    }
}
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int nwg = getenv("NWG") ? atoi(getenv("NWG")) : pr.multiProcessorCount;
    double *buf, *sink; long long *c;
    hipMalloc(&buf, (size_t)nwg * 8 * 8192 * 8); hipMemset(buf, 0, (size_t)nwg * 8 * 8192 * 8);
    hipMalloc(&sink, 4096); hipMalloc(&c, 64);
    run<11, 8, false>(buf, c, sink, nwg, "load  8 B per lane");
    run<6, 16, false>(buf, c, sink, nwg, "load 16 B per lane");
    run<22, 8, false>(buf, c, sink, nwg, "load  8 B per lane");
    run<11, 8, true>(buf, c, sink, nwg, "store  8 B per lane");
    run<6, 16, true>(buf, c, sink, nwg, "store 16 B per lane");
    return 0;
}
