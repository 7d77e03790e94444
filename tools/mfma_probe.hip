// mfma_probe.hip -- micro-probe: fp64 MFMA issue rates on gfx950 and the C/D layout of 16x16x4.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void rate_kernel(double *out, int iters, double a0, double b0)
{
    d4 acc[8];
    double acc1[8];
    for (int i = 0; i < 8; ++i) { acc[i] = d4{0, 0, 0, 0}; acc1[i] = 0; }
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else if (MODE == 1) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            else acc1[i] = fma(a, b, acc1[i]);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void layout_kernel(const double *A, const double *B, double *D)   // A[16x4], B[4x16] row-major
{
    int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

__global__ void layout4_kernel(const double *A, const double *B, double *D)   // 4 blocks of 4x4x4
{
    int l = threadIdx.x;
    // hypothesis: block = l >> 4 ; within block: A[i = l & 3][k = (l >> 2) & 3], B[k = (l>>2)&3][j = l & 3]
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
    D[l] = d;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    double *out; hipMalloc(&out, 1 << 22);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int wpb : {1, 2, 4, 8}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            dim3 grid(prop.multiProcessorCount), block(64 * wpb);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, grid, block, 0, 0, out, iters, 1.0, 2.0);
                else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, grid, block, 0, 0, out, iters, 1.0, 2.0);
                else hipLaunchKernelGGL(rate_kernel<2>, grid, block, 0, 0, out, iters, 1.0, 2.0);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double ninst = (double)iters * 8 * wpb * prop.multiProcessorCount;
            double flop_per = mode == 0 ? 2.0 * 16 * 16 * 4 : (mode == 1 ? 2.0 * 4 * 4 * 4 * 4 : 2.0 * 64);
            printf("mode %d (%s) waves/CU %d: %.3f ms, %.2f TFLOP/s, %.1f ns/inst/wave\n", mode,
                   mode == 0 ? "mfma_f64_16x16x4" : (mode == 1 ? "mfma_f64_4x4x4_4b" : "v_fma_f64"), wpb, ms,
                   ninst * flop_per / ms * 1e-9, ms * 1e6 / ((double)iters * 8));
        }
    }
    // layout check 16x16x4 with asymmetric data
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + 0.37 * i; B[i] = -2.0 + 0.11 * i * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("16x16x4 layout max err %.3e (%s)\n", err, err < 1e-9 ? "OK" : "MISMATCH");
    // 4x4x4_4b: dump so the mapping can be inferred: A[l] = 2^(l) style coding
    for (int i = 0; i < 64; ++i) { A[i] = i + 1; B[i] = 1000.0 * (i + 1); }
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout4_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    // test hypothesis: block b = l>>4, i = l&3 ... try several and report which matches
    auto test = [&](int hyp) {
        double e = 0;
        for (int l = 0; l < 64; ++l) {
            double r = 0;
            for (int k = 0; k < 4; ++k) {
                int la, lb;
                if (hyp == 0) { int blk = l >> 4, i = l & 3, j = (l >> 2) & 3; la = blk * 16 + k * 4 + i; lb = blk * 16 + k * 4 + j; }
                else if (hyp == 1) { int blk = l >> 4, j = l & 3, i = (l >> 2) & 3; la = blk * 16 + k * 4 + i; lb = blk * 16 + k * 4 + j; }
                else if (hyp == 2) { int blk = l & 3, i = (l >> 2) & 3, j = l >> 4; la = (k * 4 + i) * 4 + blk; lb = (k * 4 + j) * 4 + blk; }
                else { int blk = l >> 4, i = l & 3, j = (l >> 2) & 3; la = blk * 16 + i * 4 + k; lb = blk * 16 + j * 4 + k; }
                r += A[la] * B[lb];
            }
            e = fmax(e, fabs(D[l] - r));
        }
        return e;
    };
    for (int h = 0; h < 4; ++h) printf("4x4x4_4b hypothesis %d err %.3e\n", h, test(h));
    printf("D[0..7] = %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", D[0], D[1], D[2], D[3], D[4], D[5], D[6], D[7]);
    return 0;
}
