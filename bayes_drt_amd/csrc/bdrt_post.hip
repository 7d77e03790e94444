// bdrt_post.hip -- posterior post-processing on the device (include/bdrt.h section (4)).
//
// Replaces the numpy reductions the reference applies to the HMC draws right after `sampling`
// (reference bayes_drt/inversion.py): np.percentile(samples, q, axis=0) in coef_percentile (:2560), predict_Z (:2702,
// :2734-2735), predict_Rp (:3068, :3085), predict_sigma (:3096-3113), and the draws-times-basis products in front of
// them (predict_Z: Z_pred_matrix = x_samples A^T + offsets; predict_distribution: F = coef Phi^T).
//
//   project_kernel     Y[rows x M] = X[rows x K] Phi[M x K]^T + bias[M]      one wave per 16x16 tile, v_mfma_f64_16x16x4_f64
//   percentile_kernel  one workgroup per column: the column goes to LDS, bitonic sort, numpy's default ('linear')
//                      percentile rule with numpy's own lerp formula (bit-identical to np.percentile for finite data;
//                      a column with a NaN gives NaN, as numpy does)
//
// The draws never have to leave the GPU: bdrt_sampler_percentiles works on the sampler's device buffer.
#include <cmath>

#include "bdrt_host.h"

namespace bdrt {

constexpr int PCT_MAX_ROWS = 16384;   // a column of up to 16384 samples (128 KiB) is sorted in LDS, a longer one in HBM

__global__ __launch_bounds__(64) void project_kernel(const double *__restrict__ X, int rows, int K, long ldx,
                                                     const double *__restrict__ Phi, int M,
                                                     const double *__restrict__ bias, double *__restrict__ Y)
{
    const int lane = threadIdx.x, col = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int ra = r0 + col, mb = m0 + col;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kq;
        const double a = (ra < rows && k < K) ? X[(size_t)ra * ldx + k] : 0.0;      // A[i][k] = X[r0 + i][k]
        const double b = (mb < M && k < K) ? Phi[(size_t)mb * K + k] : 0.0;         // B[k][j] = Phi[m0 + j][k]
        acc = mfma_f64(a, b, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + kq + 4 * q, m = m0 + col;
        if (r < rows && m < M) Y[(size_t)r * M + m] = acc[q] + (bias ? bias[m] : 0.0);
    }
}

// numpy.lib.function_base._lerp (numpy >= 1.22): a + (b - a) t, and b - (b - a)(1 - t) where t >= 0.5.
// This file is compiled with -ffp-contract=off (Makefile): every product and sum is rounded separately, as in numpy's
// element-wise ufuncs; a fused multiply-add would differ in the last bit.
__device__ inline double numpy_lerp(double a, double b, double t)
{
    const double d = b - a;
    const double dt = d * t;
    double r = a + dt;
    if (t >= 0.5) {
        const double omt = 1.0 - t;
        const double dm = d * omt;
        r = b - dm;
    }
    return r;
}

// One workgroup per column.  IN_LDS: the column (padded to n2, a power of two) is sorted in LDS; otherwise (more than
// PCT_MAX_ROWS samples) in the column's slice of the global scratch `work` -- same network, same result.
// expcol[col] != 0: the samples are exp(Y) (constrained scale of a Stan <lower=0> parameter; exp is monotone, so the
// order statistics are those of Y and only the interpolation differs).  mean_out: sample mean of every column.
template <bool IN_LDS>
__global__ __launch_bounds__(512) void percentile_kernel(const double *__restrict__ Y, int rows, long ld_row, long ld_col,
                                                         int ncols, int n2, const double *__restrict__ q, int nq,
                                                         double *__restrict__ out, double *__restrict__ work,
                                                         const unsigned char *__restrict__ expcol,
                                                         double *__restrict__ mean_out)
{
    extern __shared__ __attribute__((aligned(16))) double sh_[];
    const int col = blockIdx.x, tid = threadIdx.x;
    double *red = sh_;                                              // 512 doubles: reduction scratch of the mean
    double *sh = IN_LDS ? sh_ + 512 : work + (size_t)col * n2;
    const bool ex = expcol && expcol[col];
    int flag = 0;
    double part = 0.0;
    for (int i = tid; i < n2; i += 512) {
        double v = INFINITY;
        if (i < rows) {
            v = Y[(size_t)i * ld_row + (size_t)col * ld_col];
            if (ex) v = exp(v);
            part += v;
        }
        if (v != v) { flag = 1; v = INFINITY; }
        sh[i] = v;
    }
    red[tid] = part;
    const int has_nan = __syncthreads_or(flag);
    if (mean_out) {
        for (int w = 256; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0) mean_out[col] = has_nan ? NAN : red[0] / (double)rows;
    }
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 512) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool asc = (i & k) == 0;
                    const double a = sh[i], b = sh[ixj];
                    if ((a > b) == asc) { sh[i] = b; sh[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < nq; t += 512) {
        // numpy: virtual index (n - 1) * (q / 100); previous = floor; gamma = virtual - previous; indexes clipped
        const double quant = q[t];                       // already q / 100, divided on the host (IEEE division, as numpy)
        const double virt = (double)(rows - 1) * quant;
        double prev = floor(virt);
        double gamma = virt - prev;
        long lo = (long)prev, hi = lo + 1;
        if (virt >= (double)(rows - 1)) { lo = rows - 1; hi = rows - 1; gamma = virt - prev; }
        if (virt < 0.0) { lo = 0; hi = 0; }
        lo = lo < 0 ? 0 : (lo > rows - 1 ? rows - 1 : lo);
        hi = hi < 0 ? 0 : (hi > rows - 1 ? rows - 1 : hi);
        const double r = numpy_lerp(sh[lo], sh[hi], gamma);
        out[(size_t)t * ncols + col] = has_nan ? NAN : r;
    }
}

static int launch_percentiles(const double *dY, int rows, long ld_row, int ncols, const double *dq, int nq, double *dOut,
                              const unsigned char *dExp, double *dMean, hipStream_t stream)
{
    int n2 = 1;
    while (n2 < rows) n2 <<= 1;
    if (rows <= PCT_MAX_ROWS) {
        const size_t lds = (size_t)(n2 + 512) * sizeof(double);
        static LdsAttrCache attr_cache;
        BDRT_HIP(attr_cache.ensure(lds, [&]() {
            return hipFuncSetAttribute((const void *)percentile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }));
        hipLaunchKernelGGL(percentile_kernel<true>, dim3(ncols), dim3(512), lds, stream, dY, rows, ld_row, 1L, ncols, n2, dq, nq,
                           dOut, (double *)nullptr, dExp, dMean);
        BDRT_HIP(hipGetLastError());
        BDRT_HIP(hipStreamSynchronize(stream));
        return 0;
    }
    // long columns (np.percentile has no row limit): sort in a global scratch slice per column
    double *work = nullptr;
    BDRT_HIP(hipMalloc((void **)&work, (size_t)ncols * n2 * sizeof(double)));
    hipLaunchKernelGGL(percentile_kernel<false>, dim3(ncols), dim3(512), 512 * sizeof(double), stream, dY, rows, ld_row, 1L, ncols,
                       n2, dq, nq, dOut, work, dExp, dMean);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(work);
    if (e != hipSuccess) { set_error("bdrt percentiles: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

// device core: percentiles of (X Phi^T + bias) or of X itself (Phi == nullptr); all pointers are device pointers
static int percentiles_dev(const double *dX, int rows, int K, long ldx, const double *dPhi, int M, const double *dBias,
                           const double *dq, int nq, double *dOut, const unsigned char *dExp, double *dMean, hipStream_t stream)
{
    if (rows < 1) { set_error("bdrt percentiles: no sample rows"); return -3; }
    if (dPhi) {
        double *dY = nullptr;
        BDRT_HIP(hipMalloc((void **)&dY, (size_t)rows * M * sizeof(double)));
        hipLaunchKernelGGL(project_kernel, dim3((rows + 15) / 16, (M + 15) / 16), dim3(64), 0, stream, dX, rows, K, ldx, dPhi,
                           M, dBias, dY);
        hipError_t e = hipGetLastError();
        int rc = 0;
        if (e != hipSuccess) { set_error("bdrt percentiles: %s", hipGetErrorString(e)); rc = -10; }
        if (rc == 0) rc = launch_percentiles(dY, rows, (long)M, M, dq, nq, dOut, nullptr, dMean, stream);
        hipFree(dY);
        return rc;
    }
    return launch_percentiles(dX, rows, ldx, K, dq, nq, dOut, dExp, dMean, stream);
}

int post_percentiles_device(const double *dX, int rows, int K, long ldx, const double *Phi, int M, const double *bias,
                            const double *q, int nq, double *out, const unsigned char *expcol, double *mean)
{
    const int ncols = Phi ? M : K;
    double *dPhi = nullptr, *dBias = nullptr, *dq = nullptr, *dOut = nullptr, *dMean = nullptr;
    unsigned char *dExp = nullptr;
    auto cleanup = [&]() { hipFree(dPhi); hipFree(dBias); hipFree(dq); hipFree(dOut); hipFree(dMean); hipFree(dExp); };
#define PP_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    if (Phi) {
        PP_HIP(hipMalloc((void **)&dPhi, (size_t)M * K * sizeof(double)));
        PP_HIP(hipMemcpy(dPhi, Phi, (size_t)M * K * sizeof(double), hipMemcpyHostToDevice));
        if (bias) {
            PP_HIP(hipMalloc((void **)&dBias, (size_t)M * sizeof(double)));
            PP_HIP(hipMemcpy(dBias, bias, (size_t)M * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    std::vector<double> quant(nq);
    for (int t = 0; t < nq; ++t) quant[t] = q[t] / 100.0;           // np.true_divide(q, 100)
    PP_HIP(hipMalloc((void **)&dq, (size_t)nq * sizeof(double)));
    PP_HIP(hipMemcpy(dq, quant.data(), (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
    PP_HIP(hipMalloc((void **)&dOut, (size_t)nq * ncols * sizeof(double)));
    if (expcol && !Phi) {
        PP_HIP(hipMalloc((void **)&dExp, (size_t)K));
        PP_HIP(hipMemcpy(dExp, expcol, (size_t)K, hipMemcpyHostToDevice));
    }
    if (mean) PP_HIP(hipMalloc((void **)&dMean, (size_t)ncols * sizeof(double)));
    const int rc = percentiles_dev(dX, rows, K, ldx, dPhi, M, dBias, dq, nq, dOut, dExp, dMean, nullptr);
    if (rc) { cleanup(); return rc; }
    PP_HIP(hipMemcpy(out, dOut, (size_t)nq * ncols * sizeof(double), hipMemcpyDeviceToHost));
    if (mean) PP_HIP(hipMemcpy(mean, dMean, (size_t)ncols * sizeof(double), hipMemcpyDeviceToHost));
#undef PP_HIP
    cleanup();
    return 0;
}

}  // namespace bdrt

using namespace bdrt;

extern "C" {

int bdrt_percentiles(const double *X, int rows, int K, long ldx, const double *Phi, int M, const double *bias,
                     const double *q, int nq, double *out)
{
    if (!X || rows < 1 || K < 1 || ldx < K || !q || nq < 1 || !out || (Phi && M < 1)) {
        set_error("bdrt_percentiles: bad arguments");
        return -1;
    }
    bind_process_device();
    double *dX = nullptr;
    const size_t nb = ((size_t)(rows - 1) * ldx + K) * sizeof(double);
    if (hipMalloc((void **)&dX, nb) != hipSuccess) { set_error("bdrt_percentiles: hipMalloc(%zu) failed", nb); return -10; }
    if (hipMemcpy(dX, X, nb, hipMemcpyHostToDevice) != hipSuccess) { hipFree(dX); set_error("bdrt_percentiles: copy failed"); return -10; }
    const int rc = post_percentiles_device(dX, rows, K, ldx, Phi, M, bias, q, nq, out);
    hipFree(dX);
    return rc;
}

int bdrt_summary(const double *X, int rows, int K, long ldx, const unsigned char *is_pos, const double *q, int nq,
                 double *mean, double *pct)
{
    if (!X || rows < 1 || K < 1 || ldx < K || !q || nq < 1 || !pct) { set_error("bdrt_summary: bad arguments"); return -1; }
    bind_process_device();
    double *dX = nullptr;
    const size_t nb = ((size_t)(rows - 1) * ldx + K) * sizeof(double);
    if (hipMalloc((void **)&dX, nb) != hipSuccess) { set_error("bdrt_summary: hipMalloc(%zu) failed", nb); return -10; }
    if (hipMemcpy(dX, X, nb, hipMemcpyHostToDevice) != hipSuccess) { hipFree(dX); set_error("bdrt_summary: copy failed"); return -10; }
    const int rc = post_percentiles_device(dX, rows, K, ldx, nullptr, 0, nullptr, q, nq, pct, is_pos, mean);
    hipFree(dX);
    return rc;
}

}  // extern "C"
