// bdrt_ridge.hip -- ridge path (include/bdrt.h section (3)): Gram matrices on MFMA fp64 + box-constrained QP.
//
// Replaces Inverter._convex_opt (reference bayes_drt/inversion.py:1043-1067):
//   P = WA_re^T WA_re + WA_im^T WA_im + L2_mat ,  q = -WA_re^T WZ_re - WA_im^T WZ_im + L1_vec ,  cvxopt.solvers.qp(P, q, -I, h)
// bdrt_gram: one wavefront per 16x16 tile of WA^T WA with v_mfma_f64_16x16x4_f64 (both operands are column
// slices of the same row-major WA, so every operand load is a coalesced 128-byte row segment); q by a wave-shuffle
// reduction per column.  bdrt_qp_box: primal-dual interior point (Mehrotra predictor-corrector) for
// min 1/2 x'Px + q'x s.t. x >= lo, with cvxopt's default tolerances; like cvxopt it returns a strictly interior
// point (slacks never exactly zero), which matters for the hyper-lambda fixed point downstream.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bdrt_host.h"
#include "bdrt_newton.h"   // cholesky_lower / cholesky_solve

namespace bdrt {

// P tile (ti, tj): P[i][j] = sum_r WA[r][i] WA[r][j] (+ L2[i][j]); grid (tiles, tiles), block 64
__global__ __launch_bounds__(64) void gram_kernel(const double *__restrict__ WA, int nrows, int n,
                                                  const double *__restrict__ L2, double *__restrict__ P)
{
    const int lane = threadIdx.x, col = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int ia = i0 + col, jb = j0 + col;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = 0; r0 < nrows; r0 += 4) {
        const int r = r0 + kq;
        const double a = (r < nrows && ia < n) ? WA[(size_t)r * n + ia] : 0.0;   // A[i][k] = WA[k][i]
        const double b = (r < nrows && jb < n) ? WA[(size_t)r * n + jb] : 0.0;   // B[k][j] = WA[k][j]
        acc = mfma_f64(a, b, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + kq + 4 * q, j = j0 + col;
        if (i < n && j < n) P[(size_t)i * n + j] = acc[q] + (L2 ? L2[(size_t)i * n + j] : 0.0);
    }
}

// q[i] = -sum_r WA[r][i] WZ[r] + L1[i]; one wave per 64 columns, rows strided... small: one thread per column
__global__ void gram_q_kernel(const double *__restrict__ WA, const double *__restrict__ WZ, int nrows, int n,
                              const double *__restrict__ L1, double *__restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) s += WA[(size_t)r * n + i] * WZ[r];   // coalesced across i
    q[i] = -s + (L1 ? L1[i] : 0.0);
}

}  // namespace bdrt

using namespace bdrt;

extern "C" {

int bdrt_gram(const double *WA, const double *WZ, int nrows, int n, const double *L2mat, const double *L1vec, double *P,
              double *q)
{
    if (!WA || nrows < 1 || n < 1 || (!P && !q) || (q && !WZ)) { set_error("bdrt_gram: bad arguments"); return -1; }
    bind_process_device();
    double *dWA = nullptr, *dWZ = nullptr, *dL2 = nullptr, *dL1 = nullptr, *dP = nullptr, *dq = nullptr;
    auto cleanup = [&]() { hipFree(dWA); hipFree(dWZ); hipFree(dL2); hipFree(dL1); hipFree(dP); hipFree(dq); };
#define GR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    GR_HIP(hipMalloc((void **)&dWA, (size_t)nrows * n * sizeof(double)));
    GR_HIP(hipMemcpy(dWA, WA, (size_t)nrows * n * sizeof(double), hipMemcpyHostToDevice));
    if (P) {
        GR_HIP(hipMalloc((void **)&dP, (size_t)n * n * sizeof(double)));
        if (L2mat) {
            GR_HIP(hipMalloc((void **)&dL2, (size_t)n * n * sizeof(double)));
            GR_HIP(hipMemcpy(dL2, L2mat, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
        }
        const int tiles = (n + 15) / 16;
        hipLaunchKernelGGL(gram_kernel, dim3(tiles, tiles), dim3(64), 0, 0, dWA, nrows, n, dL2, dP);
        GR_HIP(hipGetLastError());
        GR_HIP(hipMemcpy(P, dP, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (q) {
        GR_HIP(hipMalloc((void **)&dWZ, (size_t)nrows * sizeof(double)));
        GR_HIP(hipMemcpy(dWZ, WZ, (size_t)nrows * sizeof(double), hipMemcpyHostToDevice));
        GR_HIP(hipMalloc((void **)&dq, (size_t)n * sizeof(double)));
        if (L1vec) {
            GR_HIP(hipMalloc((void **)&dL1, (size_t)n * sizeof(double)));
            GR_HIP(hipMemcpy(dL1, L1vec, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(gram_q_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, dWA, dWZ, nrows, n, dL1, dq);
        GR_HIP(hipGetLastError());
        GR_HIP(hipMemcpy(q, dq, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
#undef GR_HIP
    cleanup();
    return 0;
}

// min 1/2 x'Px + q'x  s.t.  x >= lo.  Variables: slack s = x - lo > 0, multiplier z > 0.
//   r_d = P x + q - z = 0 ;  s z = mu.
// Newton step with (P + diag(z/s)) dx = -(r_d) + (sigma mu - s z)/s ... (Mehrotra predictor-corrector).
int bdrt_qp_box(const double *P, const double *q, const double *lo, int n, double *x, double *primal_objective)
{
    if (!P || !q || !x || n < 1) { set_error("bdrt_qp_box: bad arguments"); return -1; }
    const double abstol = 1e-7, reltol = 1e-6, feastol = 1e-7;      // cvxopt.solvers defaults
    std::vector<char> bounded(n);
    std::vector<double> l(n), s(n), z(n), rd(n), rhs(n), dx(n), ds(n), dz(n), Px(n), M((size_t)n * n);
    for (int i = 0; i < n; ++i) { bounded[i] = lo ? std::isfinite(lo[i]) : 0; l[i] = bounded[i] ? lo[i] : 0.0; }
    // start: strictly feasible, unit slacks and multipliers
    for (int i = 0; i < n; ++i) { x[i] = bounded[i] ? l[i] + 1.0 : 0.0; s[i] = 1.0; z[i] = bounded[i] ? 1.0 : 0.0; }
    auto matvec = [&](const std::vector<double> &v, std::vector<double> &out) {
        for (int i = 0; i < n; ++i) {
            double t = 0;
            const double *row = P + (size_t)i * n;
            for (int j = 0; j < n; ++j) t += row[j] * v[j];
            out[i] = t;
        }
    };
    std::vector<double> xv(x, x + n);
    int nb = 0;
    for (int i = 0; i < n; ++i) nb += bounded[i];
    double qnorm = 0;
    for (int i = 0; i < n; ++i) qnorm += q[i] * q[i];
    qnorm = std::max(1.0, std::sqrt(qnorm));
    int it = 0;
    const int max_it = 200;
    for (; it < max_it; ++it) {
        matvec(xv, Px);
        double pobj = 0, gap = 0, rdn = 0;
        for (int i = 0; i < n; ++i) {
            pobj += xv[i] * (0.5 * Px[i] + q[i]);
            rd[i] = Px[i] + q[i] - z[i];
            rdn += rd[i] * rd[i];
            if (bounded[i]) gap += s[i] * z[i];
        }
        rdn = std::sqrt(rdn);
        // dual objective = pobj - gap when r_d = 0
        const double dobj = pobj - gap;
        double relgap = INFINITY;
        if (pobj < 0) relgap = gap / -pobj; else if (dobj > 0) relgap = gap / dobj;
        if (rdn / qnorm <= feastol && (gap <= abstol || relgap <= reltol)) break;
        const double mu = nb ? gap / nb : 0.0;
        // factor M = P + diag(z/s)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) M[(size_t)i * n + j] = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
        for (int i = 0; i < n; ++i) if (bounded[i]) M[(size_t)i * n + i] += z[i] / s[i];
        double reg = 0.0;
        std::vector<double> L = M;
        while (!cholesky_lower(L, n)) {                 // P may be only semi-definite: regularise
            reg = reg == 0.0 ? 1e-14 * (1.0 + std::fabs(M[0])) : reg * 100.0;
            L = M;
            for (int i = 0; i < n; ++i) L[(size_t)i * n + i] += reg;
            if (reg > 1e6) { set_error("bdrt_qp_box: KKT matrix not positive definite"); return -3; }
        }
        auto solve_dir = [&](double sigma_mu, const std::vector<double> *dsa, const std::vector<double> *dza) {
            // complementarity residual rc_i = sigma_mu - s_i z_i (- dsa_i dza_i for the corrector)
            for (int i = 0; i < n; ++i) {
                double rc = 0.0;
                if (bounded[i]) {
                    rc = sigma_mu - s[i] * z[i];
                    if (dsa) rc -= (*dsa)[i] * (*dza)[i];
                }
                rhs[i] = -rd[i] + (bounded[i] ? rc / s[i] : 0.0);
            }
            dx = rhs;
            cholesky_solve(L, n, dx);
            for (int i = 0; i < n; ++i) {
                if (bounded[i]) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (dsa) rc -= (*dsa)[i] * (*dza)[i];
                    ds[i] = dx[i];
                    dz[i] = (rc - z[i] * ds[i]) / s[i];
                } else { ds[i] = 0; dz[i] = 0; }
            }
        };
        auto max_step = [&]() {
            double a = 1.0;
            for (int i = 0; i < n; ++i) {
                if (!bounded[i]) continue;
                if (ds[i] < 0) a = std::min(a, -s[i] / ds[i]);
                if (dz[i] < 0) a = std::min(a, -z[i] / dz[i]);
            }
            return a;
        };
        // predictor (affine scaling)
        solve_dir(0.0, nullptr, nullptr);
        const double a_aff = max_step();
        double gap_aff = 0;
        for (int i = 0; i < n; ++i) if (bounded[i]) gap_aff += (s[i] + a_aff * ds[i]) * (z[i] + a_aff * dz[i]);
        double sigma = nb && gap > 0 ? std::pow(gap_aff / gap, 3.0) : 0.0;
        sigma = std::min(1.0, std::max(0.0, sigma));
        std::vector<double> dsa = ds, dza = dz;
        // corrector
        solve_dir(sigma * mu, &dsa, &dza);
        const double a = std::min(1.0, 0.99 * max_step());
        for (int i = 0; i < n; ++i) {
            xv[i] += a * dx[i];
            if (bounded[i]) { s[i] += a * ds[i]; z[i] += a * dz[i]; s[i] = xv[i] - l[i] > 0 ? xv[i] - l[i] : s[i]; }
        }
    }
    memcpy(x, xv.data(), sizeof(double) * n);
    if (primal_objective) {
        matvec(xv, Px);
        double pobj = 0;
        for (int i = 0; i < n; ++i) pobj += xv[i] * (0.5 * Px[i] + q[i]);
        *primal_objective = pobj;
    }
    if (it >= max_it) { set_error("bdrt_qp_box: iteration limit"); return -4; }
    return it;
}

}  // extern "C"
