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

// min 1/2 x'Px + q'x  s.t.  x >= lo  -- the QP of Inverter._convex_opt (reference inversion.py:1043-1067), solved the way
// the reference's solver does.  cvxopt (setup.py:20, unpinned; not under /root/reference) documents `solvers.qp` as its
// `coneqp` path-following method: Mehrotra predictor-corrector with Nesterov-Todd scaling, which for the nonnegative
// orthant (G = -I, h = -lo) is the standard primal-dual Newton system.  Restated here in unscaled variables:
//   start   (P + diag(b)) x = -q + b lo  (b = 1 on bounded variables);  s = x - lo, z = -s; each shifted into the interior by
//           1 + max(-s) resp. 1 + max(-z) when not already interior (tolerance 1e-8 max(1, |.|))
//   iterate r_x = P x + q - z,  r_z = s - (x - lo);  stop when |r_z| / max(1,|lo|) <= feastol, |r_x| / max(1,|q|) <= feastol
//           and (gap <= abstol or gap / |cost| <= reltol), gap = s'z, at most 100 iterations;
//           affine direction (sigma = 0):   (P + diag(z/s)) dx = -r_x + (r_c + z r_z) / s,  ds = dx - r_z,  dz = (r_c - z ds) / s
//           with r_c = -s z;  step t = min(1, 1 / max(-ds/s, -dz/z));  sigma = clip(1 - t + t^2 ds'dz / gap, 0, 1)^3;
//           corrector  r_c = -s z - ds_a dz_a + sigma gap / m;  step = min(1, 0.99 / max(-ds/s, -dz/z)).
// The ridge problems are nearly flat (objective changes of 1e-7 relative move coefficients by 10 %), so the point on the
// central path where the iteration stops is part of the reference's answer: tests/test_ridge_reference.py checks this
// solver against solutions and objectives cvxopt itself produced (stored in the reference's pickled fits).
int bdrt_qp_box(const double *P, const double *q, const double *lo, int n, double *x, double *primal_objective)
{
    if (!P || !q || !x || n < 1) { set_error("bdrt_qp_box: bad arguments"); return -1; }
    const double abstol = 1e-7, reltol = 1e-6, feastol = 1e-7;      // cvxopt.solvers.options defaults
    const int max_it = 100;                                          // cvxopt 'maxiters'
    std::vector<char> bounded(n);
    std::vector<double> l(n), s(n, 0.0), z(n, 0.0), rx(n), rz(n, 0.0), rhs(n), dx(n), ds(n, 0.0), dz(n, 0.0), dsa(n, 0.0), dza(n, 0.0),
        Px(n), M((size_t)n * n), xv(n);
    int nb = 0;
    double qq = 0.0, hh = 0.0;
    for (int i = 0; i < n; ++i) {
        bounded[i] = lo ? std::isfinite(lo[i]) : 0;
        l[i] = bounded[i] ? lo[i] : 0.0;
        nb += bounded[i];
        qq += q[i] * q[i];
        hh += l[i] * l[i];
    }
    const double resx0 = std::max(1.0, std::sqrt(qq)), resz0 = std::max(1.0, std::sqrt(hh));
    auto matvec = [&](const std::vector<double> &v, std::vector<double> &out) {
        for (int i = 0; i < n; ++i) {
            double t = 0;
            const double *row = P + (size_t)i * n;
            for (int j = 0; j < n; ++j) t += row[j] * v[j];
            out[i] = t;
        }
    };
    // factor sym(P) + diag(dg) (+ reg I when only semi-definite); L receives the factor
    std::vector<double> L;
    auto factor = [&](const std::vector<double> &dg) -> bool {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) M[(size_t)i * n + j] = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
        for (int i = 0; i < n; ++i) M[(size_t)i * n + i] += dg[i];
        double reg = 0.0;
        L = M;
        while (!cholesky_lower(L, n)) {
            reg = reg == 0.0 ? 1e-14 * (1.0 + std::fabs(M[0])) : reg * 100.0;
            L = M;
            for (int i = 0; i < n; ++i) L[(size_t)i * n + i] += reg;
            if (!(reg <= 1e6)) return false;                // (also on NaN)
        }
        return true;
    };
    // ---- starting point ----
    {
        std::vector<double> dg(n);
        for (int i = 0; i < n; ++i) { dg[i] = bounded[i] ? 1.0 : 0.0; xv[i] = -q[i] + (bounded[i] ? l[i] : 0.0); }
        if (!factor(dg)) { set_error("bdrt_qp_box: KKT matrix not positive definite"); return -3; }
        cholesky_solve(L, n, xv);
        double nrms = 0.0, ts = -INFINITY, tz = -INFINITY;
        for (int i = 0; i < n; ++i)
            if (bounded[i]) {
                s[i] = xv[i] - l[i]; z[i] = -s[i];
                nrms += s[i] * s[i];
                ts = std::max(ts, -s[i]); tz = std::max(tz, -z[i]);
            }
        nrms = std::sqrt(nrms);                                     // |s| = |z|
        if (nb && ts >= -1e-8 * std::max(nrms, 1.0)) for (int i = 0; i < n; ++i) if (bounded[i]) s[i] += 1.0 + ts;
        if (nb && tz >= -1e-8 * std::max(nrms, 1.0)) for (int i = 0; i < n; ++i) if (bounded[i]) z[i] += 1.0 + tz;
    }
    int it = 0, status = 0;
    for (;; ++it) {
        matvec(xv, Px);
        double f0a = 0, f0b = 0, gap = 0, resx = 0, resz = 0, zrz = 0;
        for (int i = 0; i < n; ++i) {
            const double r0 = Px[i] + q[i];
            f0a += xv[i] * r0; f0b += xv[i] * q[i];
            rx[i] = r0 - (bounded[i] ? z[i] : 0.0);
            resx += rx[i] * rx[i];
            if (bounded[i]) {
                rz[i] = s[i] - (xv[i] - l[i]);
                resz += rz[i] * rz[i];
                zrz += z[i] * rz[i];
                gap += s[i] * z[i];
            }
        }
        resx = std::sqrt(resx); resz = std::sqrt(resz);
        const double pcost = 0.5 * (f0a + f0b), dcost = pcost + zrz - gap;
        double relgap = INFINITY;
        if (pcost < 0) relgap = gap / -pcost; else if (dcost > 0) relgap = gap / dcost;
        if (resz / resz0 <= feastol && resx / resx0 <= feastol && (gap <= abstol || relgap <= reltol)) break;
        if (it == max_it) { status = -4; break; }
        std::vector<double> dg(n);
        for (int i = 0; i < n; ++i) dg[i] = bounded[i] ? z[i] / s[i] : 0.0;
        if (!factor(dg)) { set_error("bdrt_qp_box: KKT matrix not positive definite"); return -3; }
        const double mu = nb ? gap / nb : 0.0;
        auto solve_dir = [&](double sigma_mu, bool corrector) {
            for (int i = 0; i < n; ++i) {
                double t = 0.0;
                if (bounded[i]) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                    t = (rc + z[i] * rz[i]) / s[i];
                }
                rhs[i] = -rx[i] + t;
            }
            dx = rhs;
            cholesky_solve(L, n, dx);
            for (int i = 0; i < n; ++i)
                if (bounded[i]) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                    ds[i] = dx[i] - rz[i];
                    dz[i] = (rc - z[i] * ds[i]) / s[i];
                }
        };
        auto boundary = [&]() {                                     // t = max(0, max -ds/s, max -dz/z)
            double t = 0.0;
            for (int i = 0; i < n; ++i)
                if (bounded[i]) { t = std::max(t, -ds[i] / s[i]); t = std::max(t, -dz[i] / z[i]); }
            return t;
        };
        solve_dir(0.0, false);
        double dsdz = 0.0;
        for (int i = 0; i < n; ++i) if (bounded[i]) dsdz += ds[i] * dz[i];
        double t = boundary();
        double step = t == 0.0 ? 1.0 : std::min(1.0, 1.0 / t);
        double sigma = gap > 0 ? 1.0 - step + dsdz / gap * step * step : 0.0;
        sigma = std::min(1.0, std::max(0.0, sigma));
        sigma = sigma * sigma * sigma;
        dsa = ds; dza = dz;
        solve_dir(sigma * mu, true);
        t = boundary();
        step = t == 0.0 ? 1.0 : std::min(1.0, 0.99 / t);
        for (int i = 0; i < n; ++i) {
            xv[i] += step * dx[i];
            if (bounded[i]) { s[i] += step * ds[i]; z[i] += step * dz[i]; }
        }
    }
    memcpy(x, xv.data(), sizeof(double) * n);
    if (primal_objective) {
        matvec(xv, Px);
        double pobj = 0;
        for (int i = 0; i < n; ++i) pobj += xv[i] * (0.5 * Px[i] + q[i]);
        *primal_objective = pobj;
    }
    if (status == -4) { set_error("bdrt_qp_box: iteration limit"); return -4; }
    return it;
}

}  // extern "C"
