// bdrt_qp.hip -- batched box-constrained QP on the GPU: the ridge / hyper-lambda ridge solve (include/bdrt.h section (3)).
//
// Replaces cvxopt.solvers.qp inside Inverter._convex_opt (reference bayes_drt/inversion.py:1043-1067), which the
// reference calls once per hyper-lambda iteration and 2 x len(lambdas) x iterations times inside ridge_ReImCV (:902-945):
//     min 1/2 x'Px + q'x   s.t.  x >= lo        (lo[i] = -inf: free variable)
// Same algorithm as the host solver bdrt_qp_box (bdrt_ridge.hip, which documents it): cvxopt's coneqp path-following
// method (Mehrotra predictor-corrector, cvxopt's starting point, residual handling, step rules and default tolerances
// abstol 1e-7 / reltol 1e-6 / feastol 1e-7, 100 iterations) specialised to G = -I.  The ridge problems are nearly flat, so
// where on the central path the iteration stops is part of the reference's answer: the solver reproduces solutions cvxopt
// itself computed (stored in the reference's pickled fits) to 1e-8 (tests/test_ridge_reference.py).
//
// One workgroup (512 threads) per problem; problems of a batch run concurrently (Re-Im cross-validation: 62 fits).  The
// KKT matrix lives in LDS as a column-major packed lower triangle (n <= 200: 157 KiB) -- or in a global work buffer for
// larger n -- and is factored by a right-looking Cholesky (two barriers per column; the trailing update is a 32 x 16
// thread tiling with conflict-free column walks).  The two triangular solves of a Newton direction are sequential by
// nature: one wavefront does them with wave-level synchronisation only (a barrier per unknown would cost more than the
// arithmetic).
#include <cmath>
#include <cstring>
#include <vector>

#include "bdrt_host.h"

namespace bdrt {

constexpr int QP_NT = 512;
constexpr int QP_NVEC = 14;          // x, s, z, rd, rhs, dx, ds, dz, dsa, dza, Px, lo, bounded, rz

__device__ __forceinline__ void qp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum64(double x)
{
    x += __shfl_xor(x, 32); x += __shfl_xor(x, 16); x += __shfl_xor(x, 8);
    x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1);
    return x;
}

// sums of up to 3 per-thread values over the workgroup (deterministic order); result to all threads
__device__ inline void block_sum3(double &a, double &b, double &c, double *red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    a = wave_sum64(a); b = wave_sum64(b); c = wave_sum64(c);
    __syncthreads();                       // red may still be read from the previous call
    if (lane == 0) { red[wave] = a; red[8 + wave] = b; red[16 + wave] = c; }
    __syncthreads();
    double sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { sa += red[w]; sb += red[8 + w]; sc += red[16 + w]; }
    a = sa; b = sb; c = sc;
}

__device__ inline double block_min(double v, double *red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmin(m, red[w]);
    return m;
}

// column-major packed lower triangle: element (i, j), i >= j
__device__ __forceinline__ size_t cidx(int i, int j, int n) { return (size_t)j * n - (size_t)j * (j - 1) / 2 + (i - j); }

// In-place Cholesky of the packed lower triangle; diag[j] receives L(j,j) (M(j,j) keeps the pivot).  Returns false
// (uniformly) when a pivot is not positive.  All threads of the workgroup must call.
__device__ inline bool chol_packed(double *M, double *diag, int n)
{
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    for (int j = 0; j < n; ++j) {
        const double d = M[cidx(j, j, n)];
        if (!(d > 0.0) || !isfinite(d)) return false;                 // same value in every thread
        const double dj = sqrt(d), inv = 1.0 / dj;
        double *colj = M + cidx(j, j, n);                             // colj[i - j] = M(i, j)
        for (int i = j + 1 + tid; i < n; i += QP_NT) colj[i - j] *= inv;
        if (tid == 0) diag[j] = dj;
        __syncthreads();
        for (int k = j + 1 + ty; k < n; k += QP_NT / 32) {
            const double lkj = colj[k - j];
            double *colk = M + cidx(k, k, n);
            for (int i = k + tx; i < n; i += 32) colk[i - k] -= colj[i - j] * lkj;
        }
        __syncthreads();
    }
    return true;
}

// v <- (L L^T)^-1 v by wavefront 0 (L strictly-lower part in M, diagonal in diag); ends with a workgroup barrier
__device__ inline void chol_solve_wave(const double *M, const double *diag, int n, double *v)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
        for (int j = 0; j < n; ++j) {                                 // forward: L y = v (column oriented)
            qp_wave_sync();
            const double yj = v[j] / diag[j];
            qp_wave_sync();
            if (lane == 0) v[j] = yj;
            const double *colj = M + cidx(j, j, n);
            for (int i = j + 1 + lane; i < n; i += 64) v[i] -= colj[i - j] * yj;
        }
        for (int j = n - 1; j >= 0; --j) {                            // backward: L^T x = y (dot with column j)
            qp_wave_sync();
            const double *colj = M + cidx(j, j, n);
            double t = 0.0;
            for (int i = j + 1 + lane; i < n; i += 64) t += colj[i - j] * v[i];
            t = wave_sum64(t);
            const double xj = (v[j] - t) / diag[j];
            qp_wave_sync();
            if (lane == 0) v[j] = xj;
        }
    }
    __syncthreads();
}

// One box-constrained QP by the calling workgroup: P [n x n] and q [n] in global memory, the solution is left in sh[0..n)
// (LDS) and, when xout != nullptr, copied there.  sh: (QP_NVEC + 1) * np + 32 doubles of LDS followed (LDSM) by the packed
// KKT triangle; Mwork: the triangle when it lives in a global work buffer.  Returns the iteration count, -3 (KKT matrix not
// positive definite) or -4 (iteration limit: the last iterate is left in place, as cvxopt does with status 'unknown');
// *pobj_out receives the primal objective.  All threads must call.
// Algorithm: cvxopt's coneqp path-following method for G = -I, h = -lo, restated in unscaled variables -- the same
// statement as the host solver bdrt_qp_box (bdrt_ridge.hip), which documents it.
template <bool LDSM>
__device__ inline int qp_box_solve(const double *__restrict__ P, const double *__restrict__ q, const double *__restrict__ lo,
                                   int n, double *sh, double *Mwork, double *xout, double *pobj_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = (n + 1) & ~1;
    double *x = sh, *s = x + np, *z = s + np, *rd = z + np, *rhs = rd + np, *dx = rhs + np, *ds = dx + np, *dz = ds + np,
           *dsa = dz + np, *dza = dsa + np, *Px = dza + np, *lv = Px + np, *bd = lv + np, *rz = bd + np, *diag = rz + np,
           *red = diag + np;
    double *M = LDSM ? red + 32 : Mwork;
    const double abstol = 1e-7, reltol = 1e-6, feastol = 1e-7;      // cvxopt.solvers.options defaults
    const int max_it = 100;                                          // cvxopt 'maxiters'

    double nbv = 0.0, qq = 0.0, hh = 0.0;
    for (int i = tid; i < n; i += QP_NT) {
        const double li = lo ? lo[i] : -INFINITY;
        const bool bnd = isfinite(li);
        bd[i] = bnd ? 1.0 : 0.0;
        lv[i] = bnd ? li : 0.0;
        s[i] = 0.0; z[i] = 0.0; rz[i] = 0.0; ds[i] = 0.0; dz[i] = 0.0; dsa[i] = 0.0; dza[i] = 0.0;
        nbv += bnd ? 1.0 : 0.0;
        qq += q[i] * q[i];
        hh += bnd ? li * li : 0.0;
    }
    block_sum3(nbv, qq, hh, red);
    const double resx0 = fmax(1.0, sqrt(qq)), resz0 = fmax(1.0, sqrt(hh));
    const int nb = (int)nbv;

    auto matvec = [&]() {                                             // Px = P x : one wavefront per row
        __syncthreads();
        for (int i = wave; i < n; i += QP_NT / 64) {
            const double *row = P + (size_t)i * n;
            double t = 0.0;
            for (int j = lane; j < n; j += 64) t += row[j] * x[j];
            t = wave_sum64(t);
            if (lane == 0) Px[i] = t;
        }
        __syncthreads();
    };
    // factor sym(P) + diag(dgv[i]) (+ reg I when only semi-definite); dgv: unit = 1 on bounded variables, else z/s
    auto factor = [&](bool unit) -> bool {
        double reg = 0.0;
        bool ok = false;
        while (true) {
            __syncthreads();
            for (int j = wave; j < n; j += QP_NT / 64) {              // one wavefront per column of the lower triangle
                double *colj = M + cidx(j, j, n);
                for (int i = j + lane; i < n; i += 64) {
                    double v = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
                    if (i == j) v += (bd[i] != 0.0 ? (unit ? 1.0 : z[i] / s[i]) : 0.0) + reg;
                    colj[i - j] = v;
                }
            }
            __syncthreads();
            const double m00 = M[0] - reg;
            ok = chol_packed(M, diag, n);
            if (ok) break;
            reg = reg == 0.0 ? 1e-14 * (1.0 + fabs(m00)) : reg * 100.0;
            if (!(reg <= 1e6)) break;                                  // (also leaves on NaN: non-finite input must not spin here)
        }
        return ok;
    };

    // ---- starting point: (P + diag(b)) x = -q + b lo; s = x - lo, z = -s, both shifted into the interior ----
    if (!factor(true)) { *pobj_out = 0.0; return -3; }
    for (int i = tid; i < n; i += QP_NT) x[i] = -q[i] + (bd[i] != 0.0 ? lv[i] : 0.0);
    __syncthreads();
    chol_solve_wave(M, diag, n, x);
    {
        double nrm2 = 0.0, mn = INFINITY, mx = INFINITY, dummy = 0.0, dummy2 = 0.0;
        for (int i = tid; i < n; i += QP_NT)
            if (bd[i] != 0.0) {
                const double si = x[i] - lv[i];
                s[i] = si; z[i] = -si;
                nrm2 += si * si;
                mn = fmin(mn, si);                                    // ts = max(-s) = -min(s)
                mx = fmin(mx, -si);                                   // tz = max(-z) = -min(z) = -min(-s)
            }
        block_sum3(nrm2, dummy, dummy2, red);
        const double ts = -block_min(mn, red), tz = -block_min(mx, red);
        const double nrm = fmax(sqrt(nrm2), 1.0);
        for (int i = tid; i < n; i += QP_NT)
            if (bd[i] != 0.0) {
                if (nb && ts >= -1e-8 * nrm) s[i] += 1.0 + ts;
                if (nb && tz >= -1e-8 * nrm) z[i] += 1.0 + tz;
            }
        __syncthreads();
    }

    int it = 0, status = 0;
    for (;; ++it) {
        matvec();
        double f0 = 0.0, gap = 0.0, resx = 0.0, resz = 0.0, zrz = 0.0, dummy = 0.0;
        for (int i = tid; i < n; i += QP_NT) {
            const double r0 = Px[i] + q[i];
            f0 += x[i] * r0 + x[i] * q[i];
            const double r = r0 - (bd[i] != 0.0 ? z[i] : 0.0);
            rd[i] = r;
            resx += r * r;
            if (bd[i] != 0.0) {
                const double rzi = s[i] - (x[i] - lv[i]);
                rz[i] = rzi;
                resz += rzi * rzi;
                zrz += z[i] * rzi;
                gap += s[i] * z[i];
            }
        }
        block_sum3(f0, gap, resx, red);
        block_sum3(resz, zrz, dummy, red);
        resx = sqrt(resx); resz = sqrt(resz);
        const double pcost = 0.5 * f0, dcost = pcost + zrz - gap;
        double relgap = INFINITY;
        if (pcost < 0) relgap = gap / -pcost; else if (dcost > 0) relgap = gap / dcost;
        if (resz / resz0 <= feastol && resx / resx0 <= feastol && (gap <= abstol || relgap <= reltol)) break;
        if (it == max_it) { status = -4; break; }
        const double mu = nb ? gap / nb : 0.0;
        if (!factor(false)) { status = -3; break; }

        auto solve_dir = [&](double sigma_mu, bool corrector) {
            for (int i = tid; i < n; i += QP_NT) {
                double t = 0.0;
                if (bd[i] != 0.0) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                    t = (rc + z[i] * rz[i]) / s[i];
                }
                rhs[i] = -rd[i] + t;
                dx[i] = rhs[i];
            }
            __syncthreads();
            chol_solve_wave(M, diag, n, dx);
            for (int i = tid; i < n; i += QP_NT) {
                if (bd[i] != 0.0) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                    ds[i] = dx[i] - rz[i];
                    dz[i] = (rc - z[i] * ds[i]) / s[i];
                } else { ds[i] = 0.0; dz[i] = 0.0; }
            }
            __syncthreads();
        };
        auto boundary = [&]() {                                       // t = max(0, max -ds/s, max -dz/z)
            double t = 0.0;
            for (int i = tid; i < n; i += QP_NT) {
                if (bd[i] == 0.0) continue;
                t = fmax(t, -ds[i] / s[i]);
                t = fmax(t, -dz[i] / z[i]);
            }
            return -block_min(-t, red);
        };
        solve_dir(0.0, false);                                        // predictor (affine scaling)
        double dsdz = 0.0, d1 = 0.0, d2 = 0.0;
        for (int i = tid; i < n; i += QP_NT)
            if (bd[i] != 0.0) dsdz += ds[i] * dz[i];
        block_sum3(dsdz, d1, d2, red);
        double t = boundary();
        double step = t == 0.0 ? 1.0 : fmin(1.0, 1.0 / t);
        double sigma = gap > 0 ? 1.0 - step + dsdz / gap * step * step : 0.0;
        sigma = fmin(1.0, fmax(0.0, sigma));
        sigma = sigma * sigma * sigma;
        for (int i = tid; i < n; i += QP_NT) { dsa[i] = ds[i]; dza[i] = dz[i]; }
        __syncthreads();
        solve_dir(sigma * mu, true);                                  // corrector
        t = boundary();
        step = t == 0.0 ? 1.0 : fmin(1.0, 0.99 / t);
        for (int i = tid; i < n; i += QP_NT) {
            x[i] += step * dx[i];
            if (bd[i] != 0.0) { s[i] += step * ds[i]; z[i] += step * dz[i]; }
        }
        __syncthreads();
    }
    matvec();
    double pobj = 0.0, d1 = 0.0, d2 = 0.0;
    for (int i = tid; i < n; i += QP_NT) {
        pobj += x[i] * (0.5 * Px[i] + q[i]);
        if (xout) xout[i] = x[i];
    }
    block_sum3(pobj, d1, d2, red);
    *pobj_out = pobj;
    __syncthreads();
    return status < 0 ? status : it;
}

template <bool LDSM>
__global__ __launch_bounds__(QP_NT) void qp_box_kernel(const double *__restrict__ Pall, const double *__restrict__ qall,
                                                       const double *__restrict__ lo, int n, double *__restrict__ Xall,
                                                       double *__restrict__ objall, int *__restrict__ itall,
                                                       double *__restrict__ work)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = blockIdx.x;
    const size_t msize = (size_t)n * (n + 1) / 2;
    double pobj;
    const int its = qp_box_solve<LDSM>(Pall + (size_t)b * n * n, qall + (size_t)b * n, lo, n, sh, LDSM ? nullptr : work + (size_t)b * msize,
                                       Xall + (size_t)b * n, &pobj);
    if (threadIdx.x == 0) { objall[b] = pobj; itall[b] = its; }
}

// ---- hyper-lambda ridge: the whole outer loop of Inverter.ridge_fit on the device ---------------------------------------------
// (reference bayes_drt/inversion.py:518-740; lambda updates :947-983; see include/bdrt.h bdrt_ridge for the arithmetic)
struct RidgeArgs {
    int n, K, off, penalty, max_iter, hyper_lambda, zero_delta1;
    double xtol, hl_fbeta;
    double reg_ord[3];
    const double *G, *qbase;     // [ng][n][n], [ng][n]
    const int *gsel;             // [nb]
    const double *base;          // [3][n][n]
    const double *Ls;            // [3][K][n] (discrete penalty)
    const double *lo;            // [n]
    const double *lambda0;       // [nb]       start value of every lambda vector
    const double *lam0s, *betas; // [nb][3]    prior rate / shape terms of the lambda update (see bdrt.h)
    const double *x0;            // [nb][n] or nullptr
    double *Pwork;               // [nb][n][n]
    double *Mwork;               // [nb][n(n+1)/2] when the KKT triangle does not fit in LDS
    double *coef, *lam, *cost, *fun;
    int *iters, *flags;
    double *hist_coef, *hist_lam, *hist_fun, *hist_cost;
};

template <bool LDSM>
__global__ __launch_bounds__(QP_NT) void ridge_kernel(RidgeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, K = a.K, off = a.off;
    const int np = (n + 1) & ~1;
    // LDS: [QP vectors + reduction scratch (+ KKT triangle) | coef | prev | lam[3] | tmp | red]
    const size_t qp_doubles = (size_t)(QP_NVEC + 1) * np + 32 + (LDSM ? (size_t)n * (n + 1) / 2 : 0);
    double *coef = sh + ((qp_doubles + 1) & ~(size_t)1), *prev = coef + np, *lam = prev + np, *tmp = lam + 3 * np, *red = tmp + np;
    const double *G = a.G + (size_t)a.gsel[b] * n * n, *q = a.qbase + (size_t)a.gsel[b] * n;
    double *P = a.Pwork + (size_t)b * n * n;
    double *Mw = LDSM ? nullptr : a.Mwork + (size_t)b * ((size_t)n * (n + 1) / 2);
    const double l0 = a.lambda0[b];
    for (int i = tid; i < n; i += QP_NT) {
        coef[i] = a.x0 ? a.x0[(size_t)b * n + i] : 1e-6;
        lam[i] = l0; lam[np + i] = l0; lam[2 * np + i] = l0;
    }
    __syncthreads();
    int it = 0, flag = 0, qp_its = 0;
    double cost = 0.0, fun = 0.0;
    const int iters_max = a.hyper_lambda ? a.max_iter : 1;
    while (it < iters_max) {
        for (int i = tid; i < n; i += QP_NT) prev[i] = coef[i];
        __syncthreads();
        if (a.hyper_lambda) {
            for (int o = 0; o < 3; ++o) {
                if (!(a.reg_ord[o] > 0.0)) continue;
                const double *Mb = a.base + (size_t)o * n * n;
                double *lv = lam + o * np;
                const double beta = a.betas[b * 3 + o], lam0 = a.lam0s[b * 3 + o];
                if (a.penalty == 0) {
                    // discrete: (L x)^2 per row of L_o, then the closed-form lambda (inversion.py:947-964)
                    const double *L = a.Ls + (size_t)o * K * n;
                    for (int r = wave; r < K; r += QP_NT / 64) {
                        const double *row = L + (size_t)r * n;
                        double t = 0.0;
                        for (int c = lane; c < n; c += 64) t += row[c] * prev[c];
                        t = wave_sum64(t);
                        if (lane == 0) tmp[r] = t * t;
                    }
                    __syncthreads();
                    if (a.hl_fbeta > 0.0) {
                        double mx = 0.0;
                        for (int r = tid; r < K; r += QP_NT) mx = fmax(mx, tmp[r]);
                        mx = -block_min(-mx, red);
                        for (int r = tid; r < K; r += QP_NT) lv[off + r] = l0 / (tmp[r] / (mx * a.hl_fbeta) + 1.0);
                    } else {
                        for (int r = tid; r < K; r += QP_NT) lv[off + r] = 1.0 / (tmp[r] / (beta - 1.0) + 1.0 / lam0);
                    }
                    for (int i = tid; i < off; i += QP_NT) lv[i] = 1.0;
                    __syncthreads();
                } else {
                    // integral: C_j = sum_{r != j} (c_r sqrt(lam_r)) M_rj c_j with c = factor * x, then the positive root
                    // (inversion.py:973-983)
                    const double factor = o == 0 ? 100.0 : (o == 1 ? 10.0 : 1.0);
                    for (int j = wave; j < n; j += QP_NT / 64) {
                        const double cj = factor * prev[j];
                        double t = 0.0;
                        for (int r = lane; r < n; r += 64)
                            if (r != j) t += ((factor * prev[r]) * sqrt(lv[r])) * Mb[(size_t)r * n + j] * cj;
                        t = wave_sum64(t);
                        if (lane == 0) tmp[j] = t;
                    }
                    __syncthreads();
                    for (int j = tid; j < n; j += QP_NT) {
                        const double c = factor * prev[j], Cv = tmp[j];
                        const double aa = beta / 2.0, bb = 0.5 * (2.0 * aa - 2.0) / lam0;
                        const double d = c * c * Mb[(size_t)j * n + j] + 2.0 * bb;
                        const double sg = Cv > 0.0 ? 1.0 : (Cv < 0.0 ? -1.0 : 0.0);
                        double l = (Cv * Cv - sg * Cv * sqrt(4.0 * d * (2.0 * aa - 2.0) + Cv * Cv) + 2.0 * d * (2.0 * aa - 2.0)) / (2.0 * d * d);
                        if (l <= 0.0) l = 1e-15;
                        lv[j] = l;
                    }
                    __syncthreads();
                }
            }
        }
        // P = G + sum_o frac_o * (sqrt(lam_o) M_o sqrt(lam_o))
        for (size_t e = tid; e < (size_t)n * n; e += QP_NT) {
            const int r = (int)(e / n), c = (int)(e - (size_t)r * n);
            double l2 = 0.0;
            for (int o = 0; o < 3; ++o)
                if (a.reg_ord[o] > 0.0) l2 += a.reg_ord[o] * ((sqrt(lam[o * np + r]) * a.base[(size_t)o * n * n + e]) * sqrt(lam[o * np + c]));
            P[e] = G[e] + l2;
        }
        __syncthreads();
        qp_its = qp_box_solve<LDSM>(P, q, a.lo, n, sh, Mw, nullptr, &fun);
        if (qp_its == -3) { flag = -3; break; }
        for (int i = tid; i < n; i += QP_NT) coef[i] = sh[i];
        __syncthreads();
        // cost = 1/2 x.P x + q.x and the relative change of the coefficients
        for (int i = wave; i < n; i += QP_NT / 64) {
            const double *row = P + (size_t)i * n;
            double t = 0.0;
            for (int j = lane; j < n; j += 64) t += row[j] * coef[j];
            t = wave_sum64(t);
            if (lane == 0) tmp[i] = t;
        }
        __syncthreads();
        double c1 = 0.0, dsum = 0.0, dnan = 0.0;
        for (int i = tid; i < n; i += QP_NT) {
            c1 += coef[i] * (0.5 * tmp[i] + q[i]);
            double dl = (coef[i] - prev[i]) / prev[i];
            if (i == 1 && a.gsel[b] < 31 && ((a.zero_delta1 >> a.gsel[b]) & 1)) dl = 0.0;     // (a property of the fit's data part)
            dl = fabs(dl);
            if (dl != dl) dnan += 1.0; else dsum += dl;
        }
        block_sum3(c1, dsum, dnan, red);
        cost = c1;
        if (a.hist_coef) {
            for (int i = tid; i < n; i += QP_NT) {
                a.hist_coef[((size_t)b * a.max_iter + it) * n + i] = coef[i];
                for (int o = 0; o < 3; ++o) a.hist_lam[(((size_t)b * a.max_iter + it) * 3 + o) * n + i] = lam[o * np + i];
            }
            if (tid == 0) { a.hist_fun[(size_t)b * a.max_iter + it] = fun; a.hist_cost[(size_t)b * a.max_iter + it] = cost; }
        }
        if (qp_its == -4) flag |= 4;                    // a QP hit its iteration limit: last iterate used (cvxopt: status "unknown")
        ++it;
        if (!a.hyper_lambda) break;
        // np.mean(np.abs(delta)) < xtol: a NaN (0/0) makes the comparison false, an inf makes the mean inf
        if (dnan == 0.0 && dsum / (double)n < a.xtol) { flag |= 1; break; }
    }
    for (int i = tid; i < n; i += QP_NT) {
        a.coef[(size_t)b * n + i] = coef[i];
        for (int o = 0; o < 3; ++o) a.lam[((size_t)b * 3 + o) * n + i] = lam[o * np + i];
    }
    if (tid == 0) { a.cost[b] = cost; a.fun[b] = fun; a.iters[b] = it; a.flags[b] = flag; }
}

}  // namespace bdrt

using namespace bdrt;

extern "C" {

int bdrt_qp_box_batch(const double *P, const double *q, const double *lo, int n, int nb, double *x,
                      double *primal_objective, int *iterations)
{
    if (!P || !q || !x || n < 1 || nb < 1) { set_error("bdrt_qp_box_batch: bad arguments"); return -1; }
    bind_process_device();
    const int np = (n + 1) & ~1;
    const size_t vec_bytes = ((size_t)(QP_NVEC + 1) * np + 32) * sizeof(double);      // + diag, reduction scratch
    const size_t msize = (size_t)n * (n + 1) / 2;
    const bool in_lds = vec_bytes + msize * sizeof(double) <= 160 * 1024;
    const size_t lds = in_lds ? vec_bytes + msize * sizeof(double) : vec_bytes;
    if (lds > 160 * 1024) { set_error("bdrt_qp_box_batch: n = %d too large", n); return -2; }
    double *dP = nullptr, *dq = nullptr, *dlo = nullptr, *dx = nullptr, *dobj = nullptr, *dwork = nullptr;
    int *dit = nullptr;
    auto cleanup = [&]() { hipFree(dP); hipFree(dq); hipFree(dlo); hipFree(dx); hipFree(dobj); hipFree(dwork); hipFree(dit); };
#define QP_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    QP_HIP(hipMalloc((void **)&dP, (size_t)nb * n * n * sizeof(double)));
    QP_HIP(hipMemcpy(dP, P, (size_t)nb * n * n * sizeof(double), hipMemcpyHostToDevice));
    QP_HIP(hipMalloc((void **)&dq, (size_t)nb * n * sizeof(double)));
    QP_HIP(hipMemcpy(dq, q, (size_t)nb * n * sizeof(double), hipMemcpyHostToDevice));
    if (lo) {
        QP_HIP(hipMalloc((void **)&dlo, (size_t)n * sizeof(double)));
        QP_HIP(hipMemcpy(dlo, lo, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
    QP_HIP(hipMalloc((void **)&dx, (size_t)nb * n * sizeof(double)));
    QP_HIP(hipMalloc((void **)&dobj, (size_t)nb * sizeof(double)));
    QP_HIP(hipMalloc((void **)&dit, (size_t)nb * sizeof(int)));
    if (!in_lds) QP_HIP(hipMalloc((void **)&dwork, (size_t)nb * msize * sizeof(double)));
    static LdsAttrCache attr_cache;
    QP_HIP(attr_cache.ensure(lds, [&]() {
        hipError_t e = hipFuncSetAttribute((const void *)qp_box_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)qp_box_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return e;
    }));
    if (in_lds)
        hipLaunchKernelGGL(qp_box_kernel<true>, dim3(nb), dim3(QP_NT), lds, 0, dP, dq, dlo, n, dx, dobj, dit, dwork);
    else
        hipLaunchKernelGGL(qp_box_kernel<false>, dim3(nb), dim3(QP_NT), lds, 0, dP, dq, dlo, n, dx, dobj, dit, dwork);
    QP_HIP(hipGetLastError());
    QP_HIP(hipDeviceSynchronize());
    QP_HIP(hipMemcpy(x, dx, (size_t)nb * n * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int> its(nb);
    QP_HIP(hipMemcpy(its.data(), dit, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    if (primal_objective) QP_HIP(hipMemcpy(primal_objective, dobj, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost));
#undef QP_HIP
    cleanup();
    int worst = 0;
    for (int b = 0; b < nb; ++b) {
        if (iterations) iterations[b] = its[b];
        if (its[b] == -3) worst = -3;                              // a non-PD problem outranks an iteration limit elsewhere
        else if (its[b] < 0 && worst == 0) worst = its[b];
    }
    if (worst == -3) { set_error("bdrt_qp_box_batch: KKT matrix not positive definite"); return -3; }
    if (worst == -4) { set_error("bdrt_qp_box_batch: iteration limit"); return -4; }
    return 0;
}


int bdrt_ridge(const bdrt_ridge_options *opt, int nb, int ng, const double *G, const double *qbase, const int *gsel,
               const double *base, const double *Ls, const double *lo, const double *lambda0, const double *lam0s,
               const double *betas, const double *x0, double *coef, double *lam, double *cost, double *fun, int *iters,
               int *flags, double *hist_coef, double *hist_lam, double *hist_fun, double *hist_cost)
{
    if (!opt || nb < 1 || ng < 1 || !G || !qbase || !gsel || !base || !lambda0 || !lam0s || !betas || !coef || opt->n < 1 ||
        opt->max_iter < 1 || (opt->penalty == 0 && opt->hyper_lambda && (!Ls || opt->K < 1)) || opt->off < 0 || opt->off > opt->n) {
        set_error("bdrt_ridge: bad arguments");
        return -1;
    }
    for (int b = 0; b < nb; ++b) if (gsel[b] < 0 || gsel[b] >= ng) { set_error("bdrt_ridge: gsel out of range"); return -1; }
    if (opt->penalty == 0 && opt->hyper_lambda && opt->off + opt->K != opt->n) { set_error("bdrt_ridge: off + K must equal n"); return -1; }
    bind_process_device();
    const int n = opt->n, K = opt->K, np = (n + 1) & ~1, mi = opt->max_iter;
    const size_t msize = (size_t)n * (n + 1) / 2;
    const size_t vec = (size_t)(QP_NVEC + 1) * np + 32, extra = (size_t)6 * np + 32 + 2;
    const bool in_lds = (vec + msize + extra) * sizeof(double) <= 160 * 1024;
    const size_t lds = (vec + (in_lds ? msize : 0) + extra) * sizeof(double);
    if (lds > 160 * 1024) { set_error("bdrt_ridge: n = %d too large", n); return -2; }
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *p : owned) hipFree(p); };
#define RG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    auto up = [&](const void *h, size_t bytes, const void **d) -> hipError_t {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return e;
        owned.push_back(p);
        if (h) e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
        *d = p;
        return e;
    };
    RidgeArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.K = K; a.off = opt->off; a.penalty = opt->penalty; a.max_iter = mi; a.hyper_lambda = opt->hyper_lambda;
    a.zero_delta1 = opt->zero_delta1; a.xtol = opt->xtol; a.hl_fbeta = opt->hl_fbeta;
    for (int o = 0; o < 3; ++o) a.reg_ord[o] = opt->reg_ord[o];
    const void *d = nullptr;
    RG_HIP(up(G, (size_t)ng * n * n * 8, &d)); a.G = (const double *)d;
    RG_HIP(up(qbase, (size_t)ng * n * 8, &d)); a.qbase = (const double *)d;
    RG_HIP(up(gsel, (size_t)nb * sizeof(int), &d)); a.gsel = (const int *)d;
    RG_HIP(up(base, (size_t)3 * n * n * 8, &d)); a.base = (const double *)d;
    if (Ls) { RG_HIP(up(Ls, (size_t)3 * K * n * 8, &d)); a.Ls = (const double *)d; }
    if (lo) { RG_HIP(up(lo, (size_t)n * 8, &d)); a.lo = (const double *)d; }
    RG_HIP(up(lambda0, (size_t)nb * 8, &d)); a.lambda0 = (const double *)d;
    RG_HIP(up(lam0s, (size_t)nb * 3 * 8, &d)); a.lam0s = (const double *)d;
    RG_HIP(up(betas, (size_t)nb * 3 * 8, &d)); a.betas = (const double *)d;
    if (x0) { RG_HIP(up(x0, (size_t)nb * n * 8, &d)); a.x0 = (const double *)d; }
    RG_HIP(up(nullptr, (size_t)nb * n * n * 8, &d)); a.Pwork = (double *)d;
    if (!in_lds) { RG_HIP(up(nullptr, (size_t)nb * msize * 8, &d)); a.Mwork = (double *)d; }
    RG_HIP(up(nullptr, (size_t)nb * n * 8, &d)); a.coef = (double *)d;
    RG_HIP(up(nullptr, (size_t)nb * 3 * n * 8, &d)); a.lam = (double *)d;
    RG_HIP(up(nullptr, (size_t)nb * 8, &d)); a.cost = (double *)d;
    RG_HIP(up(nullptr, (size_t)nb * 8, &d)); a.fun = (double *)d;
    RG_HIP(up(nullptr, (size_t)nb * sizeof(int), &d)); a.iters = (int *)d;
    RG_HIP(up(nullptr, (size_t)nb * sizeof(int), &d)); a.flags = (int *)d;
    if (hist_coef && hist_lam && hist_fun && hist_cost) {
        RG_HIP(up(nullptr, (size_t)nb * mi * n * 8, &d)); a.hist_coef = (double *)d;
        RG_HIP(up(nullptr, (size_t)nb * mi * 3 * n * 8, &d)); a.hist_lam = (double *)d;
        RG_HIP(up(nullptr, (size_t)nb * mi * 8, &d)); a.hist_fun = (double *)d;
        RG_HIP(up(nullptr, (size_t)nb * mi * 8, &d)); a.hist_cost = (double *)d;
        RG_HIP(hipMemset(a.hist_fun, 0, (size_t)nb * mi * 8));
        RG_HIP(hipMemset(a.hist_cost, 0, (size_t)nb * mi * 8));
    }
    static LdsAttrCache attr_cache;
    RG_HIP(attr_cache.ensure(lds, [&]() {
        hipError_t e = hipFuncSetAttribute((const void *)ridge_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)ridge_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return e;
    }));
    if (in_lds) hipLaunchKernelGGL(ridge_kernel<true>, dim3(nb), dim3(QP_NT), lds, 0, a);
    else hipLaunchKernelGGL(ridge_kernel<false>, dim3(nb), dim3(QP_NT), lds, 0, a);
    RG_HIP(hipGetLastError());
    RG_HIP(hipDeviceSynchronize());
    RG_HIP(hipMemcpy(coef, a.coef, (size_t)nb * n * 8, hipMemcpyDeviceToHost));
    if (lam) RG_HIP(hipMemcpy(lam, a.lam, (size_t)nb * 3 * n * 8, hipMemcpyDeviceToHost));
    if (cost) RG_HIP(hipMemcpy(cost, a.cost, (size_t)nb * 8, hipMemcpyDeviceToHost));
    if (fun) RG_HIP(hipMemcpy(fun, a.fun, (size_t)nb * 8, hipMemcpyDeviceToHost));
    std::vector<int> hfl(nb);
    RG_HIP(hipMemcpy(hfl.data(), a.flags, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    if (iters) RG_HIP(hipMemcpy(iters, a.iters, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    if (flags) memcpy(flags, hfl.data(), (size_t)nb * sizeof(int));
    if (a.hist_coef) {
        RG_HIP(hipMemcpy(hist_coef, a.hist_coef, (size_t)nb * mi * n * 8, hipMemcpyDeviceToHost));
        RG_HIP(hipMemcpy(hist_lam, a.hist_lam, (size_t)nb * mi * 3 * n * 8, hipMemcpyDeviceToHost));
        RG_HIP(hipMemcpy(hist_fun, a.hist_fun, (size_t)nb * mi * 8, hipMemcpyDeviceToHost));
        RG_HIP(hipMemcpy(hist_cost, a.hist_cost, (size_t)nb * mi * 8, hipMemcpyDeviceToHost));
    }
#undef RG_HIP
    cleanup();
    for (int b = 0; b < nb; ++b)
        if (hfl[b] == -3) { set_error("bdrt_ridge: KKT matrix of fit %d not positive definite", b); return -3; }
    return 0;
}

}  // extern "C"
