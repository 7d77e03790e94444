// bdrt_qp.hip -- batched box-constrained QP on the GPU: the ridge / hyper-lambda ridge solve (include/bdrt.h section (3)).
//
// Replaces cvxopt.solvers.qp inside Inverter._convex_opt (reference bayes_drt/inversion.py:1043-1067), which the
// reference calls once per hyper-lambda iteration and 2 x len(lambdas) x iterations times inside ridge_ReImCV (:902-945):
//     min 1/2 x'Px + q'x   s.t.  x >= lo        (lo[i] = -inf: free variable)
// Same algorithm and tolerances as the host solver bdrt_qp_box (bdrt_ridge.hip): Mehrotra predictor-corrector on
// (P + diag(z/s)) dx = -r_d + (sigma mu - s z - ds_aff dz_aff)/s, cvxopt's default abstol 1e-7 / reltol 1e-6 / feastol 1e-7,
// strictly interior iterates.
//
// One workgroup (512 threads) per problem; problems of a batch run concurrently (Re-Im cross-validation: 62 fits).  The
// KKT matrix lives in LDS as a column-major packed lower triangle (n <= 200: 157 KiB) -- or in a global work buffer for
// larger n -- and is factored by a right-looking Cholesky (two barriers per column; the trailing update is a 32 x 16
// thread tiling with conflict-free column walks).  The two triangular solves of a Newton direction are sequential by
// nature: one wavefront does them with wave-level synchronisation only (a barrier per unknown would cost more than the
// arithmetic).
#include <cmath>
#include <vector>

#include "bdrt_host.h"

namespace bdrt {

constexpr int QP_NT = 512;
constexpr int QP_NVEC = 13;          // x, s, z, rd, rhs, dx, ds, dz, dsa, dza, Px, lo, bounded
constexpr int QP_MAX_IT = 200;

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

template <bool LDSM>
__global__ __launch_bounds__(QP_NT) void qp_box_kernel(const double *__restrict__ Pall, const double *__restrict__ qall,
                                                       const double *__restrict__ lo, int n, double *__restrict__ Xall,
                                                       double *__restrict__ objall, int *__restrict__ itall,
                                                       double *__restrict__ work)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *P = Pall + (size_t)b * n * n, *q = qall + (size_t)b * n;
    const int np = (n + 1) & ~1;
    double *x = sh, *s = x + np, *z = s + np, *rd = z + np, *rhs = rd + np, *dx = rhs + np, *ds = dx + np, *dz = ds + np,
           *dsa = dz + np, *dza = dsa + np, *Px = dza + np, *lv = Px + np, *bd = lv + np, *diag = bd + np, *red = diag + np;
    const size_t msize = (size_t)n * (n + 1) / 2;
    double *M = LDSM ? red + 32 : work + (size_t)b * msize;
    const double abstol = 1e-7, reltol = 1e-6, feastol = 1e-7;      // cvxopt.solvers defaults

    double nbv = 0.0, qq = 0.0, dummy = 0.0;
    for (int i = tid; i < n; i += QP_NT) {
        const double li = lo ? lo[i] : -INFINITY;
        const bool bnd = isfinite(li);
        bd[i] = bnd ? 1.0 : 0.0;
        lv[i] = bnd ? li : 0.0;
        x[i] = bnd ? li + 1.0 : 0.0;                                  // strictly feasible start, unit slacks / multipliers
        s[i] = 1.0;
        z[i] = bnd ? 1.0 : 0.0;
        nbv += bnd ? 1.0 : 0.0;
        qq += q[i] * q[i];
    }
    block_sum3(nbv, qq, dummy, red);
    const double qnorm = fmax(1.0, sqrt(qq));
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

    int it = 0, status = 0;
    for (; it < QP_MAX_IT; ++it) {
        matvec();
        double pobj = 0.0, gap = 0.0, rdn = 0.0;
        for (int i = tid; i < n; i += QP_NT) {
            pobj += x[i] * (0.5 * Px[i] + q[i]);
            const double r = Px[i] + q[i] - z[i];
            rd[i] = r;
            rdn += r * r;
            if (bd[i] != 0.0) gap += s[i] * z[i];
        }
        block_sum3(pobj, gap, rdn, red);
        rdn = sqrt(rdn);
        const double dobj = pobj - gap;                               // dual objective when r_d = 0
        double relgap = INFINITY;
        if (pobj < 0) relgap = gap / -pobj; else if (dobj > 0) relgap = gap / dobj;
        if (rdn / qnorm <= feastol && (gap <= abstol || relgap <= reltol)) break;
        const double mu = nb ? gap / nb : 0.0;
        // factor M = sym(P) + diag(z/s) (+ reg I when P is only semi-definite)
        double reg = 0.0;
        bool ok = false;
        while (true) {
            __syncthreads();
            for (int j = wave; j < n; j += QP_NT / 64) {              // one wavefront per column of the lower triangle
                double *colj = M + cidx(j, j, n);
                for (int i = j + lane; i < n; i += 64) {
                    double v = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
                    if (i == j) v += (bd[i] != 0.0 ? z[i] / s[i] : 0.0) + reg;
                    colj[i - j] = v;
                }
            }
            __syncthreads();
            const double m00 = M[0] - reg;
            ok = chol_packed(M, diag, n);
            if (ok) break;
            reg = reg == 0.0 ? 1e-14 * (1.0 + fabs(m00)) : reg * 100.0;
            if (reg > 1e6) break;
        }
        if (!ok) { status = -3; break; }

        auto solve_dir = [&](double sigma_mu, bool corrector) {
            for (int i = tid; i < n; i += QP_NT) {
                double rc = 0.0;
                if (bd[i] != 0.0) {
                    rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                }
                rhs[i] = -rd[i] + (bd[i] != 0.0 ? rc / s[i] : 0.0);
                dx[i] = rhs[i];
            }
            __syncthreads();
            chol_solve_wave(M, diag, n, dx);
            for (int i = tid; i < n; i += QP_NT) {
                if (bd[i] != 0.0) {
                    double rc = sigma_mu - s[i] * z[i];
                    if (corrector) rc -= dsa[i] * dza[i];
                    ds[i] = dx[i];
                    dz[i] = (rc - z[i] * ds[i]) / s[i];
                } else { ds[i] = 0.0; dz[i] = 0.0; }
            }
            __syncthreads();
        };
        auto max_step = [&]() {
            double a = 1.0;
            for (int i = tid; i < n; i += QP_NT) {
                if (bd[i] == 0.0) continue;
                if (ds[i] < 0) a = fmin(a, -s[i] / ds[i]);
                if (dz[i] < 0) a = fmin(a, -z[i] / dz[i]);
            }
            return block_min(a, red);
        };
        solve_dir(0.0, false);                                        // predictor (affine scaling)
        const double a_aff = max_step();
        double gap_aff = 0.0, d1 = 0.0, d2 = 0.0;
        for (int i = tid; i < n; i += QP_NT)
            if (bd[i] != 0.0) gap_aff += (s[i] + a_aff * ds[i]) * (z[i] + a_aff * dz[i]);
        block_sum3(gap_aff, d1, d2, red);
        double sigma = (nb && gap > 0) ? pow(gap_aff / gap, 3.0) : 0.0;
        sigma = fmin(1.0, fmax(0.0, sigma));
        for (int i = tid; i < n; i += QP_NT) { dsa[i] = ds[i]; dza[i] = dz[i]; }
        __syncthreads();
        solve_dir(sigma * mu, true);                                  // corrector
        const double a = fmin(1.0, 0.99 * max_step());
        for (int i = tid; i < n; i += QP_NT) {
            x[i] += a * dx[i];
            if (bd[i] != 0.0) {
                s[i] += a * ds[i];
                z[i] += a * dz[i];
                s[i] = x[i] - lv[i] > 0 ? x[i] - lv[i] : s[i];
            }
        }
        __syncthreads();
    }
    matvec();
    double pobj = 0.0, d1 = 0.0, d2 = 0.0;
    for (int i = tid; i < n; i += QP_NT) {
        pobj += x[i] * (0.5 * Px[i] + q[i]);
        Xall[(size_t)b * n + i] = x[i];
    }
    block_sum3(pobj, d1, d2, red);
    if (tid == 0) {
        objall[b] = pobj;
        itall[b] = status < 0 ? status : (it >= QP_MAX_IT ? -4 : it);
    }
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
        if (its[b] < 0 && worst == 0) worst = its[b];
    }
    if (worst == -3) { set_error("bdrt_qp_box_batch: KKT matrix not positive definite"); return -3; }
    if (worst == -4) { set_error("bdrt_qp_box_batch: iteration limit"); return -4; }
    return 0;
}

}  // extern "C"
