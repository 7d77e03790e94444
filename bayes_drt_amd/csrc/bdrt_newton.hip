// bdrt_newton.hip -- second-order MAP polish on the device (the state machine of bdrt_newton.h, which remains as the
// host form the tests drive with oracle evaluations).
//
// Round 1 kept this iteration on the host: every Levenberg-Marquardt step copied the 2 D probe gradients (1.75 MB at
// D = 331) over PCIe and factored the D x D Hessian with a scalar Cholesky on one host core -- ~6 ms per step, ~3 s per fit,
// no faster than the reference's CPU.  Here nothing but a 64-byte status record leaves the GPU:
//   probes      x +- h_j e_j for all j                                   newton_probe_kernel   (2 D rows per fit)
//   gradients   the batched MFMA log-posterior kernel on the 2 D probes   launch_logp_grad
//   step        H from central differences, symmetrised; (-H + lam I) = L L^T by a blocked right-looking Cholesky
//               (16 x 16 diagonal blocks in LDS, panel solves one row per thread, trailing update on v_mfma_f64_16x16x4 tiles);
//               s = (L L^T)^-1 g; trial point x + s; predicted increase       newton_solve_kernel   (one workgroup per fit)
//   trial       log-posterior + gradient at x + s                            launch_logp_grad
//   decision    accept / reject, lam update, convergence                     newton_accept_kernel
// The host only sequences these launches and reads the status records; fits of a batch advance together.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bdrt_host.h"

namespace bdrt {

constexpr int NW_NT = 512;

struct NewtonState {            // one per fit, in device memory
    double lp, lam, pred, lp_trial, grad_inf;
    int iters, rc, done, need_hess, n_evals, max_iter, pad0, pad1;
    double tol;
};

struct NewtonBufs {
    int D, Dp;                  // Dp = D rounded up to 16
    double *x, *g, *s, *xt, *gt, *hstep;   // [n_fits][D]
    double *H, *M;              // [n_fits][Dp][Dp]
    double *probes, *pgrad;     // [n_fits][2 D][D]
    double *plp;                // [n_fits][2 D]
    NewtonState *st;            // [n_fits]
};

__global__ void newton_probe_kernel(NewtonBufs b, const int *active, int n_active)
{
    const int a = blockIdx.y, row = blockIdx.x;          // row in [0, 2 D)
    if (a >= n_active) return;
    const int f = active[a], D = b.D, j = row >> 1;
    const double *x = b.x + (size_t)f * D;
    double *p = b.probes + ((size_t)a * 2 * D + row) * D;
    const double h = 1e-5 * fmax(1.0, fabs(x[j]));
    if (threadIdx.x == 0 && (row & 1) == 0) b.hstep[(size_t)f * D + j] = h;
    for (int k = threadIdx.x; k < D; k += blockDim.x) p[k] = x[k] + (k == j ? ((row & 1) ? -h : h) : 0.0);
}

// In-place blocked Cholesky of the lower triangle of the Dp x Dp row-major matrix M (Dp a multiple of 16).
// lds: 16*17 (diagonal block) + 16 (reciprocal pivots) + Dp*17 (panel) + 1 (flag) doubles.  Returns false when not positive definite.
__device__ inline bool chol_blocked(double *M, int Dp, double *lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *Dg = lds, *rinv = Dg + 16 * 17, *Pn = rinv + 16;
    int *bad = reinterpret_cast<int *>(Pn + (size_t)Dp * 17);
    const int nblk = Dp / 16;
    if (tid == 0) *bad = 0;
    __syncthreads();
    for (int J = 0; J < nblk; ++J) {
        const int j0 = 16 * J;
        // (a) diagonal block -> LDS, factored by 16 threads of wave 0
        if (tid < 256) Dg[(tid >> 4) * 17 + (tid & 15)] = M[(size_t)(j0 + (tid >> 4)) * Dp + j0 + (tid & 15)];
        __syncthreads();
        if (wave == 0) {
            // lane i keeps row i of the block in registers; column j of the factor travels through LDS (one write, then
            // independent reads): no read-modify-write chains through LDS
            const int i = lane & 15;
            double row[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k] = Dg[i * 17 + k];
            double *colb = rinv;                                      // 16 doubles, rewritten with 1 / L[j][j] at the end
            bool okb = true;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(row[j]), j),
                                                  __builtin_amdgcn_readlane(__double2loint(row[j]), j));
                if (!(d > 0.0) || !isfinite(d)) { okb = false; break; }
                const double dj = sqrt(d), inv = 1.0 / dj;
                const double lij = i == j ? dj : row[j] * inv;
                row[j] = lij;
                if (lane < 16) colb[i] = lij;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = j + 1; k < 16; ++k) {
                    const double lkj = colb[k];
                    if (i > j && k <= i) row[k] -= lij * lkj;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (!okb) { if (lane == 0) *bad = 1; }
            else if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; ++k) Dg[i * 17 + k] = row[k];
                rinv[i] = 1.0 / row[i];
            }
        }
        __syncthreads();
        if (*bad) return false;
        if (tid < 256 && (tid & 15) <= (tid >> 4)) M[(size_t)(j0 + (tid >> 4)) * Dp + j0 + (tid & 15)] = Dg[(tid >> 4) * 17 + (tid & 15)];
        // (b) panel below the block: x L11^T = a, one row per thread.  The m x 16 panel moves between global memory and
        // LDS with coalesced accesses (16 lanes per 128-byte row segment); the solve itself works on the LDS copy.
        const int m = Dp - j0 - 16;
        for (int e = tid; e < m * 16; e += NW_NT) Pn[(size_t)(e >> 4) * 17 + (e & 15)] = M[(size_t)(j0 + 16 + (e >> 4)) * Dp + j0 + (e & 15)];
        __syncthreads();
        for (int r = tid; r < m; r += NW_NT) {
            double *row = Pn + (size_t)r * 17;
            double xk[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) xk[k] = row[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                double t = xk[k];
#pragma unroll
                for (int u = 0; u < k; ++u) t -= xk[u] * Dg[k * 17 + u];
                xk[k] = t * rinv[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k] = xk[k];
        }
        __syncthreads();
        for (int e = tid; e < m * 16; e += NW_NT) M[(size_t)(j0 + 16 + (e >> 4)) * Dp + j0 + (e & 15)] = Pn[(size_t)(e >> 4) * 17 + (e & 15)];
        // (c) trailing update C -= P P^T on 16 x 16 tiles of the lower triangle (tiles on the diagonal are computed in full)
        const int mb = m / 16, ntile = mb * (mb + 1) / 2;
        const int col = lane & 15, kq = lane >> 4;
        auto tile_of = [&](int t, int &I, int &Jc) {
            I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            while (I * (I + 1) / 2 > t) --I;
            Jc = t - I * (I + 1) / 2;
        };
        // two tiles per trip: the loads of the second overlap the MFMAs of the first
        for (int t = wave; t < ntile; t += 2 * (NW_NT / 64)) {
            const int t2 = t + NW_NT / 64;
            const bool two = t2 < ntile;
            int I0, J0, I1, J1;
            tile_of(t, I0, J0);
            tile_of(two ? t2 : t, I1, J1);
            double *C0 = M + (size_t)(j0 + 16 + 16 * I0) * Dp + j0 + 16 + 16 * J0;
            double *C1 = M + (size_t)(j0 + 16 + 16 * I1) * Dp + j0 + 16 + 16 * J1;
            d4 a0, a1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a0[r] = C0[(size_t)(kq + 4 * r) * Dp + col]; a1[r] = C1[(size_t)(kq + 4 * r) * Dp + col]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a0 = mfma_f64(-Pn[(size_t)(16 * I0 + col) * 17 + 4 * u + kq], Pn[(size_t)(16 * J0 + col) * 17 + 4 * u + kq], a0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a1 = mfma_f64(-Pn[(size_t)(16 * I1 + col) * 17 + 4 * u + kq], Pn[(size_t)(16 * J1 + col) * 17 + 4 * u + kq], a1);
#pragma unroll
            for (int r = 0; r < 4; ++r) C0[(size_t)(kq + 4 * r) * Dp + col] = a0[r];
            if (two) {
#pragma unroll
                for (int r = 0; r < 4; ++r) C1[(size_t)(kq + 4 * r) * Dp + col] = a1[r];
            }
        }
        __syncthreads();
    }
    return true;
}

// v <- (L L^T)^-1 v, L = lower triangle of the row-major Dp x Dp matrix M (Dp a multiple of 16, identity on the padding);
// v [Dp] in LDS; Dg: 16*17 doubles of LDS scratch.  Blocked substitution: the 16 x 16 diagonal block is solved by one
// thread from LDS, the rest of the right-hand side is updated by the whole workgroup with coalesced row reads.
__device__ inline void chol_blocked_solve(const double *M, int Dp, double *v, double *Dg)
{
    const int tid = threadIdx.x;
    const int nblk = Dp / 16;
    for (int J = 0; J < nblk; ++J) {                                  // forward: L y = v
        const int j0 = 16 * J;
        if (tid < 256) Dg[(tid >> 4) * 17 + (tid & 15)] = M[(size_t)(j0 + (tid >> 4)) * Dp + j0 + (tid & 15)];
        __syncthreads();
        if (tid < 64) {
            // lane j holds row j of the block; y_k is broadcast as soon as it is known (16 steps of one multiply-add)
            const int j = tid & 15;
            double Lr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) Lr[k] = Dg[j * 17 + k];
            const double rd = 1.0 / Lr[j];
            double t = v[j0 + j];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double c = t * rd;
                const double yk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), k), __builtin_amdgcn_readlane(__double2loint(c), k));
                if (j > k) t -= Lr[k] * yk;
                if (j == k) t = yk * Lr[j];                           // keeps t / L[j][j] = y_j for the store below
            }
            if (tid < 16) v[j0 + j] = t * rd;
        }
        __syncthreads();
        // rows below: v_i -= L[i][j0 .. j0+16) . y  -- 16 lanes per row
        const int lane16 = tid & 15;
        const double yk = v[j0 + lane16];
        for (int i = j0 + 16 + (tid >> 4); i < Dp; i += NW_NT / 16) {
            double t = M[(size_t)i * Dp + j0 + lane16] * yk;
            t += dpp_perm<0xB1>(t); t += dpp_perm<0x4E>(t); t += dpp_perm<0x141>(t); t += dpp_perm<0x140>(t);   // sum over the 16-lane row
            if (lane16 == 0) v[i] -= t;
        }
        __syncthreads();
    }
    for (int J = nblk - 1; J >= 0; --J) {                             // backward: L^T x = y
        const int j0 = 16 * J;
        if (tid < 256) Dg[(tid >> 4) * 17 + (tid & 15)] = M[(size_t)(j0 + (tid >> 4)) * Dp + j0 + (tid & 15)];
        __syncthreads();
        if (tid < 64) {
            // lane j holds column j of the block (row j of L^T); x_k broadcast from the last unknown upwards
            const int j = tid & 15;
            double Lc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) Lc[k] = Dg[k * 17 + j];
            const double rd = 1.0 / Lc[j];
            double t = v[j0 + j];
#pragma unroll
            for (int k = 15; k >= 0; --k) {
                const double c = t * rd;
                const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), k), __builtin_amdgcn_readlane(__double2loint(c), k));
                if (j < k) t -= Lc[k] * xk;
                if (j == k) t = xk * Lc[j];
            }
            if (tid < 16) v[j0 + j] = t * rd;
        }
        __syncthreads();
        // columns to the left: v_k -= sum_i L[j0 + i][k] x_{j0 + i}  -- one k per thread, coalesced along k
        for (int k = tid; k < j0; k += NW_NT) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += M[(size_t)(j0 + i) * Dp + k] * v[j0 + i];
            v[k] -= t;
        }
        __syncthreads();
    }
}

// One Levenberg-Marquardt step per active fit: (optionally) H from the probe gradients, then the damped solve and the trial point.
__global__ __launch_bounds__(NW_NT) void newton_solve_kernel(NewtonBufs b, const int *active, int n_active)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int a = blockIdx.x, tid = threadIdx.x;
    const int f = active[a], D = b.D, Dp = b.Dp;
    NewtonState &S = b.st[f];
    if (S.done) return;
    double *H = b.H + (size_t)f * Dp * Dp, *M = b.M + (size_t)f * Dp * Dp;
    const double *x = b.x + (size_t)f * D, *g = b.g + (size_t)f * D, *hs = b.hstep + (size_t)f * D;
    double *s = b.s + (size_t)f * D, *xt = b.xt + (size_t)f * D;
    double *v = sh;                               // [Dp] right-hand side / solution
    double *red = v + Dp;                         // 64 doubles
    double *cl = red + 64;                        // Cholesky scratch
    __shared__ int s_fin;
    if (S.need_hess) {
        const double *pg = b.pgrad + (size_t)a * 2 * D * D;
        int fin = 1;
        for (size_t e = tid; e < (size_t)D * D; e += NW_NT) {
            const int j = (int)(e / D), k = (int)(e - (size_t)j * D);
            const double hv = (pg[(size_t)(2 * j) * D + k] - pg[(size_t)(2 * j + 1) * D + k]) / (2.0 * hs[j]);
            H[(size_t)j * Dp + k] = hv;
            if (!isfinite(hv)) fin = 0;
        }
        fin = __syncthreads_and(fin);
        if (!fin) { if (tid == 0) { S.rc = 2; S.done = 1; } return; }
        for (size_t e = tid; e < (size_t)D * D; e += NW_NT) {        // symmetrise (each pair by the thread of its lower element)
            const int j = (int)(e / D), k = (int)(e - (size_t)j * D);
            if (k < j) {
                const double m = 0.5 * (H[(size_t)j * Dp + k] + H[(size_t)k * Dp + j]);
                H[(size_t)j * Dp + k] = m; H[(size_t)k * Dp + j] = m;
            }
        }
        __syncthreads();
        if (tid == 0) { S.need_hess = 0; S.n_evals += 2 * D; }
    }
    double lam = S.lam;
    bool ok = false;
    while (lam <= 1e12) {
        for (size_t e = tid; e < (size_t)Dp * Dp; e += NW_NT) {
            const int i = (int)(e / Dp), k = (int)(e - (size_t)i * Dp);
            double m = 0.0;
            if (i < D && k < D) m = -H[e] + (i == k ? lam : 0.0);
            else if (i == k) m = 1.0;                              // identity on the padding
            M[e] = m;
        }
        __syncthreads();
        if (chol_blocked(M, Dp, cl)) {
            for (int i = tid; i < Dp; i += NW_NT) v[i] = i < D ? g[i] : 0.0;
            __syncthreads();
            chol_blocked_solve(M, Dp, v, cl);
            int fin = 1;
            for (int i = tid; i < D; i += NW_NT) {
                const double t = x[i] + v[i];
                if (!isfinite(t)) fin = 0;
            }
            if (tid == 0) s_fin = 1;
            __syncthreads();
            if (!fin) s_fin = 0;
            __syncthreads();
            if (s_fin) { ok = true; break; }
        }
        lam *= 4.0;
        __syncthreads();
    }
    if (!ok) { if (tid == 0) { S.rc = 2; S.done = 1; S.lam = lam; } return; }
    // trial point and predicted increase g.s + 1/2 s^T H s
    double pred = 0.0;
    for (int i = tid; i < D; i += NW_NT) { s[i] = v[i]; xt[i] = x[i] + v[i]; }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < D; j += NW_NT / 64) {
        const double *rowj = H + (size_t)j * Dp;
        double t = 0.0;
        for (int k = lane; k < D; k += 64) t += rowj[k] * v[k];
        t = sum32(t); t += __shfl_xor(t, 32);
        if (lane == 0) pred += v[j] * (g[j] + 0.5 * t);
    }
    if (lane == 0) red[wave] = pred;
    __syncthreads();
    if (tid == 0) {
        double p = 0.0;
        for (int w = 0; w < NW_NT / 64; ++w) p += red[w];
        S.pred = p; S.lam = lam;
    }
}

// consume the trial evaluation (lp_t, grad_t in gt): accept / reject (bdrt_newton.h NewtonFit::consume, NEED_TRIAL)
__global__ __launch_bounds__(256) void newton_accept_kernel(NewtonBufs b, const int *active, int n_active, const double *lp_t)
{
    const int a = blockIdx.x, tid = threadIdx.x;
    const int f = active[a], D = b.D;
    NewtonState &S = b.st[f];
    if (S.done) return;
    __shared__ double red[256];
    __shared__ int s_acc;
    double *x = b.x + (size_t)f * D, *g = b.g + (size_t)f * D;
    const double *xt = b.xt + (size_t)f * D, *gt = b.gt + (size_t)a * D;
    const double lpn = lp_t[a];
    int fin = isfinite(lpn) ? 1 : 0;
    double gi = 0.0;
    for (int j = tid; j < D; j += 256) { const double v = gt[j]; if (!isfinite(v)) fin = 0; gi = fmax(gi, fabs(v)); }
    fin = __syncthreads_and(fin);
    red[tid] = gi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] = fmax(red[tid], red[tid + w]); __syncthreads(); }
    const double ginf = red[0];
    if (tid == 0) {
        const double lp = S.lp, pred = S.pred;
        s_acc = fin && (lpn - lp >= 1e-4 * pred) && (lpn >= lp - 1e-12 * fabs(lp));
        // close to the optimum the predicted increase drops below the resolution of lp itself (a few ulp of |lp|): the
        // sufficient-increase test then compares rounding noise.  There the step is judged by what it is meant to do --
        // reduce the gradient -- as long as lp does not visibly decrease.
        const double noise = 64.0 * 2.220446049250313e-16 * fmax(1.0, fabs(lp));
        if (fin && !s_acc && pred < noise && lpn >= lp - noise && ginf < S.grad_inf) s_acc = 1;
        S.n_evals += 1;
        S.lp_trial = lpn;
    }
    __syncthreads();
    if (s_acc) {
        for (int j = tid; j < D; j += 256) { x[j] = xt[j]; g[j] = gt[j]; }
        if (tid == 0) {
            const double pred = S.pred, rho = pred > 0.0 ? (lpn - S.lp) / pred : 0.0;
            S.lp = lpn;
            if (rho > 0.75) S.lam = fmax(S.lam / 5.0, 1e-12);
            else if (rho < 0.25) S.lam *= 2.0;
            S.iters += 1;
            S.grad_inf = ginf;
            if (ginf < S.tol) { S.rc = 0; S.done = 1; }
            else if (S.iters >= S.max_iter) { S.rc = 1; S.done = 1; }
            else S.need_hess = 1;
        }
    } else if (tid == 0) {
        S.lam *= 4.0;                                             // same Hessian, more damping
    }
}

// dense batch of the active fits' trial points: row a <- xt of fit active[a]
__global__ void newton_gather_kernel(NewtonBufs b, const int *active, int n_active, double *dst)
{
    const int a = blockIdx.x;
    const double *src = b.xt + (size_t)active[a] * b.D;
    for (int k = threadIdx.x; k < b.D; k += blockDim.x) dst[(size_t)a * b.D + k] = src[k];
}

// first evaluation at the start point: lp, g, convergence test
__global__ __launch_bounds__(256) void newton_init_kernel(NewtonBufs b, int n_fits, const double *lp0, const double *g0)
{
    const int f = blockIdx.x, tid = threadIdx.x, D = b.D;
    NewtonState &S = b.st[f];
    __shared__ double red[256];
    double *g = b.g + (size_t)f * D;
    int fin = isfinite(lp0[f]) ? 1 : 0;
    double gi = 0.0;
    for (int j = tid; j < D; j += 256) { const double v = g0[(size_t)f * D + j]; g[j] = v; if (!isfinite(v)) fin = 0; gi = fmax(gi, fabs(v)); }
    fin = __syncthreads_and(fin);
    red[tid] = gi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] = fmax(red[tid], red[tid + w]); __syncthreads(); }
    if (tid == 0) {
        S.lp = lp0[f]; S.n_evals = 1; S.grad_inf = red[0];
        if (!fin) { S.rc = -1; S.done = 1; }
        else if (red[0] < S.tol) { S.rc = 0; S.done = 1; }
        else S.need_hess = 1;
    }
}

// Newton polish of n_fits points (x0 [n_fits][D] on the host); results back to x_out, reports filled
int newton_polish_device(Problem &P, const double *x0, const int *spec, int n_fits, int max_iter, double tol, double *x_out,
                         double *lp_out, double *ginf_out, int *iters_out, int *rc_out, int *n_evals_out)
{
    const int D = P.dev.D, Dp = (D + 15) & ~15;
    BDRT_HIP(hipSetDevice(P.device));
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *p : owned) hipFree(p); };
#define NW_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    auto alloc = [&](size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 8);
        if (e == hipSuccess) owned.push_back(*d);
        return e;
    };
    NewtonBufs b;
    b.D = D; b.Dp = Dp;
    const size_t nD = (size_t)n_fits * D * sizeof(double);
    NW_HIP(alloc(nD, (void **)&b.x)); NW_HIP(alloc(nD, (void **)&b.g)); NW_HIP(alloc(nD, (void **)&b.s));
    NW_HIP(alloc(nD, (void **)&b.xt)); NW_HIP(alloc(nD, (void **)&b.gt)); NW_HIP(alloc(nD, (void **)&b.hstep));
    NW_HIP(alloc((size_t)n_fits * Dp * Dp * sizeof(double), (void **)&b.H));
    NW_HIP(alloc((size_t)n_fits * Dp * Dp * sizeof(double), (void **)&b.M));
    NW_HIP(alloc((size_t)n_fits * 2 * D * D * sizeof(double), (void **)&b.probes));
    NW_HIP(alloc((size_t)n_fits * 2 * D * D * sizeof(double), (void **)&b.pgrad));
    NW_HIP(alloc((size_t)n_fits * 2 * D * sizeof(double), (void **)&b.plp));
    NW_HIP(alloc((size_t)n_fits * sizeof(NewtonState), (void **)&b.st));
    int *d_active = nullptr, *d_spec1 = nullptr, *d_specp = nullptr;
    double *d_lpt = nullptr;
    NW_HIP(alloc((size_t)n_fits * sizeof(int), (void **)&d_active));
    NW_HIP(alloc((size_t)n_fits * sizeof(int), (void **)&d_spec1));
    NW_HIP(alloc((size_t)n_fits * 2 * D * sizeof(int), (void **)&d_specp));
    NW_HIP(alloc((size_t)n_fits * sizeof(double), (void **)&d_lpt));
    std::vector<NewtonState> hst((size_t)n_fits);
    for (auto &s : hst) { memset(&s, 0, sizeof(s)); s.lam = 1e-3; s.max_iter = max_iter; s.tol = tol; s.rc = 1; s.done = max_iter > 0 ? 0 : 1; }
    NW_HIP(hipMemcpy(b.st, hst.data(), hst.size() * sizeof(NewtonState), hipMemcpyHostToDevice));
    NW_HIP(hipMemcpy(b.x, x0, nD, hipMemcpyHostToDevice));
    hipStream_t st = P.stream;
    // evaluation at the start points
    std::vector<int> hspec((size_t)n_fits), hact((size_t)n_fits), hspecp, sp1((size_t)n_fits);
    for (int i = 0; i < n_fits; ++i) hspec[i] = spec ? spec[i] : 0;
    NW_HIP(hipMemcpyAsync(d_spec1, hspec.data(), (size_t)n_fits * sizeof(int), hipMemcpyHostToDevice, st));
    int rc;
    if ((rc = launch_logp_grad(&P, b.x, d_spec1, n_fits, 0, d_lpt, b.gt, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
    hipLaunchKernelGGL(newton_init_kernel, dim3(n_fits), dim3(256), 0, st, b, n_fits, (const double *)d_lpt, (const double *)b.gt);
    const size_t lds_solve = ((size_t)Dp + 64 + 16 * 17 + 16 + (size_t)Dp * 17 + 2) * sizeof(double);
    NW_HIP(hipFuncSetAttribute((const void *)newton_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
    const long long max_rounds = (long long)max_iter * 40 + 100;
    for (long long round = 0; round < max_rounds; ++round) {
        NW_HIP(hipMemcpyAsync(hst.data(), b.st, hst.size() * sizeof(NewtonState), hipMemcpyDeviceToHost, st));
        NW_HIP(hipStreamSynchronize(st));
        int n_active = 0, n_hess = 0;
        // fits that need a fresh Hessian first, so that their probe rows are contiguous [0, n_hess)
        for (int i = 0; i < n_fits; ++i) if (!hst[i].done && hst[i].need_hess) hact[n_active++] = i;
        n_hess = n_active;
        for (int i = 0; i < n_fits; ++i) if (!hst[i].done && !hst[i].need_hess) hact[n_active++] = i;
        if (n_active == 0) break;
        NW_HIP(hipMemcpyAsync(d_active, hact.data(), (size_t)n_active * sizeof(int), hipMemcpyHostToDevice, st));
        if (n_hess > 0) {
            hspecp.resize((size_t)n_hess * 2 * D);
            for (int a = 0; a < n_hess; ++a) std::fill(hspecp.begin() + (size_t)a * 2 * D, hspecp.begin() + (size_t)(a + 1) * 2 * D, hspec[hact[a]]);
            NW_HIP(hipMemcpyAsync(d_specp, hspecp.data(), hspecp.size() * sizeof(int), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(newton_probe_kernel, dim3(2 * D, n_hess), dim3(128), 0, st, b, (const int *)d_active, n_hess);
            if ((rc = launch_logp_grad(&P, b.probes, d_specp, n_hess * 2 * D, 0, b.plp, b.pgrad, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
        }
        hipLaunchKernelGGL(newton_solve_kernel, dim3(n_active), dim3(NW_NT), lds_solve, st, b, (const int *)d_active, n_active);
        // trial points of the active fits, gathered into contiguous rows of gt / lp_t: evaluate xt of fit hact[a] into slot a
        // (xt rows are per fit; the evaluator wants a dense batch: copy the active rows)
        hipLaunchKernelGGL(newton_gather_kernel, dim3(n_active), dim3(128), 0, st, b, (const int *)d_active, n_active, b.probes);
        for (int a = 0; a < n_active; ++a) sp1[a] = hspec[hact[a]];      // (host buffers are reused only after the sync above)
        NW_HIP(hipMemcpyAsync(d_spec1, sp1.data(), (size_t)n_active * sizeof(int), hipMemcpyHostToDevice, st));
        if ((rc = launch_logp_grad(&P, b.probes, d_spec1, n_active, 0, d_lpt, b.gt, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
        hipLaunchKernelGGL(newton_accept_kernel, dim3(n_active), dim3(256), 0, st, b, (const int *)d_active, n_active, (const double *)d_lpt);
        NW_HIP(hipGetLastError());
    }
    NW_HIP(hipMemcpyAsync(hst.data(), b.st, hst.size() * sizeof(NewtonState), hipMemcpyDeviceToHost, st));
    NW_HIP(hipMemcpyAsync(x_out, b.x, nD, hipMemcpyDeviceToHost, st));
    NW_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < n_fits; ++i) {
        if (lp_out) lp_out[i] = hst[i].lp;
        if (ginf_out) ginf_out[i] = hst[i].grad_inf;
        if (iters_out) iters_out[i] = hst[i].iters;
        if (rc_out) rc_out[i] = hst[i].done ? hst[i].rc : 1;
        if (n_evals_out) n_evals_out[i] = hst[i].n_evals;
    }
#undef NW_HIP
    cleanup();
    return 0;
}

}  // namespace bdrt
