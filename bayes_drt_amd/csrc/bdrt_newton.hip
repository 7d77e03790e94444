// bdrt_newton.hip -- second-order MAP polish on the device (the state machine of bdrt_newton.h, which remains as the
// host form the tests drive with oracle evaluations).
//
// Round 1 kept this iteration on the host: every Levenberg-Marquardt step copied the 2 D probe gradients (1.75 MB at
// D = 331) over PCIe and factored the D x D Hessian with a scalar Cholesky on one host core -- ~6 ms per step, ~3 s per fit,
// no faster than the reference's CPU.  Here nothing but a 64-byte status record leaves the GPU:
//   Hessian     single series distribution (the headline family): CLOSED FORM, bdrt_newton_hess.h -- newton_hess_prep_kernel (one
//               workgroup per fit) + newton_hess_fill_kernel (16 rows per workgroup; the dense A^T C A product, bands, borders); four
//               gradient evaluations per round (the trial points) instead of 2 D + 4.  Once the damping is below 1e-4 the coefficients
//               of a Series_pos fit are iterated on the linear scale with a floor (projected Newton): 105 rounds instead of 250 at K = 161.
//               Every other family: central differences -- probes x +- h_j e_j (newton_probe_kernel), the batched MFMA log-posterior
//               kernel on the 2 D probes (launch_logp_grad), H symmetrised by newton_build_kernel.  BDRT_NEWTON_FD=1 forces this path.
//   step        (-H + lam I) = L L^T by a blocked right-looking Cholesky (16 x 16 diagonal blocks in registers, panel solves one row
//               per thread, trailing update on v_mfma_f64_16x16x4 tiles); s = (L L^T)^-1 g; trial point; predicted increase
//                                                                            newton_solve_kernel   (one workgroup per fit)
//   trial       log-posterior + gradient at x + s, x + s/2, x + s/4, x + s/8   launch_logp_grad
//   decision    accept / reject, lam update, convergence                     newton_accept_kernel
// The host only sequences these launches and reads the status records; fits of a batch advance together.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>

#include "bdrt_host.h"
#include "bdrt_newton_hess.h"

namespace bdrt {

constexpr int NW_NT = 512;
constexpr int NW_ATT = 6;           // dampings lam, 4 lam, ... 4^5 lam factored side by side when the machine has room (newton_solve_kernel)
constexpr int NW_TRY = 4;           // step lengths tried per factorisation: s, s/2, s/4, s/8 -- one batched evaluation

struct NewtonState {            // one per fit, in device memory
    double lp, lam, pred, lp_trial, grad_inf;
    double gs, ss;              // g.s and s.s of the current step: predicted increase of t s is  t g.s + t^2/2 (lam s.s - g.s)
    int iters, rc, done, need_hess, n_evals, max_iter, hbad, lin;       // hbad: non-finite Hessian entry seen by the build kernel; lin: coefficients on the linear scale (bdrt_newton_hess.h)
    double tol;
};

// what one factorisation attempt of a launch leaves for the gather kernel (up to NW_ATT attempts per fit when the launch speculates, see newton_solve_kernel)
struct NewtonAttempt { double pred, gs, ss, lam; int ok, pad; };

struct NewtonBufs {
    int D, Dp;                  // Dp = D rounded up to 16
    double *x, *g, *gt, *hstep;            // [n_fits][D]
    double *s, *xt;             // [n_fits][NW_ATT][D]: step and trial point of each attempt
    double *H, *M;              // [n_fits][Dp][Dp]
    double *M2;                 // [n_fits][NW_ATT - 1][Dp][Dp]: the speculative attempts' matrices (nullptr: never more than one attempt)
    NewtonAttempt *att;         // [n_fits][NW_ATT]
    double *probes, *pgrad;     // [n_fits][2 D][D]
    double *plp;                // [n_fits][2 D]
    NewtonState *st;            // [n_fits]
    long long *prof;            // BDRT_NEWTON_PROF=1: cycle counters of the solve kernel's phases (nullptr otherwise)
    // closed-form Hessian (bdrt_newton_hess.h; single series distribution): no probes, no probe gradients
    int analytic;
    double *hws;                // [n_fits][HessLayout::total]
    int hl_total, hl_tz, hl_act, hl_floor, o_x, Kx, lin_ok;   // layout offsets the solve / gather kernels need; lin_ok: Series_pos (the linear scale applies)
    const int *fspec;           // [n_fits] spectrum of each fit
    const DevProblem *dP;
};

// phase counters: thread 0 of workgroup 0 adds the core-clock cycles since the previous mark to slot k
struct NewtonProf {
    long long *p, t;
    __device__ NewtonProf(long long *p_) : p(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 ? p_ : nullptr), t(0) { if (p) t = clock64(); }
    __device__ void mark(int k) { if (p) { const long long n = clock64(); p[k] += n - t; t = n; } }
};
enum { NP_HESS = 0, NP_BUILD, NP_DIAG, NP_PANEL, NP_TRAIL, NP_SOLVE, NP_PRED, NP_WALL, NP_CALLS, NP_ATTEMPTS /* factorisations started */, NP_PREP0 = 16 /* .. 24: stages of hess_prep */, NP_FILL0 = 32 /* .. 34: stages of hess_fill */, NP_COUNT = 40 };

__global__ void newton_probe_kernel(NewtonBufs b, const int *active, int n_active)
{
    const int a = blockIdx.y, row = blockIdx.x;          // row in [0, 2 D)
    if (a >= n_active) return;
    const int f = active[a], D = b.D, j = row >> 1;
    if (b.st[f].done || !b.st[f].need_hess) return;      // (rejected trial: same Hessian; the probe rows keep whatever they hold)
    const double *x = b.x + (size_t)f * D;
    double *p = b.probes + ((size_t)a * 2 * D + row) * D;
    const double h = 1e-5 * fmax(1.0, fabs(x[j]));
    if (threadIdx.x == 0 && (row & 1) == 0) b.hstep[(size_t)f * D + j] = h;
    for (int k = threadIdx.x; k < D; k += blockDim.x) p[k] = x[k] + (k == j ? ((row & 1) ? -h : h) : 0.0);
}

__device__ __forceinline__ double bcast_lane(double v, int l)        // value of lane l (l uniform), as a scalar operand
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// In-place blocked Cholesky of the lower triangle of the Dp x Dp row-major matrix M (Dp a multiple of 16).  The forward
// substitution L y = v of a right-hand side v [Dp] (LDS) rides along: block J of y from the freshly factored diagonal
// block while it is in registers, the rows below from the panel while it is in LDS -- v holds y on return.
// lds: 16*17 (diagonal block) + 16 (reciprocal pivots) + Dp*17 (panel) + 1 (flag) doubles.  Returns false when not positive definite.
__device__ inline bool chol_blocked(double *M, int Dp, double *lds, NewtonProf &pf, double *v)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *Dg = lds, *rinv = Dg + 16 * 17, *Pn = rinv + 16;
    int *bad = reinterpret_cast<int *>(Pn + (size_t)Dp * 17);
    const int nblk = Dp / 16;
    if (tid == 0) *bad = 0;
    // Factor the 16 x 16 diagonal block that sits in Dg (row-major, stride 17) by the 16 rows' lanes of ONE wave, and the block J of the
    // forward substitution with it.  Lane i keeps row i of the block in registers; column j of the factor is read out of the lanes that
    // own it (v_readlane -> SGPR operand of the multiply-add): no LDS round trips, no branches inside the 16 steps; a non-positive pivot
    // poisons the block with NaNs and is reported after the last step.
    auto factor_diag = [&](int j0) {
        const int i = lane & 15;
        double row[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) row[k] = Dg[i * 17 + k];
        bool okb = true;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double d = bcast_lane(row[j], j);
            okb = okb && d > 0.0 && isfinite(d);
            double rs = __builtin_amdgcn_rsq(d);                      // 1 / sqrt(d): hardware estimate + two Newton steps
            rs = rs * (1.5 - 0.5 * d * rs * rs);
            rs = rs * (1.5 - 0.5 * d * rs * rs);
            const double lij = i == j ? d * rs : row[j] * rs;
            row[j] = lij;
#pragma unroll
            for (int k = j + 1; k < 16; ++k) {
                const double lkj = bcast_lane(lij, k);                // L[k][j], owned by lane k
                row[k] -= (i > j && k <= i) ? lij * lkj : 0.0;
            }
        }
        if (!okb) { if (lane == 0) *bad = 1; }
        else {
            // y_J = L11^-1 v_J: lane i holds row i of L11; y_k is broadcast as soon as it is known
            const double rd = 1.0 / row[i];
            double t = v[j0 + i];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double c = t * rd;
                const double yk = bcast_lane(c, k);
                if (i > k) t -= row[k] * yk;
                if (i == k) t = yk * row[i];                          // keeps t / L[i][i] = y_i for the store below
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; ++k) Dg[i * 17 + k] = row[k];
                rinv[i] = rd;
                v[j0 + i] = t * rd;
            }
        }
    };
    // block 0: from global memory; every later diagonal block is produced by the step before it (look-ahead, below)
    if (tid < 256) Dg[(tid >> 4) * 17 + (tid & 15)] = M[(size_t)(tid >> 4) * Dp + (tid & 15)];
    __syncthreads();
    if (wave == 0) factor_diag(0);
    __syncthreads();
    pf.mark(NP_DIAG);
    for (int J = 0; J < nblk; ++J) {
        const int j0 = 16 * J;
        if (*bad) return false;
        if (tid < 256 && (tid & 15) <= (tid >> 4)) M[(size_t)(j0 + (tid >> 4)) * Dp + j0 + (tid & 15)] = Dg[(tid >> 4) * 17 + (tid & 15)];
        // (b) panel below the block: x L11^T = a, one row per thread.  The m x 16 panel moves between global memory and
        // LDS with coalesced accesses (16 lanes per 128-byte row segment); the solve itself works on the LDS copy.
        const int m = Dp - j0 - 16;
        if (m == 0) break;
        {
            const double *src = M + (size_t)(j0 + 16 + (tid >> 4)) * Dp + j0 + (tid & 15);
            double *dst = Pn + (tid >> 4) * 17 + (tid & 15);
            const int nrow = NW_NT / 16;                                  // rows per pass
            int r = tid >> 4;
            for (; r + 3 * nrow < m; r += 4 * nrow) {                    // four row-passes per trip: the loads overlap
                const double v0 = src[0], v1 = src[(size_t)nrow * Dp], v2 = src[(size_t)2 * nrow * Dp], v3 = src[(size_t)3 * nrow * Dp];
                dst[0] = v0; dst[nrow * 17] = v1; dst[2 * nrow * 17] = v2; dst[3 * nrow * 17] = v3;
                src += (size_t)4 * nrow * Dp; dst += 4 * nrow * 17;
            }
            for (; r < m; r += nrow) { dst[0] = src[0]; src += (size_t)nrow * Dp; dst += nrow * 17; }
        }
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
                __builtin_amdgcn_sched_barrier(0);        // one row of L11 at a time: hoisting all 120 LDS operands costs 240 VGPRs
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) row[k] = xk[k];
            double dotv = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) dotv += xk[k] * v[j0 + k];
            v[j0 + 16 + r] -= dotv;
        }
        __syncthreads();
        for (int e = tid; e < m * 16; e += NW_NT) M[(size_t)(j0 + 16 + (e >> 4)) * Dp + j0 + (e & 15)] = Pn[(size_t)(e >> 4) * 17 + (e & 15)];
        pf.mark(NP_PANEL);
        // (c) trailing update C -= P P^T on 16 x 16 tiles of the lower triangle (tiles on the diagonal are computed in full).
        // Look-ahead: tile 0 is the next diagonal block -- wave 0 updates it into Dg (the panel above was the last reader of this
        // step's factor) and factors it at once, while the other seven waves update the rest of the trailing matrix: the 16
        // sequential steps of the diagonal block (8 k cycles, a quarter of the routine at D = 331) run in the shadow of the update.
        const int mb = m / 16, ntile = mb * (mb + 1) / 2;
        const int col = lane & 15, kq = lane >> 4;
        auto tile_of = [&](int t, int &I, int &Jc) {
            I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            while (I * (I + 1) / 2 > t) --I;
            Jc = t - I * (I + 1) / 2;
        };
        constexpr int NWV = NW_NT / 64, TPT = 4;
        if (wave == 0) {
            const double *C = M + (size_t)(j0 + 16 + kq) * Dp + j0 + 16 + col;
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = C[(size_t)(4 * r) * Dp];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mfma_f64(-Pn[(size_t)col * 17 + 4 * u + kq], Pn[(size_t)col * 17 + 4 * u + kq], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[(kq + 4 * r) * 17 + col] = acc[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            factor_diag(j0 + 16);
        } else {
            // four tiles per trip and wavefront: the global loads of all four are in flight before the first MFMA issues.
            // (Measured and dropped: six or eight tiles per trip -- 270 -> 290 / 317 us per launch at D = 331 --; two trips in flight,
            // the next one's loads requested before this one's MFMAs -- the registers it takes slow the panel and the substitutions
            // by more than the update gains: 270 -> 326 us.)
            for (int t = wave; t < ntile; t += TPT * (NWV - 1)) {
                double *C[TPT];
                int Ib[TPT], Jb[TPT];
                d4 acc[TPT];
#pragma unroll
                for (int q = 0; q < TPT; ++q) {
                    const int tq = t + q * (NWV - 1);
                    tile_of(tq < ntile ? tq : t, Ib[q], Jb[q]);
                    C[q] = M + (size_t)(j0 + 16 + 16 * Ib[q] + kq) * Dp + j0 + 16 + 16 * Jb[q] + col;
                }
#pragma unroll
                for (int q = 0; q < TPT; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[q][r] = C[q][(size_t)(4 * r) * Dp];
#pragma unroll
                for (int q = 0; q < TPT; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[q] = mfma_f64(-Pn[(size_t)(16 * Ib[q] + col) * 17 + 4 * u + kq], Pn[(size_t)(16 * Jb[q] + col) * 17 + 4 * u + kq], acc[q]);
#pragma unroll
                for (int q = 0; q < TPT; ++q)
                    if (q == 0 || t + q * (NWV - 1) < ntile) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) C[q][(size_t)(4 * r) * Dp] = acc[q][r];
                    }
            }
        }
        __syncthreads();
        pf.mark(NP_TRAIL);
    }
    return true;
}

__device__ __forceinline__ double sum16(double t)                  // sum over each aligned group of 16 lanes (valid in the group's lane 0)
{
    t += dpp_perm<0xB1>(t); t += dpp_perm<0x4E>(t); t += dpp_perm<0x141>(t); t += dpp_perm<0x140>(t);
    return t;
}

// v <- L^-T v (the forward substitution was done by chol_blocked), L = lower triangle of the row-major Dp x Dp matrix M
// (Dp a multiple of 16, identity on the padding); v [Dp] in LDS; Dg: 16*17 doubles of LDS scratch.  Blocked substitution:
// the 16 x 16 diagonal block by one wave with the block's columns in registers, the rest of the right-hand side is updated
// by the whole workgroup with coalesced row reads.
__device__ inline void chol_blocked_solve(const double *M, int Dp, double *v, double *Dg)
{
    const int tid = threadIdx.x;
    const int nblk = Dp / 16;
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
                const double xk = bcast_lane(c, k);
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

// H (symmetrised central differences of the probe gradients, when the fit asked for a fresh Hessian) and the damped
// matrix M = -H + lam I of every active fit: 16 rows of one fit per workgroup, so the 2 D^2 probe gradients are read at
// the whole chip's bandwidth instead of one CU's load latency (the solve kernel spent 0.4 M of its 1.9 M cycles here at D = 331).
__global__ __launch_bounds__(256) void newton_build_kernel(NewtonBufs b, const int *active, int n_active)
{
    const int a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = active[a], D = b.D, Dp = b.Dp;
    NewtonState &S = b.st[f];
    if (S.done) return;
    double *H = b.H + (size_t)f * Dp * Dp, *M = b.M + (size_t)f * Dp * Dp;
    const double lam = S.lam;
    const int r0 = 16 * blockIdx.x;
    if (S.need_hess) {
        // (probe slot a = position in the active list)
        const double *pg = b.pgrad + (size_t)a * 2 * D * D, *hs = b.hstep + (size_t)f * D;
        int bad = 0;
        for (int i = r0 + wv; i < r0 + 16; i += 4) {
            if (i >= D) {
                for (int k = lane; k < Dp; k += 64) M[(size_t)i * Dp + k] = i == k ? 1.0 : 0.0;
                continue;
            }
            const double ri = 0.5 / hs[i];
            for (int k = lane; k < Dp; k += 64) {
                double m = 0.0;
                if (k < D) {
                    const double hik = (pg[(size_t)(2 * i) * D + k] - pg[(size_t)(2 * i + 1) * D + k]) * ri;
                    const double hki = (pg[(size_t)(2 * k) * D + i] - pg[(size_t)(2 * k + 1) * D + i]) * (0.5 / hs[k]);
                    const double h = 0.5 * (hik + hki);
                    if (!isfinite(h)) bad = 1;
                    H[(size_t)i * Dp + k] = h;
                    m = -h + (i == k ? lam : 0.0);
                }
                M[(size_t)i * Dp + k] = m;
            }
        }
        if (__syncthreads_or(bad) && tid == 0) atomicOr(&S.hbad, 1);
    } else {
        for (int i = r0 + wv; i < r0 + 16; i += 4)
            for (int k = lane; k < Dp; k += 64) {
                double m = i == k ? 1.0 : 0.0;
                if (i < D && k < D) m = -H[(size_t)i * Dp + k] + (i == k ? lam : 0.0);
                M[(size_t)i * Dp + k] = m;
            }
    }
}

// The closed-form Hessian (bdrt_newton_hess.h) in place of probes + evaluator + newton_build_kernel: forward quantities and border
// vectors per fit, then H and M = -H + lam I in tiles of 16 rows.  A fit that keeps its Hessian (rejected trial) only gets M again.
__global__ __launch_bounds__(HP_NT) void newton_hess_prep_kernel(NewtonBufs b, const int *active, int n_active)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int f = active[blockIdx.x];
    NewtonState &S = b.st[f];
    if (S.done || !S.need_hess) return;
    // the switch to the linear scale of the coefficients: once, when the damping has dropped to 1e-4 (every later kernel of the round
    // reads the flag this kernel leaves)
    const int lin = (b.lin_ok && (S.lin || S.lam <= 1e-4)) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) S.lin = lin;
    HessArgs ha{b.dP, b.x, b.g, b.fspec, b.hws, b.H, b.M, b.D, b.Dp, lin, b.prof};
    const long long w0 = (b.prof && blockIdx.x == 0 && threadIdx.x == 0) ? wall_clock64() : 0;
    hess_prep(ha, f, sh);
    if (b.prof && blockIdx.x == 0 && threadIdx.x == 0) b.prof[NP_PREP0 + 9] += wall_clock64() - w0;
}
__global__ __launch_bounds__(HP_NT) void newton_hess_fill_kernel(NewtonBufs b, const int *active, int n_active)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int f = active[blockIdx.y], tid = threadIdx.x, D = b.D, Dp = b.Dp;
    NewtonState &S = b.st[f];
    if (S.done) return;
    const int r0 = 16 * blockIdx.x;
    if (S.need_hess) {
        HessArgs ha{b.dP, b.x, b.g, b.fspec, b.hws, b.H, b.M, b.D, b.Dp, S.lin, b.prof};
        const bool bad = hess_fill(ha, f, r0, S.lam, sh);
        if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicOr(&S.hbad, 1);
    } else {
        const double *H = b.H + (size_t)f * Dp * Dp;
        double *M = b.M + (size_t)f * Dp * Dp;
        const double lam = S.lam;
        for (int i = r0 + (tid >> 6); i < r0 + 16; i += HP_NT / 64)
            for (int k = tid & 63; k < Dp; k += 64) {
                double m = i == k ? 1.0 : 0.0;
                if (i < D && k < D) m = -H[(size_t)i * Dp + k] + (i == k ? lam : 0.0);
                M[(size_t)i * Dp + k] = m;
            }
    }
}

// One Levenberg-Marquardt step per active fit: the damped solve and the trial point.
// natt > 1 (natt workgroups per fit, blockIdx.y = attempt): four factorisations in ten at K = 161 find -H + lam I indefinite, and the
// next one with 4 lam is another 250 us on the one workgroup -- up to five in a row where the damping has to grow by orders of magnitude
// (the longest launches took 1.4 ms).  When the machine has room (natt workgroups per active fit <= CUs) attempt j factors
// -H + 4^j lam I at the same time on a CU of its own (its matrix built from H; the last attempt goes on to higher dampings by itself if
// it must), each once; the gather kernel takes the first attempt that succeeded: the same sequence of dampings and the same numbers as
// the one-workgroup loop, the failed factorisations off the critical path.  With natt = 1 attempt 0 runs the whole loop.
// The kernel only reads the fit's state; the gather kernel commits the chosen attempt.
__global__ __launch_bounds__(NW_NT) void newton_solve_kernel(NewtonBufs b, const int *active, int n_active, int natt)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int a = blockIdx.x, att = blockIdx.y, tid = threadIdx.x;
    const int f = active[a], D = b.D, Dp = b.Dp;
    const NewtonState &S = b.st[f];
    if (S.done) return;
    NewtonAttempt &A = b.att[NW_ATT * f + att];
    const double *H = b.H + (size_t)f * Dp * Dp;
    double *M = att ? b.M2 + ((size_t)f * (NW_ATT - 1) + (att - 1)) * Dp * Dp : b.M + (size_t)f * Dp * Dp;
    const double *x = b.x + (size_t)f * D, *g = b.g + (size_t)f * D;
    double *s = b.s + ((size_t)f * NW_ATT + att) * D, *xt = b.xt + ((size_t)f * NW_ATT + att) * D;
    const bool lin = b.analytic && S.lin;
    const double *hw = b.analytic ? b.hws + (size_t)f * b.hl_total : nullptr;
    double *v = sh;                               // [Dp] right-hand side / solution
    double *red = v + Dp;                         // 64 doubles
    double *cl = red + 64;                        // Cholesky scratch
    __shared__ int s_fin;
    NewtonProf pf(b.prof);
    const long long wall0 = b.prof ? wall_clock64() : 0;
    // H (when new probe gradients arrived) and M = -H + lam I were written by the build / fill kernel
    if (S.need_hess && S.hbad) { if (tid == 0) { A.ok = 0; A.lam = S.lam; } return; }       // (the gather kernel reports it)
    pf.mark(NP_HESS);
    double lam = S.lam;
    bool ok = false, rebuilt = true;
    if (att) { lam *= ldexp(1.0, 2 * att); rebuilt = false; }
    while (lam <= 1e12) {
        if (!rebuilt) {
            // more damping after a failed factorisation (or the speculative attempt's first matrix): M again from H -- the lower block
            // triangle only (what chol_blocked reads: diagonal tiles in full), four rows of a wave's column slice in flight at a time
            // (one dependent load per row was 160 k cycles per rebuild)
            const int wv = tid >> 6, ln = tid & 63;
            for (int i0 = 4 * wv; i0 < Dp; i0 += 4 * (NW_NT / 64)) {
                const int kend = ((i0 + 3) | 15) + 1;               // columns up to the end of the rows' diagonal tile
                for (int k = ln; k < kend; k += 64) {
                    double h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int i = i0 + r; h[r] = (i < D && k < D) ? H[(size_t)i * Dp + k] : 0.0; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + r;
                        double m = i == k ? 1.0 : 0.0;             // identity on the padding
                        if (i < D && k < D) m = -h[r] + (i == k ? lam : 0.0);
                        M[(size_t)i * Dp + k] = m;
                    }
                }
            }
            __syncthreads();
        }
        rebuilt = false;
        pf.mark(NP_BUILD);
        // right-hand side: the gradient in the iteration's coordinates (g_z = g_y dy/dz; zero for a coefficient held at its floor)
        for (int i = tid; i < Dp; i += NW_NT) {
            double gi = i < D ? g[i] : 0.0;
            if (lin && i < D) gi = hw[b.hl_act + i] != 0.0 ? 0.0 : gi * hw[b.hl_tz + i];
            v[i] = gi;
        }
        __syncthreads();
        if (pf.p) pf.p[NP_ATTEMPTS] += 1;
        if (chol_blocked(M, Dp, cl, pf, v)) {
            chol_blocked_solve(M, Dp, v, cl);
            pf.mark(NP_SOLVE);
            int fin = 1;
            for (int i = tid; i < D; i += NW_NT) {
                const double t = x[i] + v[i];              // (finite here is finite on either scale)
                if (!isfinite(t)) fin = 0;
            }
            if (tid == 0) s_fin = 1;
            __syncthreads();
            if (!fin) s_fin = 0;
            __syncthreads();
            if (s_fin) { ok = true; break; }
        }
        if (att + 1 < natt) break;                         // (the next damping is another workgroup's)
        lam *= 4.0;
        __syncthreads();
    }
    if (!ok) { if (tid == 0) { A.ok = 0; A.lam = lam; } return; }
    // trial point and the model's predicted increase g.s + 1/2 s^T H s.  With (-H + lam I) s = g this is
    // 1/2 (g.s + lam s.s): no product with H is needed (the difference is the residual of the backward-stable solve).
    double gs = 0.0, ss = 0.0;
    for (int i = tid; i < D; i += NW_NT) {
        const double si = v[i];
        double gi = g[i], xn = x[i] + si;
        if (lin) {
            gi = hw[b.hl_act + i] != 0.0 ? 0.0 : gi * hw[b.hl_tz + i];
            if (i >= b.o_x && i < b.o_x + b.Kx) xn = log(fmax(exp(x[i]) + si, hw[b.hl_floor]));     // projected step on the linear scale
        }
        s[i] = si; xt[i] = xn; gs += gi * si; ss += si * si;
    }
    const int lane = tid & 63, wave = tid >> 6;
    gs = sum32(gs); gs += __shfl_xor(gs, 32);
    ss = sum32(ss); ss += __shfl_xor(ss, 32);
    if (lane == 0) { red[wave] = gs; red[NW_NT / 64 + wave] = ss; }
    __syncthreads();
    if (tid == 0) {
        double p = 0.0, q = 0.0;
        for (int w = 0; w < NW_NT / 64; ++w) { p += red[w]; q += red[NW_NT / 64 + w]; }
        A.pred = 0.5 * (p + lam * q); A.lam = lam;
        A.gs = p; A.ss = q; A.ok = 1;
    }
    pf.mark(NP_PRED);
    if (pf.p) { pf.p[NP_WALL] += wall_clock64() - wall0; pf.p[NP_CALLS] += 1; }
}

// consume the trial evaluations (lp_t, gradients gt; rows NW_TRY a + t): the longest of the step lengths s, s/2, s/4, s/8 that
// passes the sufficient-increase test is accepted (bdrt_newton.h NewtonFit::consume, NEED_TRIAL, is the t = 0 case; the
// shorter steps along the same Levenberg-Marquardt direction cost one batched evaluation instead of a re-factorisation
// with more damping -- 40 % of the solves at K = 161 were such re-factorisations)
template <int NT>
__device__ __forceinline__ void newton_accept_body(const NewtonBufs &b, int f, int a, const double *lp_t, const double *trial)
{
    const int tid = threadIdx.x;
    const int D = b.D;
    NewtonState &S = b.st[f];
    __shared__ double red[NT];
    __shared__ int s_acc, s_t;
    double *x = b.x + (size_t)f * D, *g = b.g + (size_t)f * D;
    if (tid == 0) { s_acc = 0; s_t = 0; S.n_evals += NW_TRY; }
    __syncthreads();
    const bool lin = b.analytic && S.lin;
    const double floor_x = lin ? b.hws[(size_t)f * b.hl_total + b.hl_floor] : 0.0;
    for (int t = 0; t < NW_TRY; ++t) {
        const double *gt = b.gt + ((size_t)a * NW_TRY + t) * D;
        const double *xs_t = trial + ((size_t)a * NW_TRY + t) * D;
        const double lpn = lp_t[a * NW_TRY + t];
        int fin = isfinite(lpn) ? 1 : 0;
        double gi = 0.0;
        for (int j = tid; j < D; j += NT) {
            double v = gt[j];
            if (!isfinite(v)) fin = 0;
            // a coefficient on the linear scale that sits at its floor passes |g_y| = x |g_x| < tol whatever g_x is: there the test is
            // the bound's own condition, g_x <= 0 (a positive g_x, scaled by the largest coefficient, counts as gradient)
            if (lin && j >= b.o_x && j < b.o_x + b.Kx) {
                const double xj = exp(xs_t[j]);
                if (xj <= 4.0 * floor_x && v > 0.0) v = v / xj * (1e14 * floor_x);
            }
            gi = fmax(gi, fabs(v));
        }
        fin = __syncthreads_and(fin);
        red[tid] = gi;
        __syncthreads();
        for (int w = NT / 2; w > 0; w >>= 1) { if (tid < w) red[tid] = fmax(red[tid], red[tid + w]); __syncthreads(); }
        const double ginf = red[0];
        if (tid == 0) {
            const double lp = S.lp, sc = ldexp(1.0, -t);
            // predicted increase of the scaled step: with (-H + lam I) s = g,  t g.s + t^2/2 s^T H s = t g.s + t^2/2 (lam s.s - g.s)
            const double pred = t == 0 ? S.pred : sc * S.gs + 0.5 * sc * sc * (S.lam * S.ss - S.gs);
            int acc = fin && (lpn - lp >= 1e-4 * pred) && (lpn >= lp - 1e-12 * fabs(lp));
            // close to the optimum the predicted increase drops below the resolution of lp itself (a few ulp of |lp|): the
            // sufficient-increase test then compares rounding noise.  There the step is judged by what it is meant to do --
            // reduce the gradient -- as long as lp does not visibly decrease.
            const double noise = 64.0 * 2.220446049250313e-16 * fmax(1.0, fabs(lp));
            if (fin && !acc && pred < noise && lpn >= lp - noise && ginf < S.grad_inf) acc = 1;
            if (t == 0) S.lp_trial = lpn;
            if (acc) {
                const double rho = pred > 0.0 ? (lpn - lp) / pred : 0.0;
                S.lp = lpn;
                if (t == 0) {
                    if (rho > 0.75) S.lam = fmax(S.lam / 5.0, 1e-12);
                    else if (rho < 0.25) S.lam *= 2.0;
                } else {
                    S.lam *= ldexp(1.0, t);                       // the full step was too long: more damping next time
                }
                S.iters += 1;
                S.grad_inf = ginf;
                if (ginf < S.tol) { S.rc = 0; S.done = 1; }
                else if (S.iters >= S.max_iter) { S.rc = 1; S.done = 1; }
                else S.need_hess = 1;
                s_acc = 1; s_t = t;
            }
        }
        __syncthreads();
        if (s_acc) break;
    }
    if (s_acc) {
        const double *xs = trial + ((size_t)a * NW_TRY + s_t) * D, *gt = b.gt + ((size_t)a * NW_TRY + s_t) * D;
        for (int j = tid; j < D; j += NT) { x[j] = xs[j]; g[j] = gt[j]; }
    } else if (tid == 0) {
        S.lam *= 16.0;                                            // none of the lengths: same Hessian, much more damping
    }
}

__global__ __launch_bounds__(256) void newton_accept_kernel(NewtonBufs b, const int *active, int n_active, const double *lp_t,
                                                            const double *trial)
{
    const int f = active[blockIdx.x];
    if (b.st[f].done) return;
    newton_accept_body<256>(b, f, blockIdx.x, lp_t, trial);
}

// closed-form Hessian: the verdict on the trial points and, for a fit that goes on with a new point, the forward quantities of the NEXT
// round's Hessian (newton_hess_prep_kernel) in one launch: as a launch of its own -- one or two workgroups between two short kernels --
// the prep kernel took 48 us of which 25 us are its work (kernel timeline of a K = 161 fit, profiles/r06/newton_timeline.txt).
__global__ __launch_bounds__(HP_NT) void newton_accept_prep_kernel(NewtonBufs b, const int *active, int n_active, const double *lp_t,
                                                                const double *trial)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const long long w0 = (b.prof && blockIdx.x == 0 && threadIdx.x == 0) ? wall_clock64() : 0;
    const int f = active[blockIdx.x];
    NewtonState &S = b.st[f];
    if (S.done) return;
    newton_accept_body<HP_NT>(b, f, blockIdx.x, lp_t, trial);
    __syncthreads();                                       // the verdict (thread 0) and the new point are visible to the workgroup
    if (S.done || !S.need_hess) return;
    const int lin = (b.lin_ok && (S.lin || S.lam <= 1e-4)) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) S.lin = lin;
    HessArgs ha{b.dP, b.x, b.g, b.fspec, b.hws, b.H, b.M, b.D, b.Dp, lin, b.prof};
    hess_prep(ha, f, sh);
    if (b.prof) __syncthreads();                           // (profiling runs: the wall time of the whole workgroup, not of thread 0's wave)
    if (b.prof && blockIdx.x == 0 && threadIdx.x == 0) b.prof[NP_PREP0 + 9] += wall_clock64() - w0;
}

// dense batch of the active fits' trial points: rows NW_TRY a + t <- x + 2^-t s of fit active[a]
// (and the commit of the launch's solve: attempt 0 when it succeeded, else the speculative attempt 1 -- newton_solve_kernel)
__global__ void newton_gather_kernel(NewtonBufs b, const int *active, int n_active, double *dst, int natt)
{
    const int a = blockIdx.x, t = blockIdx.y;
    const int f = active[a];
    int c = 0;
    while (c + 1 < natt && !b.att[NW_ATT * f + c].ok) ++c;            // the first attempt that succeeded (the last one if none did)
    const double *x = b.x + (size_t)f * b.D, *sv = b.s + ((size_t)f * NW_ATT + c) * b.D, *xt = b.xt + ((size_t)f * NW_ATT + c) * b.D;
    if (t == 0 && threadIdx.x == 0) {
        NewtonState &S = b.st[f];
        if (!S.done) {
            const NewtonAttempt &A = b.att[NW_ATT * f + c];
            if (S.need_hess) {
                if (S.hbad) { S.rc = 2; S.done = 1; }
                else { S.need_hess = 0; S.n_evals += b.analytic ? 0 : 2 * b.D; }
            }
            if (!S.done) {
                if (A.ok) { S.pred = A.pred; S.lam = A.lam; S.gs = A.gs; S.ss = A.ss; }
                else { S.rc = 2; S.done = 1; S.lam = A.lam; }
            }
        }
    }
    const double sc = ldexp(1.0, -t);
    double *d = dst + ((size_t)a * NW_TRY + t) * b.D;
    const bool lin = b.analytic && b.st[f].lin;
    const double floor_x = lin ? b.hws[(size_t)f * b.hl_total + b.hl_floor] : 0.0;
    for (int k = threadIdx.x; k < b.D; k += blockDim.x) {
        double v = t == 0 ? xt[k] : x[k] + sc * sv[k];
        // a coefficient on the linear scale: the shortened step is shortened there, y = log(max(x + 2^-t s, floor))
        if (lin && t > 0 && k >= b.o_x && k < b.o_x + b.Kx) v = log(fmax(exp(x[k]) + sc * sv[k], floor_x));
        d[k] = v;
    }
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
    const auto t_host0 = std::chrono::steady_clock::now();
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
    NW_HIP(alloc(nD, (void **)&b.x)); NW_HIP(alloc(nD, (void **)&b.g)); NW_HIP(alloc(NW_ATT * nD, (void **)&b.s));
    NW_HIP(alloc(NW_ATT * nD, (void **)&b.xt)); NW_HIP(alloc(nD * NW_TRY, (void **)&b.gt)); NW_HIP(alloc(nD, (void **)&b.hstep));
    NW_HIP(alloc((size_t)n_fits * NW_ATT * sizeof(NewtonAttempt), (void **)&b.att));
    NW_HIP(hipMemsetAsync(b.att, 0, (size_t)n_fits * NW_ATT * sizeof(NewtonAttempt), P.stream));
    NW_HIP(alloc((size_t)n_fits * Dp * Dp * sizeof(double), (void **)&b.H));
    NW_HIP(alloc((size_t)n_fits * Dp * Dp * sizeof(double), (void **)&b.M));
    // the speculative factorisations (newton_solve_kernel) need matrices of their own: allocated while they stay below 2 GB (a 512-spectrum batch with them -- 2.3 GB of hipMalloc -- was 0.31 -> 0.38 s), used in the
    // launches that have the workgroups to spare (BDRT_NEWTON_SPEC=0: never)
    b.M2 = nullptr;
    int n_cu = 0;
    NW_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, P.device));
    const char *spec_env = getenv("BDRT_NEWTON_SPEC");
    if (!(spec_env && spec_env[0] == '0') && (size_t)n_fits * (NW_ATT - 1) * Dp * Dp * sizeof(double) <= ((size_t)2 << 30))
        NW_HIP(alloc((size_t)n_fits * (NW_ATT - 1) * Dp * Dp * sizeof(double), (void **)&b.M2));
    // closed-form Hessian where the model has one (BDRT_NEWTON_FD=1: central differences of the gradient as in rounds 1-5)
    const char *fd_env = getenv("BDRT_NEWTON_FD");
    b.analytic = (hess_analytic_ok(P.dev) && !(fd_env && fd_env[0] == '1')) ? 1 : 0;
    b.hws = nullptr; b.fspec = nullptr; b.dP = (const DevProblem *)P.d_dev;
    const HessLayout hlay(P.dev.nf, P.dev.blk[0].K);
    b.hl_total = hlay.total; b.hl_tz = hlay.tz; b.hl_act = hlay.act; b.hl_floor = hlay.h0 + 11; b.o_x = P.dev.blk[0].o_x; b.Kx = P.dev.blk[0].K;
    const char *lin_env = getenv("BDRT_NEWTON_LINEAR");      // 0: the coefficients stay on the log scale for the whole iteration
    b.lin_ok = (b.analytic && P.dev.blk[0].is_pos && !(lin_env && lin_env[0] == '0')) ? 1 : 0;
    if (b.analytic) {
        // (the probe buffer only carries the trial points of a round)
        NW_HIP(alloc((size_t)n_fits * NW_TRY * D * sizeof(double), (void **)&b.probes));
        b.pgrad = nullptr; b.plp = nullptr;
        NW_HIP(alloc((size_t)n_fits * hlay.total * sizeof(double), (void **)&b.hws));
    } else {
        NW_HIP(alloc((size_t)n_fits * 2 * D * D * sizeof(double), (void **)&b.probes));
        NW_HIP(alloc((size_t)n_fits * 2 * D * D * sizeof(double), (void **)&b.pgrad));
        NW_HIP(alloc((size_t)n_fits * 2 * D * sizeof(double), (void **)&b.plp));
    }
    NW_HIP(alloc((size_t)n_fits * sizeof(NewtonState), (void **)&b.st));
    b.prof = nullptr;
    const char *prof_env = getenv("BDRT_NEWTON_PROF");
    if (prof_env && prof_env[0] == '1') {
        NW_HIP(alloc(NP_COUNT * sizeof(long long), (void **)&b.prof));
        NW_HIP(hipMemset(b.prof, 0, NP_COUNT * sizeof(long long)));
        NW_HIP(hipStreamSynchronize(nullptr));          // the fill is asynchronous; the kernels run on the problem's own stream
    }
    int *d_active = nullptr, *d_spec1 = nullptr, *d_specp = nullptr, *d_fspec = nullptr;
    double *d_lpt = nullptr;
    NW_HIP(alloc((size_t)n_fits * sizeof(int), (void **)&d_active));
    NW_HIP(alloc((size_t)n_fits * NW_TRY * sizeof(int), (void **)&d_spec1));
    NW_HIP(alloc((size_t)n_fits * (b.analytic ? 1 : 2 * D) * sizeof(int), (void **)&d_specp));
    NW_HIP(alloc((size_t)n_fits * sizeof(int), (void **)&d_fspec));
    b.fspec = d_fspec;
    NW_HIP(alloc((size_t)n_fits * NW_TRY * sizeof(double), (void **)&d_lpt));
    std::vector<NewtonState> hst((size_t)n_fits);
    for (auto &s : hst) { memset(&s, 0, sizeof(s)); s.lam = 1e-3; s.max_iter = max_iter; s.tol = tol; s.rc = 1; s.done = max_iter > 0 ? 0 : 1; }
    NW_HIP(hipMemcpy(b.st, hst.data(), hst.size() * sizeof(NewtonState), hipMemcpyHostToDevice));
    NW_HIP(hipMemcpy(b.x, x0, nD, hipMemcpyHostToDevice));
    hipStream_t st = P.stream;
    const auto t_host1 = std::chrono::steady_clock::now();
    // evaluation at the start points
    std::vector<int> hspec((size_t)n_fits), hact((size_t)n_fits), hspecp, sp1((size_t)n_fits * NW_TRY);
    for (int i = 0; i < n_fits; ++i) hspec[i] = spec ? spec[i] : 0;
    NW_HIP(hipMemcpyAsync(d_spec1, hspec.data(), (size_t)n_fits * sizeof(int), hipMemcpyHostToDevice, st));
    NW_HIP(hipMemcpyAsync(d_fspec, hspec.data(), (size_t)n_fits * sizeof(int), hipMemcpyHostToDevice, st));
    const size_t lds_prep = hess_prep_lds_doubles(P.dev.nf, P.dev.blk[0].K) * sizeof(double), lds_fill = hess_fill_lds_doubles(P.dev.nf, P.dev.blk[0].K) * sizeof(double);
    if (b.analytic) {
        NW_HIP(hipFuncSetAttribute((const void *)newton_hess_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
        NW_HIP(hipFuncSetAttribute((const void *)newton_accept_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
        NW_HIP(hipFuncSetAttribute((const void *)newton_hess_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill));
    }
    int rc;
    if ((rc = launch_logp_grad(&P, b.x, d_spec1, n_fits, 0, d_lpt, b.gt, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
    hipLaunchKernelGGL(newton_init_kernel, dim3(n_fits), dim3(256), 0, st, b, n_fits, (const double *)d_lpt, (const double *)b.gt);
    const size_t lds_solve = ((size_t)Dp + 64 + 16 * 17 + 16 + (size_t)Dp * 17 + 2) * sizeof(double);
    NW_HIP(hipFuncSetAttribute((const void *)newton_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
    // The sequence of launches of one Levenberg-Marquardt round does not depend on what the round decides (kernels skip the
    // fits that are done; a fit that keeps its Hessian after a rejected trial skips the probe writes and ignores the probe
    // gradients), so several rounds are enqueued back to back and the host reads the status records once per group.
    const char *tf_env = getenv("BDRT_NEWTON_TRIAL_FEW");      // 0: the trial points through the tile evaluator (as before round 6)
    const bool trial_few = !(tf_env && tf_env[0] == '0');
    const char *rp_env = getenv("BDRT_NEWTON_ROUNDS");
    const char *trace_env = getenv("BDRT_NEWTON_TRACE");      // 1: the status records of the first eight fits at every host synchronisation, on stderr
    const int rounds_per_sync = std::max(1, rp_env ? atoi(rp_env) : 4);
    const long long max_rounds = (long long)max_iter * 40 + 100;
    // A host synchronisation costs ~90 us (status records down, the active list and its spectrum ids up): the lists go up only when they
    // changed, and while some active fit is far from its stationary point (|g| > 0.01: at least three more rounds) a group is twice as long
    std::vector<int> hact_prev;
    int group = rounds_per_sync;
    for (long long round = 0; round < max_rounds; round += group) {
        NW_HIP(hipMemcpyAsync(hst.data(), b.st, hst.size() * sizeof(NewtonState), hipMemcpyDeviceToHost, st));
        NW_HIP(hipStreamSynchronize(st));
        int n_active = 0;
        double g_far = 0.0;                                    // (the fit that will finish last decides how long the iteration runs)
        for (int i = 0; i < n_fits; ++i) if (!hst[i].done) { hact[n_active++] = i; g_far = std::max(g_far, hst[i].grad_inf); }
        // (small batches only: a fit that finishes inside a group keeps its rows in the evaluator's batch until the next synchronisation)
        group = (rp_env || !(g_far > 1e-2) || n_active > 32) ? rounds_per_sync : 2 * rounds_per_sync;
        if (trace_env && trace_env[0] == '1')
            for (int i = 0; i < n_fits && i < 8; ++i)
                fprintf(stderr, "[bdrt newton trace] round %lld fit %d spec %d: lp %.12g lam %.3g |g| %.3g iters %d done %d rc %d lin %d\n", round, i, hspec[i],
                        hst[i].lp, hst[i].lam, hst[i].grad_inf, hst[i].iters, hst[i].done, hst[i].rc, hst[i].lin);
        if (n_active == 0) break;
        if ((int)hact_prev.size() != n_active || !std::equal(hact_prev.begin(), hact_prev.end(), hact.begin())) {
            hact_prev.assign(hact.begin(), hact.begin() + n_active);
            NW_HIP(hipMemcpyAsync(d_active, hact.data(), (size_t)n_active * sizeof(int), hipMemcpyHostToDevice, st));
            if (!b.analytic) hspecp.resize((size_t)n_active * 2 * D);
            for (int a = 0; a < n_active; ++a) {
                if (!b.analytic) std::fill(hspecp.begin() + (size_t)a * 2 * D, hspecp.begin() + (size_t)(a + 1) * 2 * D, hspec[hact[a]]);
                for (int t = 0; t < NW_TRY; ++t) sp1[(size_t)a * NW_TRY + t] = hspec[hact[a]];
            }
            if (!b.analytic) NW_HIP(hipMemcpyAsync(d_specp, hspecp.data(), hspecp.size() * sizeof(int), hipMemcpyHostToDevice, st));
            NW_HIP(hipMemcpyAsync(d_spec1, sp1.data(), (size_t)n_active * NW_TRY * sizeof(int), hipMemcpyHostToDevice, st));
        }
        const int natt = b.M2 ? std::max(1, std::min(NW_ATT, n_cu / n_active)) : 1;
        for (int r = 0; r < group; ++r) {
            // probes of the fits that ask for a fresh Hessian (probe slot a = position in the active list) and their gradients
            if (b.analytic) {
                // (the forward quantities of this round's Hessian: left by the previous round's accept + prep kernel, by a launch of
                //  their own in the first round)
                if (round == 0 && r == 0)
                    hipLaunchKernelGGL(newton_hess_prep_kernel, dim3(n_active), dim3(HP_NT), lds_prep, st, b, (const int *)d_active, n_active);
                hipLaunchKernelGGL(newton_hess_fill_kernel, dim3(Dp / 16, n_active), dim3(HP_NT), lds_fill, st, b, (const int *)d_active, n_active);
            } else {
                hipLaunchKernelGGL(newton_probe_kernel, dim3(2 * D, n_active), dim3(128), 0, st, b, (const int *)d_active, n_active);
                if ((rc = launch_logp_grad(&P, b.probes, d_specp, n_active * 2 * D, 0, b.plp, b.pgrad, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
                hipLaunchKernelGGL(newton_build_kernel, dim3(Dp / 16, n_active), dim3(256), 0, st, b, (const int *)d_active, n_active);
            }
            hipLaunchKernelGGL(newton_solve_kernel, dim3(n_active, natt), dim3(NW_NT), lds_solve, st, b, (const int *)d_active, n_active, natt);
            // trial points (NW_TRY step lengths per fit) as a dense batch for the evaluator, then the verdict
            hipLaunchKernelGGL(newton_gather_kernel, dim3(n_active, NW_TRY), dim3(128), 0, st, b, (const int *)d_active, n_active, b.probes, natt);
            // (the one-workgroup-per-point evaluator where the family has one: 8 us against the tile evaluator's 24 for the eight points of
            //  a two-start fit; the same evaluator for every batch size -- a fit's numbers do not depend on its batch)
            rc = trial_few ? launch_logp_grad_few(&P, b.probes, d_spec1, n_active * NW_TRY, 0, d_lpt, b.gt, st, 1) : 1;
            if (rc < 0) { cleanup(); return rc; }
            if (rc == 1 && (rc = launch_logp_grad(&P, b.probes, d_spec1, n_active * NW_TRY, 0, d_lpt, b.gt, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
            if (b.analytic)
                hipLaunchKernelGGL(newton_accept_prep_kernel, dim3(n_active), dim3(HP_NT), lds_prep, st, b, (const int *)d_active, n_active,
                                   (const double *)d_lpt, (const double *)b.probes);
            else
                hipLaunchKernelGGL(newton_accept_kernel, dim3(n_active), dim3(256), 0, st, b, (const int *)d_active, n_active, (const double *)d_lpt,
                                   (const double *)b.probes);
        }
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
    if (b.prof) {
        long long hp[NP_COUNT];
        NW_HIP(hipMemcpy(hp, b.prof, sizeof(hp), hipMemcpyDeviceToHost));
        const double n = hp[NP_CALLS] ? (double)hp[NP_CALLS] : 1.0;
        fprintf(stderr, "[bdrt newton prof] D %d, %lld solve launches (workgroup 0); core cycles per launch: hessian %.0f, build %.0f, "
                "chol diag %.0f / panel %.0f / trailing %.0f, solve %.0f, pred %.0f; wall %.1f us per launch\n", D, hp[NP_CALLS],
                hp[NP_HESS] / n, hp[NP_BUILD] / n, hp[NP_DIAG] / n, hp[NP_PANEL] / n, hp[NP_TRAIL] / n, hp[NP_SOLVE] / n, hp[NP_PRED] / n,
                hp[NP_WALL] / n / 100.0);
        fprintf(stderr, "[bdrt newton prof] factorisations started per launch: %.2f\n", hp[NP_ATTEMPTS] / n);
        fprintf(stderr, "[bdrt newton prof] closed-form Hessian, cycles per launch: prep");
        for (int k = NP_PREP0; k <= NP_PREP0 + 8; ++k) fprintf(stderr, " %.0f", hp[k] / n);
        fprintf(stderr, "; fill (row tile 8): stage %.0f, dense product %.0f, entries + stores %.0f; prep wall %.1f us\n", hp[NP_FILL0] / n, hp[NP_FILL0 + 1] / n, hp[NP_FILL0 + 2] / n, hp[NP_PREP0 + 9] / n / 100.0);
    }
#undef NW_HIP
    const auto t_host2 = std::chrono::steady_clock::now();
    cleanup();
    if (prof_env && prof_env[0] == '1') {
        const auto t_host3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "[bdrt newton prof] host: allocations + uploads %.0f us, iteration %.0f us, frees %.0f us\n", us(t_host0, t_host1), us(t_host1, t_host2),
                us(t_host2, t_host3));
    }
    return 0;
}

// the closed-form Hessian at ONE point (tests): H_out [D][D] row-major on the host; 1 when the problem has no closed form here
int hessian_at_point(Problem &P, const double *theta, int spec, double *H_out)
{
    if (!hess_analytic_ok(P.dev)) return 1;
    const int D = P.dev.D, Dp = (D + 15) & ~15;
    BDRT_HIP(hipSetDevice(P.device));
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *p : owned) hipFree(p); };
#define HP_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    auto alloc = [&](size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 8);
        if (e == hipSuccess) owned.push_back(*d);
        return e;
    };
    NewtonBufs b;
    memset(&b, 0, sizeof(b));
    b.D = D; b.Dp = Dp; b.analytic = 1; b.dP = (const DevProblem *)P.d_dev;
    const HessLayout hlay(P.dev.nf, P.dev.blk[0].K);
    int *d_act = nullptr, *d_spec = nullptr;
    double *d_lp = nullptr;
    HP_HIP(alloc((size_t)D * sizeof(double), (void **)&b.x)); HP_HIP(alloc((size_t)D * sizeof(double), (void **)&b.g));
    HP_HIP(alloc((size_t)Dp * Dp * sizeof(double), (void **)&b.H)); HP_HIP(alloc((size_t)Dp * Dp * sizeof(double), (void **)&b.M));
    HP_HIP(alloc((size_t)hlay.total * sizeof(double), (void **)&b.hws)); HP_HIP(alloc(sizeof(NewtonState), (void **)&b.st));
    HP_HIP(alloc(sizeof(int), (void **)&d_act)); HP_HIP(alloc(sizeof(int), (void **)&d_spec)); HP_HIP(alloc(sizeof(double), (void **)&d_lp));
    b.fspec = d_spec;
    NewtonState hs;
    memset(&hs, 0, sizeof(hs)); hs.need_hess = 1; hs.lam = 0.0;
    const int zero = 0;
    hipStream_t st = P.stream;
    HP_HIP(hipMemcpyAsync(b.st, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    HP_HIP(hipMemcpyAsync(b.x, theta, (size_t)D * sizeof(double), hipMemcpyHostToDevice, st));
    HP_HIP(hipMemcpyAsync(d_act, &zero, sizeof(int), hipMemcpyHostToDevice, st));
    HP_HIP(hipMemcpyAsync(d_spec, &spec, sizeof(int), hipMemcpyHostToDevice, st));
    int rc;
    if ((rc = launch_logp_grad(&P, b.x, d_spec, 1, 0, d_lp, b.g, nullptr, nullptr, nullptr, st))) { cleanup(); return rc; }
    const size_t lds_prep = hess_prep_lds_doubles(P.dev.nf, P.dev.blk[0].K) * sizeof(double), lds_fill = hess_fill_lds_doubles(P.dev.nf, P.dev.blk[0].K) * sizeof(double);
    HP_HIP(hipFuncSetAttribute((const void *)newton_hess_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
    HP_HIP(hipFuncSetAttribute((const void *)newton_hess_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill));
    hipLaunchKernelGGL(newton_hess_prep_kernel, dim3(1), dim3(HP_NT), lds_prep, st, b, (const int *)d_act, 1);
    hipLaunchKernelGGL(newton_hess_fill_kernel, dim3(Dp / 16, 1), dim3(HP_NT), lds_fill, st, b, (const int *)d_act, 1);
    HP_HIP(hipGetLastError());
    std::vector<double> hh((size_t)Dp * Dp);
    HP_HIP(hipMemcpyAsync(hh.data(), b.H, hh.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HP_HIP(hipStreamSynchronize(st));
    // (the kernels write the block lower triangle: mirror it)
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) H_out[(size_t)i * D + j] = H_out[(size_t)j * D + i] = hh[(size_t)i * Dp + j];
#undef HP_HIP
    cleanup();
    return 0;
}

}  // namespace bdrt
