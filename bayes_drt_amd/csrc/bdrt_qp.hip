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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bdrt_host.h"

namespace bdrt {

constexpr int QP_NT = 512;
// BDRT_QP_PROF=1: core-clock cycles of workgroup 0's phases (form KKT, factor, triangular solves, everything else), printed per call
__device__ long long g_qp_prof[8];
__device__ int g_qp_prof_on;
struct QpProf {
    long long t; bool on;
    __device__ QpProf() : t(0), on(g_qp_prof_on && blockIdx.x == 0 && threadIdx.x == 0) { if (on) t = clock64(); }
    __device__ void mark(int k) { if (on) { const long long n = clock64(); g_qp_prof[k] += n - t; t = n; } }
};
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
__device__ __forceinline__ int cidx(int i, int j, int n) { return j * n - ((j * (j - 1)) >> 1) + (i - j); }      // (n < 46 000: 32-bit index arithmetic)

// In-place Cholesky of the packed lower triangle; diag[j] receives L(j,j) (M(j,j) keeps the pivot).  Returns false
// (uniformly) when a pivot is not positive.  All threads of the workgroup must call.  Right-looking in blocks of 16 columns
// (round 4; round 2's form updated the whole trailing triangle after every column -- two workgroup barriers and a pass over up to
// n^2 / 2 LDS cells per column: 570 k cycles at n = 163, 60 % of an interior-point iteration):
//   (a) the 16 x 16 diagonal block by wavefront 0 on a register copy, lane r = row r; the scaled column reaches the other lanes
//       through a 16-double LDS buffer (one round trip per column);
//   (b) the rows below by forward substitution against the block, one row per thread;
//   (c) the trailing triangle on v_mfma_f64_16x16x4 tiles, operands and accumulators straight from the packed triangle.
// Three barriers per block.  `scr`: 32 doubles of scratch.
// LDSM: M is in LDS -- addressed as such (through a generic pointer every access is a flat load: several times the latency)
template <bool LDSM>
__device__ __noinline__ bool chol_packed(double *M, double *diag, int n, double *scr)
{
    typedef __attribute__((address_space(3))) double *lds_dptr;
    auto LD = [&](int idx) -> double { if constexpr (LDSM) return ((lds_dptr)M)[idx]; else return M[idx]; };
    auto ST = [&](int idx, double val) { if constexpr (LDSM) ((lds_dptr)M)[idx] = val; else M[idx] = val; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_bad;
    __shared__ double s_inv[16];
    __shared__ double s_tile[16 * 17];
    int *bad = &s_bad;
    if (tid == 0) *bad = 0;
    __syncthreads();
    QpProf pq;
    for (int j0 = 0; j0 < n; j0 += 16) {
        const int jb = min(16, n - j0), m = n - j0 - jb;
        pq.mark(7);
        // (a) the block in a 16 x 17 LDS tile, thread (r, k) of the first 256 owns element (r, k), k <= r.  Column c: every thread
        //     reads the pivot and its two column-c entries as they stand, scales them itself (the columns stay unscaled in the
        //     tile until the end: nobody writes what another thread reads in the same step) and updates its own element --
        //     one barrier per column.
        {
            const int r = tid >> 4, k = tid & 15;
            const bool mine = tid < 256 && r < jb && k <= r;
            if (mine) s_tile[r * 17 + k] = LD(cidx(j0 + r, j0 + k, n));
            __syncthreads();
            bool ok = true;
            double mineval = mine ? s_tile[r * 17 + k] : 0.0;
            for (int c = 0; c < jb; ++c) {
                const double d = s_tile[c * 17 + c];
                ok = ok && d > 0.0 && isfinite(d);
                // 1 / sqrt(d): hardware estimate + two Newton steps
                double inv = __builtin_amdgcn_rsq(d);
                inv = inv * (1.5 - 0.5 * d * inv * inv);
                inv = inv * (1.5 - 0.5 * d * inv * inv);
                if (tid == c) { diag[j0 + c] = d * inv; s_inv[c] = inv; }
                if (mine && k > c) {
                    const double lr = s_tile[r * 17 + c] * inv, lk = s_tile[k * 17 + c] * inv;
                    mineval -= lr * lk;
                    s_tile[r * 17 + k] = mineval;
                }
                __syncthreads();
            }
            if (!ok) { if (tid == 0) *bad = 1; }
            else if (mine) ST(cidx(j0 + r, j0 + k, n), k < r ? mineval * s_inv[k] : mineval);     // (the diagonal keeps the pivot)
        }
        __syncthreads();
        pq.mark(5);
        if (*bad) return false;
        if (m > 0) {
            // (b) x_c = (a_c - sum_{k < c} x_k L(c, k)) / L(c, c), one row per thread (jb = 16 here)
            for (int i = j0 + 16 + tid; i < n; i += QP_NT) {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = LD(cidx(i, j0 + c, n));
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double t = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) t -= x[k] * LD(cidx(j0 + c, j0 + k, n));
                    x[c] = t * s_inv[c];
                    __builtin_amdgcn_sched_barrier(0);      // one row of the block at a time (else all 120 operands are requested up front)
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) ST(cidx(i, j0 + c, n), x[c]);
            }
            __syncthreads();
            pq.mark(6);
            // (c) C(i, k) -= sum_c X(i, c) X(k, c) on the tiles of the lower triangle of the trailing part
            const int r0 = j0 + 16, mb = (m + 15) >> 4, ntile = mb * (mb + 1) / 2;
            const int col = lane & 15, kq = lane >> 4;
            for (int t = wave; t < ntile; t += QP_NT / 64) {
                int I = (int)((__fsqrt_rn((float)(8 * t + 1)) - 1.0f) * 0.5f);
                while ((I + 1) * (I + 2) / 2 <= t) ++I;
                while (I * (I + 1) / 2 > t) --I;
                const int Jc = t - I * (I + 1) / 2;
                const int R0 = r0 + 16 * I, C0 = r0 + 16 * Jc;
                const int ia = R0 + col, ib = C0 + col;                     // operand rows of this lane
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xa[u] = ia < n ? LD(cidx(ia, j0 + 4 * u + kq, n)) : 0.0;
                    xb[u] = ib < n ? LD(cidx(ib, j0 + 4 * u + kq, n)) : 0.0;
                }
                d4 acc;
                bool own[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ri = R0 + kq + 4 * rr, ci = C0 + col;
                    own[rr] = ri < n && ci <= ri;
                    acc[rr] = own[rr] ? LD(cidx(ri, ci, n)) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma_f64(-xa[u], xb[u], acc);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (own[rr]) ST(cidx(R0 + kq + 4 * rr, C0 + col, n), acc[rr]);
            }
            __syncthreads();
        }
    }
    return true;
}

// v <- (L L^T)^-1 v (L strictly-lower part in M, diagonal in diag); ends with a workgroup barrier.  Blocked by 16 unknowns:
// the 16 x 16 triangle of a block by 16 lanes of wavefront 0 with the block's right-hand side in registers (the unknown just found
// goes to the other lanes through v_readlane), the rest of the right-hand side by the whole workgroup -- forward one row per
// thread, backward 32 threads per column.  (Round 2's form did both substitutions with one wavefront, two wave-level exchanges and a
// division per unknown: about 40 % of an interior-point iteration at n = 163.)  rd: 1 / diag, scratch of n doubles; tb: 16 doubles.
template <bool LDSM>
__device__ __noinline__ void chol_solve_blocked(const double *M, const double *diag, int n, double *v, double *rd, double *tb)
{
    typedef const __attribute__((address_space(3))) double *lds_cdptr;
    auto LD = [&](int idx) -> double { if constexpr (LDSM) return ((lds_cdptr)M)[idx]; else return M[idx]; };
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < n; i += QP_NT) rd[i] = 1.0 / diag[i];
    __syncthreads();
    const int nb = (n + 15) / 16;
    auto bcast = [](double y, int c) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), c), __builtin_amdgcn_readlane(__double2loint(y), c)); };
    for (int J = 0; J < nb; ++J) {                                      // forward: L y = v
        const int j0 = 16 * J, jb = min(16, n - j0);
        if (tid < 64) {
            // lane r: row j0 + r of the block's triangle in registers (zeros outside it), its own reciprocal pivot
            const int r = lane & 15, i = j0 + r;
            const bool in = r < jb;
            double Lr[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Lr[c] = (in && c < r) ? LD(cidx(i, j0 + c, n)) : 0.0;
            const double rdo = in ? rd[i] : 1.0;
            double t = in ? v[i] : 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double yc = bcast(t * rdo, c);
                t = r == c ? yc : t - Lr[c] * yc;
            }
            if (lane < jb) v[i] = t;
        }
        __syncthreads();
        for (int i = j0 + jb + tid; i < n; i += QP_NT) {
            // (jb = 16 here: only the last block is shorter, and nothing lies below it; all sixteen requests go out together)
            double Lrow[16], yv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { Lrow[c] = LD(cidx(i, j0 + c, n)); yv[c] = v[j0 + c]; }
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) t += Lrow[c] * yv[c];
            v[i] -= t;
        }
        __syncthreads();
    }
    for (int J = nb - 1; J >= 0; --J) {                                 // backward: L^T x = y
        const int j0 = 16 * J, jb = min(16, n - j0);
        {
            const int c = tid >> 5, l32 = tid & 31;                     // 32 threads per column of the block
            double t = 0.0;
            if (c < jb) {
                const int cj = cidx(j0 + c, j0 + c, n);                 // LD(cj + i - j) = L(i, j)
                for (int i = j0 + jb + l32; i < n; i += 32) t += LD(cj + i - (j0 + c)) * v[i];
            }
            t += __shfl_xor(t, 16); t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
            if (l32 == 0 && c < 16) tb[c] = t;
        }
        __syncthreads();
        if (tid < 64) {
            // lane r: column j0 + r of the block's triangle (row r of its transpose)
            const int r = lane & 15, j = j0 + r;
            const bool in = r < jb;
            double Lc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Lc[c] = (in && c > r && c < jb) ? LD(cidx(j0 + c, j, n)) : 0.0;
            const double rdo = in ? rd[j] : 1.0;
            double u = in ? v[j] - tb[r] : 0.0;
#pragma unroll
            for (int c = 15; c >= 0; --c) {
                const double xc = bcast(u * rdo, c);
                u = r == c ? xc : u - Lc[c] * xc;
            }
            if (lane < jb) v[j] = u;
        }
        __syncthreads();
    }
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
    QpProf pf;
    auto factor = [&](bool unit) -> bool {
        double reg = 0.0;
        bool ok = false;
        while (true) {
            __syncthreads();
            pf.mark(3);
            for (int j = wave; j < n; j += QP_NT / 64) {              // one wavefront per column of the lower triangle
                double *colj = M + cidx(j, j, n);
                for (int i = j + lane; i < n; i += 64) {
                    double v = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
                    if (i == j) v += (bd[i] != 0.0 ? (unit ? 1.0 : z[i] / s[i]) : 0.0) + reg;
                    colj[i - j] = v;
                }
            }
            __syncthreads();
            pf.mark(0);
            const double m00 = M[0] - reg;
            ok = chol_packed<LDSM>(M, diag, n, red);
            pf.mark(1);
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
    pf.mark(3);
    chol_solve_blocked<LDSM>(M, diag, n, x, rhs, red);
    pf.mark(2);
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
            pf.mark(3);
            chol_solve_blocked<LDSM>(M, diag, n, dx, Px, red);
            pf.mark(2);
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
    pf.mark(3);
    if (pf.on) g_qp_prof[4] += it;
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
    const bool in_lds = vec_bytes + msize * sizeof(double) <= 160 * 1024 - 2560;      // (2.5 KB of static LDS in chol_packed)
    const size_t lds = in_lds ? vec_bytes + msize * sizeof(double) : vec_bytes;
    if (lds + 2560 > 160 * 1024) { set_error("bdrt_qp_box_batch: n = %d too large", n); return -2; }      // (+ chol_packed's static LDS)
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
    const bool in_lds = (vec + msize + extra) * sizeof(double) <= 160 * 1024 - 2560;
    const size_t lds = (vec + (in_lds ? msize : 0) + extra) * sizeof(double);
    if (lds + 2560 > 160 * 1024) { set_error("bdrt_ridge: n = %d too large", n); return -2; }      // (+ chol_packed's static LDS)
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
    { const int on = getenv("BDRT_QP_PROF") ? 1 : 0; static int was = 0; if (on != was) { RG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof_on), &on, sizeof(on))); was = on; } }
    if (in_lds) hipLaunchKernelGGL(ridge_kernel<true>, dim3(nb), dim3(QP_NT), lds, 0, a);
    else hipLaunchKernelGGL(ridge_kernel<false>, dim3(nb), dim3(QP_NT), lds, 0, a);
    RG_HIP(hipGetLastError());
    RG_HIP(hipDeviceSynchronize());
    if (getenv("BDRT_QP_PROF")) {
        long long h[8] = {0}, z[8] = {0};
        RG_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qp_prof), sizeof(h)));
        RG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof), z, sizeof(z)));
        fprintf(stderr, "[bdrt qp prof] n %d, fit 0: %lld interior-point iterations; core cycles: form KKT %lld, factor %lld (diagonal blocks %lld, panels %lld, trailing %lld), triangular solves %lld, other %lld\n",
                n, h[4], h[0], h[1], h[5], h[6], h[7], h[2], h[3]);
    }
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
