// bdrt_big.h -- the evaluator for problems BEYOND the LDS budget of the tile / one-chain evaluators (more than 128 frequencies,
// more than ~200 basis functions per distribution, three wide distributions ...): the reference builds whatever grids it is
// given (bayes_drt/inversion.py:2127-2209) and Stan has no size limit, so neither may the drop-in have one.
//
// Slow but working: one workgroup of 512 threads per point, every intermediate vector in a per-point HBM workspace (L2-resident),
// the matrices as plain row-major copies (and their transposes) in HBM, dense products as one wavefront per output row
// (coalesced reads along the row, DPP reduction).  All model families of include/bdrt.h (series / parallel blocks, both
// outlier models, the x_sum prior).  Same model code as every other evaluator (bayes_drt/stan_model_files/*_modelcode.txt;
// gradient: SURVEY 8(a)), the arithmetic order of none of them: parity with the oracle to the usual 1e-10.
#pragma once
#include "bdrt_device.h"

namespace bdrt {

constexpr int BIG_NT = 512, BIG_NW = BIG_NT / 64;

__host__ __device__ inline int big_kmax(const DevProblem &P)
{
    int k = 0;
    for (int b = 0; b < P.nblocks; ++b) k = P.blk[b].K > k ? P.blk[b].K : k;
    return k;
}
// doubles of workspace per point: p, gp [D]; Zhat, gZ, gY [2 nf]; Y per block [2 nf]; v [3 Kmax], w [3 Kmax], xs [Kmax]
__host__ __device__ inline size_t big_ws_doubles(const DevProblem &P)
{
    const size_t N = 2 * (size_t)P.nf, K = (size_t)big_kmax(P);
    return ((2 * (size_t)P.D + 3 * N + N * MAXB + 7 * K + 16) + 7) & ~(size_t)7;
}

// sum of x over the 64 lanes, in every lane
__device__ __forceinline__ double big_wave_sum(double x)
{
    x = sum32(x);
    const double lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 0), __builtin_amdgcn_readlane(__double2loint(x), 0));
    const double hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 32), __builtin_amdgcn_readlane(__double2loint(x), 32));
    return lo + hi;
}

// sums over the workgroup of N per-thread values (fixed order); every thread gets the totals.  red: (BIG_NW + 1) * N doubles of LDS
template <int N>
__device__ inline void big_block_sum(double (&v)[N], double *red, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t = big_wave_sum(v[i]);
        if (lane == 0) red[wave * N + i] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        for (int w = 0; w < BIG_NW; ++w) s += red[w * N + i];
        v[i] = s;
    }
    __syncthreads();
}

// y[r] (+)= alpha * sum_c M[r][c] x[c], r < rows: one wavefront per row
template <bool ACC>
__device__ inline void big_matvec(const double *__restrict__ M, int rows, int cols, const double *x, double alpha, double *y, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < rows; r += BIG_NW) {
        const double *row = M + (size_t)r * cols;
        double s = 0.0;
        for (int c = lane; c < cols; c += 64) s = fma(row[c], x[c], s);
        s = big_wave_sum(s);
        if (lane == 0) y[r] = ACC ? y[r] + alpha * s : alpha * s;
    }
}

// log-posterior + gradient of one point.  theta / grad: [D] in global memory; ws: big_ws_doubles(P) doubles of global memory;
// red: 9 * 8 doubles of LDS.  Optional outputs (may be null): params [D] (constrained), Zhat, sig [2 nf].  All BIG_NT threads call.
__device__ inline void big_eval(const DevProblem &P, double *ws, const double *theta, double *grad, double *lp_out, int spec, int jacobian,
                                double *red, int tid, double *o_params = nullptr, double *o_Zhat = nullptr, double *o_sig = nullptr)
{
    const int nf = P.nf, N = 2 * nf, nb = P.nblocks, D = P.D;
    const int Kmax = big_kmax(P);
    double *p = ws, *gp = p + D, *Zhat = gp + D, *gZ = Zhat + N, *gY = gZ + N, *Yb = gY + N;
    double *v = Yb + (size_t)N * MAXB, *w = v + 3 * (size_t)Kmax, *xs = w + 3 * (size_t)Kmax;
    const double *Zm = P.Z + (size_t)spec * N;
    const double jac = jacobian ? 1.0 : 0.0;
    const int o_err = P.o_err, o_so = P.o_so;
    double lp = 0.0;                                       // this thread's share

    // ---- constrained parameters; everything is <lower=0> except the x of a sign-free series block ------------------------
    for (int j = tid; j < D; j += BIG_NT) {
        bool pos = true;
        for (int b = 0; b < nb; ++b)
            if (j >= P.blk[b].o_x && j < P.blk[b].o_x + P.blk[b].K && !P.blk[b].is_pos) pos = false;
        const double t = theta[j];
        p[j] = pos ? exp(t) : t;
        gp[j] = 0.0;
        if (pos) lp += jac * t;
    }
    __syncthreads();
    if (o_params) for (int j = tid; j < D; j += BIG_NT) o_params[j] = p[j];
    const double Rinf = 100.0 * p[0], induc = p[1] * P.induc_scale;
    const double s_res = 0.05 * p[o_err], a_p = 0.05 * p[o_err + 1], a_r = 0.05 * p[o_err + 2], a_i = 0.05 * p[o_err + 3];

    // ---- Z_hat ---------------------------------------------------------------------------------------------------------
    for (int b = 0; b < nb; ++b) {
        const DevBlock &B = P.blk[b];
        const double *xr = p + B.o_x;
        if (B.is_parallel) {
            for (int k = tid; k < B.K; k += BIG_NT) xs[k] = xr[k] * B.x_scale;
            __syncthreads();
            xr = xs;
        }
        big_matvec<false>(B.Ad, N, B.K, xr, 1.0, Yb + (size_t)b * N, tid);
        __syncthreads();
    }
    for (int n = tid; n < nf; n += BIG_NT) {
        double zr = Rinf, zi = induc * P.w[n];
        for (int b = 0; b < nb; ++b) {
            const double yr = Yb[(size_t)b * N + n], yi = Yb[(size_t)b * N + nf + n];
            if (P.blk[b].is_parallel) { const double Dn = yr * yr + yi * yi; zr += yr / Dn; zi += -yi / Dn; }
            else { zr += yr; zi += yi; }
        }
        Zhat[n] = zr; Zhat[nf + n] = zi;
    }
    // ---- x_sum prior (Series-Parallel_modelcode.txt:56-57, 89) ------------------------------------------------------------------
    double x_sum_raw = 0.0;
    if (P.use_x_sum) {
        double s[1] = {0.0};
        for (int b = 0; b < nb; ++b)
            for (int k = tid; k < P.blk[b].K; k += BIG_NT) s[0] += p[P.blk[b].o_x + k];
        big_block_sum<1>(s, red, tid);
        x_sum_raw = s[0];
        const double x_sum = x_sum_raw * P.x_sum_invscale;
        if (tid == 0) lp += -0.5 * x_sum * x_sum;
        for (int b = 0; b < nb; ++b)
            for (int k = tid; k < P.blk[b].K; k += BIG_NT) gp[P.blk[b].o_x + k] += -x_sum * P.x_sum_invscale;
    }
    __syncthreads();

    // ---- likelihood: Z ~ normal(Z_hat, sigma_tot) ----------------------------------------------------------------------------------
    {
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        double S[6] = {0, 0, 0, 0, 0, 0};                  // sR, sL, S_h, S_hz2, S_hzr2, S_hzi2
        for (int n = tid; n < nf; n += BIG_NT) {
            const double zr = Zhat[n], zi = Zhat[nf + n];
            double so_re = 0.0, so_im = 0.0;
            if (P.outlier_mode == 1) so_re = so_im = 0.05 * p[o_so + n] * p[o_so + nf + n];
            else if (P.outlier_mode == 2) { so_re = 0.05 * p[o_so + n]; so_im = 0.05 * p[o_so + nf + n]; }
            const double common = (a_r * zr) * (a_r * zr) + (a_i * zi) * (a_i * zi);
            const double s2_re = c0 + (a_p * zr) * (a_p * zr) + common + so_re * so_re;
            const double s2_im = c0 + (a_p * zi) * (a_p * zi) + common + so_im * so_im;
            const double e_re = Zm[n] - zr, e_im = Zm[nf + n] - zi;
            lp += -0.5 * log(s2_re) - 0.5 * e_re * e_re / s2_re - 0.5 * log(s2_im) - 0.5 * e_im * e_im / s2_im;
            if (o_sig) { o_sig[n] = sqrt(s2_re); o_sig[nf + n] = sqrt(s2_im); }
            const double w_re = 1.0 / s2_re, w_im = 1.0 / s2_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re, h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (a_p * a_p + a_r * a_r) + h_im * a_r * a_r);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (a_p * a_p + a_i * a_i) + h_re * a_i * a_i);
            gZ[n] = gzr; gZ[nf + n] = gzi;
            S[0] += gzr; S[1] += gzi * P.w[n]; S[2] += h_re + h_im; S[3] += h_re * zr * zr + h_im * zi * zi;
            S[4] += (h_re + h_im) * zr * zr; S[5] += (h_re + h_im) * zi * zi;
            if (P.outlier_mode == 1) {
                const double r = p[o_so + n], s = p[o_so + nf + n];
                const double dso = 2.0 * so_re * (h_re + h_im);
                gp[o_so + n] += 0.05 * s * dso - P.so_lambda;
                gp[o_so + nf + n] += 0.05 * r * dso - (P.so_alpha + 1.0) / s + P.so_beta / (s * s);
                lp += -P.so_lambda * r - (P.so_alpha + 1.0) * log(s) - P.so_beta / s;
            } else if (P.outlier_mode == 2) {
                gp[o_so + n] += 0.05 * 2.0 * so_re * h_re - P.so_lambda;
                gp[o_so + nf + n] += 0.05 * 2.0 * so_im * h_im - P.so_lambda;
                lp += -P.so_lambda * (p[o_so + n] + p[o_so + nf + n]);
            }
        }
        if (o_Zhat) for (int i = tid; i < N; i += BIG_NT) o_Zhat[i] = Zhat[i];
        big_block_sum<6>(S, red, tid);
        if (tid == 0) {
            gp[0] += 100.0 * S[0] - p[0];
            gp[1] += P.induc_scale * S[1] - p[1];
            lp += -0.5 * p[0] * p[0] - 0.5 * p[1] * p[1];
            gp[o_err] += 0.05 * 2.0 * s_res * S[2] - p[o_err];
            gp[o_err + 1] += 0.05 * 2.0 * a_p * S[3] - p[o_err + 1];
            gp[o_err + 2] += 0.05 * 2.0 * a_r * S[4] - p[o_err + 2];
            gp[o_err + 3] += 0.05 * 2.0 * a_i * S[5] - p[o_err + 3];
            for (int j = 0; j < 4; ++j) lp += -0.5 * p[o_err + j] * p[o_err + j];
        }
    }
    __syncthreads();

    // ---- per block: A^T back-propagation and the complexity prior ----------------------------------------------------------------------
    for (int b = 0; b < nb; ++b) {
        const DevBlock &B = P.blk[b];
        const int K = B.K;
        const double *xr = p + B.o_x, *ur = p + B.o_ups, *dd = p + B.o_d;
        double *gx = gp + B.o_x;
        if (!B.is_parallel) {
            big_matvec<true>(B.At, K, N, gZ, 1.0, gx, tid);
        } else {
            const double *Y = Yb + (size_t)b * N;
            for (int n = tid; n < nf; n += BIG_NT) {
                const double yr = Y[n], yi = Y[nf + n], Dn = yr * yr + yi * yi, D2 = Dn * Dn;
                const double d_diag = (yi * yi - yr * yr) / D2, d_off = 2.0 * yr * yi / D2;
                gY[n] = gZ[n] * d_diag + gZ[nf + n] * d_off;
                gY[nf + n] = -gZ[n] * d_off + gZ[nf + n] * d_diag;
            }
            __syncthreads();
            big_matvec<true>(B.At, K, N, gY, B.x_scale, gx, tid);
        }
        // v_i = L_i x (the stack [L0; L1; L2] is one 3K x K matrix)
        big_matvec<false>(B.Ld, 3 * K, K, xr, 1.0, v, tid);
        __syncthreads();
        double Sv[3] = {0, 0, 0};
        for (int k = tid; k < K; k += BIG_NT) {
            const double u = 0.15 * ur[k];
            double q2 = 0.0;
            for (int i = 0; i < 3; ++i) { const double vi = v[(size_t)i * K + k]; q2 += dd[i] * vi * vi; Sv[i] += vi * vi / (u * u); w[(size_t)i * K + k] = vi / (u * u); }
            lp += -log(u) - 0.5 * q2 / (u * u) - (P.ups_alpha + 1.0) * log(ur[k]) - P.ups_beta / ur[k];
            double gu = 0.15 * (-1.0 / u + q2 / (u * u * u)) - (P.ups_alpha + 1.0) / ur[k] + P.ups_beta / (ur[k] * ur[k]);
            // dups[c] = 0.5 (ups[c] - 0.5 (ups[c-1] + ups[c+1])) / ups[c], c = 1 .. K-2; dups ~ std_normal(): this element as the
            // centre, as the right neighbour of centre k-1, as the left neighbour of centre k+1
            if (k >= 1 && k + 1 < K) {
                const double um = 0.15 * ur[k - 1], up = 0.15 * ur[k + 1];
                const double du = 0.5 * (u - 0.5 * (um + up)) / u;
                lp += -0.5 * du * du;
                gu += 0.15 * (-du * 0.25 * (um + up) / (u * u));
            }
            if (k >= 2) {
                const double um = 0.15 * ur[k - 2], u0 = 0.15 * ur[k - 1];
                const double du = 0.5 * (u0 - 0.5 * (um + u)) / u0;
                gu += 0.15 * (du * 0.25 / u0);
            }
            if (k + 2 < K) {
                const double u0 = 0.15 * ur[k + 1], up = 0.15 * ur[k + 2];
                const double du = 0.5 * (u0 - 0.5 * (u + up)) / u0;
                gu += 0.15 * (du * 0.25 / u0);
            }
            gp[B.o_ups + k] += gu;
        }
        big_block_sum<3>(Sv, red, tid);
        if (tid == 0)
            for (int i = 0; i < 3; ++i) {                  // d ~ inv_gamma(5, 5)
                gp[B.o_d + i] += -0.5 * Sv[i] - 6.0 / dd[i] + 5.0 / (dd[i] * dd[i]);
                lp += -6.0 * log(dd[i]) - 5.0 / dd[i];
            }
        // g_x -= sum_i d_i L_i^T w_i: the transposed stack [L0^T | L1^T | L2^T] is one K x 3K matrix; fold d_i into w
        for (int i = 0; i < 3; ++i)
            for (int k = tid; k < K; k += BIG_NT) w[(size_t)i * K + k] *= dd[i];
        __syncthreads();
        big_matvec<true>(B.Lt, K, 3 * K, w, -1.0, gx, tid);
        __syncthreads();
    }

    // ---- unconstraining transform; the log-posterior -----------------------------------------------------------------------------------
    {
        double s[1] = {lp};
        big_block_sum<1>(s, red, tid);
        lp = s[0];
    }
    const bool rejected = P.use_x_sum && x_sum_raw < 0.0;  // real<lower=0> x_sum_raw: Stan rejects the point
    if (grad)
        for (int j = tid; j < D; j += BIG_NT) {
            bool pos = true;
            for (int b = 0; b < nb; ++b)
                if (j >= P.blk[b].o_x && j < P.blk[b].o_x + P.blk[b].K && !P.blk[b].is_pos) pos = false;
            grad[j] = rejected ? 0.0 : (pos ? p[j] * gp[j] + jac : gp[j]);
        }
    if (tid == 0 && lp_out) *lp_out = rejected ? -INFINITY : lp;
    __syncthreads();
}

}  // namespace bdrt
