/*
 * bdrt_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See bdrt_oracle.h.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Straight scalar C, fp64, no tricks: this is the checker.
 */
#define _GNU_SOURCE
#include "bdrt_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------
 * Parameter layout = Stan declaration order (bayes_drt/stan_model_files/<name>_modelcode.txt `parameters`):
 *   Rinf_raw, induc_raw, x-blocks (xs, xp1_raw, xp2_raw), sigma_res_raw, alpha_prop_raw, alpha_re_raw,
 *   alpha_im_raw, [sigma_out_raw (, sigma_out_scale)], ups-blocks, (d0,d1,d2)-blocks
 * e.g. Series_pos_modelcode.txt:24-36, Series_pos_outliers_modelcode.txt:22-37,
 *      Series-Parallel_pos_outliers_modelcode.txt:32-50, Series-2Parallel_pos_modelcode.txt:39-61.
 * ---------------------------------------------------------------------------------------------- */
int orc_layout(const orc_model *m, int *o_x, int *o_err, int *o_so, int *o_ups, int *o_d, unsigned char *is_pos)
{
    int o = 0;
    if (is_pos) { is_pos[0] = 1; is_pos[1] = 1; }
    o = 2;
    for (int b = 0; b < m->nblocks; ++b) {
        if (o_x) o_x[b] = o;
        if (is_pos)
            for (int k = 0; k < m->K[b]; ++k) is_pos[o + k] = (unsigned char)(m->is_parallel[b] || m->nonneg[b]);
        o += m->K[b];
    }
    if (o_err) *o_err = o;
    if (is_pos) for (int j = 0; j < 4; ++j) is_pos[o + j] = 1;
    o += 4;
    if (o_so) *o_so = o;
    if (m->outlier_mode) {
        if (is_pos) for (int j = 0; j < 2 * m->nf; ++j) is_pos[o + j] = 1;
        o += 2 * m->nf;
    }
    for (int b = 0; b < m->nblocks; ++b) {
        if (o_ups) o_ups[b] = o;
        if (is_pos) for (int k = 0; k < m->K[b]; ++k) is_pos[o + k] = 1;
        o += m->K[b];
    }
    for (int b = 0; b < m->nblocks; ++b) {
        if (o_d) o_d[b] = o;
        if (is_pos) for (int j = 0; j < 3; ++j) is_pos[o + j] = 1;
        o += 3;
    }
    return o;
}

int orc_num_params(const orc_model *m) { return orc_layout(m, 0, 0, 0, 0, 0, 0); }

void orc_constrain(const orc_model *m, const double *theta, double *params)
{
    int D = orc_num_params(m);
    unsigned char *pos = (unsigned char *)malloc((size_t)D);
    orc_layout(m, 0, 0, 0, 0, 0, pos);
    for (int j = 0; j < D; ++j) params[j] = pos[j] ? exp(theta[j]) : theta[j];
    free(pos);
}

void orc_unconstrain(const orc_model *m, const double *params, double *theta)
{
    int D = orc_num_params(m);
    unsigned char *pos = (unsigned char *)malloc((size_t)D);
    orc_layout(m, 0, 0, 0, 0, 0, pos);
    for (int j = 0; j < D; ++j) theta[j] = pos[j] ? log(params[j]) : params[j];
    free(pos);
}

/* y = M x, M [r x c] row-major */
static void matvec(const double *M, int r, int c, const double *x, double *y)
{
    for (int i = 0; i < r; ++i) {
        double s = 0.0;
        const double *row = M + (size_t)i * c;
        for (int j = 0; j < c; ++j) s += row[j] * x[j];
        y[i] = s;
    }
}

/* y += alpha * M^T v */
static void matTvec_acc(const double *M, int r, int c, const double *v, double alpha, double *y)
{
    for (int i = 0; i < r; ++i) {
        const double *row = M + (size_t)i * c;
        double a = alpha * v[i];
        for (int j = 0; j < c; ++j) y[j] += row[j] * a;
    }
}

/*
 * Core evaluation.  Follows (for the single-series family) Series_modelcode.txt:37-69 and for the
 * series/parallel families Series-Parallel_pos_outliers_modelcode.txt:51-108,
 * Series-2Parallel_pos_modelcode.txt:62-127, Parallel_modelcode.txt:35-73.
 * The reverse pass is hand-derived (SURVEY.md 8(a) "Gradient of S1").
 */
static int eval_core(const orc_model *m, const double *theta, int jacobian, int want_grad, double *lp_out, double *grad,
                     double *o_Zhat, double *o_sig, double *o_q, double *o_ups, double *o_dups, double *o_xsum)
{
    const int nf = m->nf, N = 2 * nf, nb = m->nblocks;
    int o_x[ORC_MAX_BLOCKS], o_u[ORC_MAX_BLOCKS], o_d[ORC_MAX_BLOCKS], o_err, o_so;
    const int D = orc_num_params(m);
    unsigned char *pos = (unsigned char *)malloc((size_t)D);
    orc_layout(m, o_x, &o_err, &o_so, o_u, o_d, pos);

    int Kmax = 0, Ktot = 0;
    for (int b = 0; b < nb; ++b) { if (m->K[b] > Kmax) Kmax = m->K[b]; Ktot += m->K[b]; }

    double *p = (double *)malloc(sizeof(double) * (size_t)D);
    double *gp = (double *)calloc((size_t)D, sizeof(double));
    for (int j = 0; j < D; ++j) p[j] = pos[j] ? exp(theta[j]) : theta[j];

    double *Zhat = (double *)calloc((size_t)N, sizeof(double));
    double *Yb = (double *)calloc((size_t)N * ORC_MAX_BLOCKS, sizeof(double));
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(N > Kmax ? N : Kmax));
    double *xs = (double *)malloc(sizeof(double) * (size_t)Kmax);
    double *gZ = (double *)calloc((size_t)N, sizeof(double));
    double *gY = (double *)calloc((size_t)N, sizeof(double));
    double *v = (double *)malloc(sizeof(double) * 3 * (size_t)Kmax);
    double *w = (double *)malloc(sizeof(double) * (size_t)Kmax);

    const double Rinf = 100.0 * p[0];             /* real Rinf = Rinf_raw*100            */
    const double induc = p[1] * m->induc_scale;   /* real induc = induc_raw*induc_scale  */
    const double s_res = 0.05 * p[o_err + 0];
    const double a_p = 0.05 * p[o_err + 1];
    const double a_r = 0.05 * p[o_err + 2];
    const double a_i = 0.05 * p[o_err + 3];

    double lp = 0.0;
    int rejected = 0;

    /* ---- Z_hat ---------------------------------------------------------------------------- */
    for (int b = 0; b < nb; ++b) {
        const int K = m->K[b];
        const double *xr = p + o_x[b];
        if (!m->is_parallel[b]) {
            matvec(m->A[b], N, K, xr, tmp);       /* As*xs */
            for (int i = 0; i < N; ++i) Zhat[i] += tmp[i];
        } else {
            for (int k = 0; k < K; ++k) xs[k] = xr[k] * m->x_scale[b];   /* xp = xp_raw*xp_scale */
            double *Y = Yb + (size_t)b * N;
            matvec(m->A[b], N, K, xs, Y);         /* Y_hat = Ap*xp */
            for (int n = 0; n < nf; ++n) {
                double yr = Y[n], yi = Y[nf + n], Dn = yr * yr + yi * yi;
                Zhat[n] += yr / Dn;               /* Y_hat_re ./ (square(Y_hat_re)+square(Y_hat_im)) */
                Zhat[nf + n] += -yi / Dn;
            }
        }
    }
    for (int n = 0; n < nf; ++n) {
        Zhat[n] += Rinf;                                   /* Rinf*Rinf_vec   */
        Zhat[nf + n] += induc * (2.0 * M_PI * m->freq[n]); /* induc*induc_vec */
    }

    /* ---- x_sum (Series-Parallel_modelcode.txt:56-57,89) ------------------------------------ */
    double x_sum_raw = 0.0, x_sum = 0.0;
    if (m->use_x_sum) {
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < m->K[b]; ++k) x_sum_raw += p[o_x[b] + k];
        x_sum = x_sum_raw * m->x_sum_invscale;
        if (x_sum_raw < 0.0) rejected = 1;     /* real<lower=0> x_sum_raw : Stan rejects the proposal */
        lp += -0.5 * x_sum * x_sum;            /* x_sum ~ std_normal() */
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < m->K[b]; ++k) gp[o_x[b] + k] += -x_sum * m->x_sum_invscale;
    }
    if (o_xsum) *o_xsum = x_sum;

    /* ---- likelihood: Z ~ normal(Z_hat, sigma_tot) ------------------------------------------- */
    const double c0 = m->sigma_min * m->sigma_min + s_res * s_res;
    double S_h = 0, S_hz2 = 0, S_hzr2 = 0, S_hzi2 = 0;
    for (int n = 0; n < nf; ++n) {
        const double zr = Zhat[n], zi = Zhat[nf + n];
        double so_re = 0.0, so_im = 0.0;
        if (m->outlier_mode == 1) {            /* sigma_out = raw .* scale * 0.05, same for re and im rows */
            so_re = so_im = 0.05 * p[o_so + n] * p[o_so + nf + n];
        } else if (m->outlier_mode == 2) {     /* sigma_out = raw*0.05, one per stacked row */
            so_re = 0.05 * p[o_so + n];
            so_im = 0.05 * p[o_so + nf + n];
        }
        const double common = (a_r * zr) * (a_r * zr) + (a_i * zi) * (a_i * zi);
        const double s2_re = c0 + (a_p * zr) * (a_p * zr) + common + so_re * so_re;
        const double s2_im = c0 + (a_p * zi) * (a_p * zi) + common + so_im * so_im;
        const double e_re = m->Z[n] - zr, e_im = m->Z[nf + n] - zi;
        lp += -0.5 * log(s2_re) - 0.5 * e_re * e_re / s2_re;
        lp += -0.5 * log(s2_im) - 0.5 * e_im * e_im / s2_im;
        if (o_sig) { o_sig[n] = sqrt(s2_re); o_sig[nf + n] = sqrt(s2_im); }
        const double w_re = 1.0 / s2_re, w_im = 1.0 / s2_im;
        const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;   /* d lp / d sigma^2 */
        const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
        gZ[n] = e_re * w_re + 2.0 * zr * (h_re * (a_p * a_p + a_r * a_r) + h_im * a_r * a_r);
        gZ[nf + n] = e_im * w_im + 2.0 * zi * (h_im * (a_p * a_p + a_i * a_i) + h_re * a_i * a_i);
        S_h += h_re + h_im;
        S_hz2 += h_re * zr * zr + h_im * zi * zi;
        S_hzr2 += (h_re + h_im) * zr * zr;
        S_hzi2 += (h_re + h_im) * zi * zi;
        if (m->outlier_mode == 1) {
            const double r = p[o_so + n], s = p[o_so + nf + n];
            const double dso = 2.0 * so_re * (h_re + h_im);
            gp[o_so + n] += 0.05 * s * dso - m->so_lambda;                       /* raw ~ exponential(lambda) */
            gp[o_so + nf + n] += 0.05 * r * dso - (m->so_alpha + 1.0) / s + m->so_beta / (s * s); /* scale ~ inv_gamma */
            lp += -m->so_lambda * r - (m->so_alpha + 1.0) * log(s) - m->so_beta / s;
        } else if (m->outlier_mode == 2) {
            gp[o_so + n] += 0.05 * 2.0 * so_re * h_re - m->so_lambda;
            gp[o_so + nf + n] += 0.05 * 2.0 * so_im * h_im - m->so_lambda;
            lp += -m->so_lambda * (p[o_so + n] + p[o_so + nf + n]);
        }
    }
    if (o_Zhat) memcpy(o_Zhat, Zhat, sizeof(double) * (size_t)N);

    /* offsets and error-structure scalars (all ~ std_normal on the raw scale) */
    {
        double sR = 0, sL = 0;
        for (int n = 0; n < nf; ++n) { sR += gZ[n]; sL += gZ[nf + n] * (2.0 * M_PI * m->freq[n]); }
        gp[0] += 100.0 * sR - p[0];
        gp[1] += m->induc_scale * sL - p[1];
        lp += -0.5 * p[0] * p[0] - 0.5 * p[1] * p[1];
        gp[o_err + 0] += 0.05 * 2.0 * s_res * S_h - p[o_err + 0];
        gp[o_err + 1] += 0.05 * 2.0 * a_p * S_hz2 - p[o_err + 1];
        gp[o_err + 2] += 0.05 * 2.0 * a_r * S_hzr2 - p[o_err + 2];
        gp[o_err + 3] += 0.05 * 2.0 * a_i * S_hzi2 - p[o_err + 3];
        for (int j = 0; j < 4; ++j) lp += -0.5 * p[o_err + j] * p[o_err + j];
    }

    /* ---- per-block: A^T back-propagation and complexity prior -------------------------------- */
    int qoff = 0, doff = 0;
    for (int b = 0; b < nb; ++b) {
        const int K = m->K[b];
        const double *xr = p + o_x[b];
        double *gx = gp + o_x[b];
        if (!m->is_parallel[b]) {
            if (want_grad) matTvec_acc(m->A[b], N, K, gZ, 1.0, gx);
        } else if (want_grad) {
            const double *Y = Yb + (size_t)b * N;
            for (int n = 0; n < nf; ++n) {
                double yr = Y[n], yi = Y[nf + n], Dn = yr * yr + yi * yi, D2 = Dn * Dn;
                double d_diag = (yi * yi - yr * yr) / D2, d_off = 2.0 * yr * yi / D2;
                gY[n] = gZ[n] * d_diag + gZ[nf + n] * d_off;
                gY[nf + n] = -gZ[n] * d_off + gZ[nf + n] * d_diag;
            }
            matTvec_acc(m->A[b], N, K, gY, m->x_scale[b], gx);
        }
        /* q = sqrt(d0*square(L0*x)+d1*square(L1*x)+d2*square(L2*x)); q ~ normal(0, ups).  Only q^2
         * enters the density: evaluated on q^2 (finite gradient at q=0). */
        const double *Ls[3] = {m->L0[b], m->L1[b], m->L2[b]};
        const double *dd = p + o_d[b];
        for (int i = 0; i < 3; ++i) matvec(Ls[i], K, K, xr, v + (size_t)i * K);
        const double *ur = p + o_u[b];
        double Sv[3] = {0, 0, 0};
        for (int k = 0; k < K; ++k) {
            const double u = 0.15 * ur[k];
            double q2 = 0.0;
            for (int i = 0; i < 3; ++i) q2 += dd[i] * v[(size_t)i * K + k] * v[(size_t)i * K + k];
            lp += -log(u) - 0.5 * q2 / (u * u);
            for (int i = 0; i < 3; ++i) Sv[i] += v[(size_t)i * K + k] * v[(size_t)i * K + k] / (u * u);
            if (o_q) o_q[qoff + k] = sqrt(q2);
            if (o_ups) o_ups[qoff + k] = u;
            /* d/du of (-log u - q2/(2u^2)) ; chain 0.15 ; ups_raw ~ inv_gamma(alpha,beta) */
            double gu = -1.0 / u + q2 / (u * u * u);
            gp[o_u[b] + k] += 0.15 * gu - (m->ups_alpha + 1.0) / ur[k] + m->ups_beta / (ur[k] * ur[k]);
            lp += -(m->ups_alpha + 1.0) * log(ur[k]) - m->ups_beta / ur[k];
        }
        if (want_grad) {
            for (int i = 0; i < 3; ++i) {
                for (int k = 0; k < K; ++k) { double u = 0.15 * ur[k]; w[k] = v[(size_t)i * K + k] / (u * u); }
                matTvec_acc(Ls[i], K, K, w, -dd[i], gx);
            }
        }
        for (int i = 0; i < 3; ++i) {           /* d ~ inv_gamma(5,5) */
            gp[o_d[b] + i] += -0.5 * Sv[i] - 6.0 / dd[i] + 5.0 / (dd[i] * dd[i]);
            lp += -6.0 * log(dd[i]) - 5.0 / dd[i];
        }
        /* dups[k] = 0.5*(ups[k+1] - 0.5*(ups[k]+ups[k+2]))/ups[k+1]; dups ~ std_normal() */
        for (int c = 1; c + 1 < K; ++c) {
            const double um = 0.15 * ur[c - 1], u0 = 0.15 * ur[c], up = 0.15 * ur[c + 1];
            const double du = 0.5 * (u0 - 0.5 * (um + up)) / u0;
            lp += -0.5 * du * du;
            if (o_dups) o_dups[doff + c - 1] = du;
            /* d du/d u0 = 0.25 (um+up)/u0^2 ; d du/d um = d du/d up = -0.25/u0 */
            gp[o_u[b] + c] += 0.15 * (-du * 0.25 * (um + up) / (u0 * u0));
            gp[o_u[b] + c - 1] += 0.15 * (du * 0.25 / u0);
            gp[o_u[b] + c + 1] += 0.15 * (du * 0.25 / u0);
        }
        qoff += K;
        doff += K - 2;
    }

    /* ---- unconstraining transform: lower=0 -> exp; Jacobian only when sampling -------------- */
    if (grad) {
        for (int j = 0; j < D; ++j) {
            if (pos[j]) grad[j] = p[j] * gp[j] + (jacobian ? 1.0 : 0.0);
            else grad[j] = gp[j];
        }
    }
    if (jacobian) for (int j = 0; j < D; ++j) if (pos[j]) lp += theta[j];

    if (rejected) {
        lp = -INFINITY;
        if (grad) memset(grad, 0, sizeof(double) * (size_t)D);
    }
    if (lp_out) *lp_out = lp;

    free(pos); free(p); free(gp); free(Zhat); free(Yb); free(tmp); free(xs); free(gZ); free(gY); free(v); free(w);
    return rejected;
}

int orc_logp_grad(const orc_model *m, const double *theta, int jacobian, double *lp, double *grad)
{
    return eval_core(m, theta, jacobian, grad != 0, lp, grad, 0, 0, 0, 0, 0, 0);
}

int orc_forward(const orc_model *m, const double *theta, double *Z_hat, double *sigma_tot, double *q_all,
                double *ups_all, double *dups_all, double *x_sum)
{
    double lp;
    return eval_core(m, theta, 0, 0, &lp, 0, Z_hat, sigma_tot, q_all, ups_all, dups_all, x_sum);
}

/* ================================================================================================
 * Matrix construction (bayes_drt/matrices.py)
 * ============================================================================================== */
#define ORC_NQUAD 1000

/* get_basis_func (matrices.py:8-24): 0 gaussian exp(-(eps y)^2) (:12-13); 1 Cole-Cole (:15-17); 2 Zic, eps unused (:19-21) */
static double basis_phi(double y, double eps, int basis)
{
    if (basis == ORC_BASIS_COLE_COLE)
        return (1.0 / (2.0 * M_PI)) * sin((1.0 - eps) * M_PI) / (cosh(eps * y) - cos((1.0 - eps) * M_PI));
    if (basis == ORC_BASIS_ZIC) return 2.0 * exp(y) / (1.0 + exp(2.0 * y));
    return exp(-(eps * y) * (eps * y));
}

/* integrand of get_A_func (matrices.py:27-117) */
static double integrand(double y, double w_n, double t_m, double eps, int kernel, int part, int dist_series,
                        int use_ct, double k_ct, int basis)
{
    const double phi = basis_phi(y, eps, basis);
    if (kernel == ORC_KERNEL_DRT) {
        const double den = 1.0 + exp(2.0 * (y + log(w_n * t_m)));
        if (part == 0) return phi / den;                       /* :48-49 */
        return -phi * exp(y) * w_n * t_m / den;                /* :51-52 */
    }
    double complex arg = use_ct ? t_m * exp(y) * (k_ct + I * w_n) : (I * w_n * t_m * exp(y));
    double complex x = csqrt(arg);
    double complex th = ctanh(x);
    double complex ZD;
    if (kernel == ORC_KERNEL_DDT_BLOCK_PLANAR) ZD = 1.0 / (th * x);          /* :62-70 */
    else if (kernel == ORC_KERNEL_DDT_BLOCK_SPHER) ZD = th / (x - th);        /* :74-80 */
    else ZD = th / x;                                                         /* :86-92 */
    double complex val = dist_series ? ZD : 1.0 / ZD;                          /* :97-110 */
    return phi * (part == 0 ? creal(val) : cimag(val));
}

/* np.trapz(func(y), x=y) with y = np.linspace(-20, 20, 1000)  (matrices.py:236-238, :262-263) */
static double trapz_entry(double w_n, double t_m, double eps, int kernel, int part, int dist_series, int use_ct,
                          double k_ct, int basis)
{
    const double step = 40.0 / (ORC_NQUAD - 1);
    double s = 0.0, yprev = -20.0;
    double fprev = integrand(yprev, w_n, t_m, eps, kernel, part, dist_series, use_ct, k_ct, basis);
    for (int i = 1; i < ORC_NQUAD; ++i) {
        double y = (i == ORC_NQUAD - 1) ? 20.0 : -20.0 + i * step;
        double f = integrand(y, w_n, t_m, eps, kernel, part, dist_series, use_ct, k_ct, basis);
        s += (y - yprev) * (f + fprev) / 2.0;
        yprev = y; fprev = f;
    }
    return s;
}

int orc_build_A(const double *freq, int nf, const double *tau, int k, double eps, int kernel, int part,
                int dist_series, int use_ct, double k_ct, int toeplitz, double *out)
{
    return orc_build_A_basis(freq, nf, tau, k, eps, kernel, part, dist_series, use_ct, k_ct, toeplitz, ORC_BASIS_GAUSSIAN, out);
}

int orc_build_A_basis(const double *freq, int nf, const double *tau, int k, double eps, int kernel, int part,
                      int dist_series, int use_ct, double k_ct, int toeplitz, int basis, double *out)
{
    if (toeplitz) {
        double *c = (double *)malloc(sizeof(double) * (size_t)nf);
        double *r = (double *)malloc(sizeof(double) * (size_t)k);
        const double w0 = freq[0] * 2.0 * M_PI, t0 = tau[0];
        for (int n = 0; n < nf; ++n) c[n] = trapz_entry(freq[n] * 2.0 * M_PI, t0, eps, kernel, part, dist_series, use_ct, k_ct, basis);
        for (int j = 0; j < k; ++j) r[j] = trapz_entry(w0, tau[j], eps, kernel, part, dist_series, use_ct, k_ct, basis);
        int bad = (r[0] != c[0]);                     /* matrices.py:239-241 */
        for (int n = 0; n < nf; ++n)
            for (int j = 0; j < k; ++j) out[(size_t)n * k + j] = (n >= j) ? c[n - j] : r[j - n];  /* toeplitz(c, r) */
        free(c); free(r);
        return bad ? -1 : 0;
    }
    for (int n = 0; n < nf; ++n)
        for (int j = 0; j < k; ++j)
            out[(size_t)n * k + j] = trapz_entry(freq[n] * 2.0 * M_PI, tau[j], eps, kernel, part, dist_series, use_ct, k_ct, basis);
    return 0;
}

static double l_entry(double w_n, double t_m, double eps, const double *coef4, int basis)
{
    const double y = log(1.0 / (w_n * t_m));                     /* matrices.py:323 */
    if (basis == ORC_BASIS_ZIC) return coef4[0] * basis_phi(y, eps, basis);       /* :316-318: order 0 only */
    const double g = exp(-(eps * y) * (eps * y));
    const double e2 = eps * eps;
    double val = 0.0;
    if (coef4[0] != 0.0) val += coef4[0] * g;                                              /* :288 */
    if (coef4[1] != 0.0) val += coef4[1] * (-2.0 * e2 * y * g);                            /* :292 */
    if (coef4[2] != 0.0) val += coef4[2] * ((-2.0 * e2 + 4.0 * e2 * e2 * y * y) * g);      /* :296 */
    if (coef4[3] != 0.0) val += coef4[3] * ((12.0 * e2 * e2 * y - 8.0 * e2 * e2 * e2 * y * y * y) * g); /* :300 */
    return val;
}

void orc_build_L(const double *tau, int k, double eps, const double *coef4, double *out)
{
    /* construct_L is called with frequencies = 1/(2 pi tau) (inversion.py:2302-2307) */
    for (int n = 0; n < k; ++n) {
        const double f_n = 1.0 / (2.0 * M_PI * tau[n]);
        const double w_n = 2.0 * M_PI * f_n;
        for (int j = 0; j < k; ++j) out[(size_t)n * k + j] = l_entry(w_n, tau[j], eps, coef4, ORC_BASIS_GAUSSIAN);
    }
}

/* construct_L as the function itself is written (matrices.py:268-325): any `frequencies` against any `tau`, [nf x k] */
void orc_build_L_rect(const double *freq, int nf, const double *tau, int k, double eps, const double *coef4, int basis,
                      double *out)
{
    for (int n = 0; n < nf; ++n) {
        const double w_n = 2.0 * M_PI * freq[n];
        for (int j = 0; j < k; ++j) out[(size_t)n * k + j] = l_entry(w_n, tau[j], eps, coef4, basis);
    }
}

static double m_entry(double w_n, double t_m, double eps, const double *coef3)
{
    const double a = eps * log(1.0 / (w_n * t_m));
    const double g = exp(-(a * a / 2.0));
    const double rt = sqrt(M_PI / 2.0);
    double val = 0.0;
    if (coef3[0] != 0.0) val += coef3[0] * (rt / eps * g);                                   /* matrices.py:344 */
    if (coef3[1] != 0.0) val += coef3[1] * (-rt * eps * (-1.0 + a * a) * g);                 /* :351 */
    if (coef3[2] != 0.0) val += coef3[2] * (rt * eps * eps * eps * (3.0 - 6.0 * a * a + a * a * a * a) * g); /* :358 */
    return val;
}

void orc_build_M(const double *tau, int k, double eps, const double *coef3, int toeplitz, double *out)
{
    /* construct_M(frequencies = 1/(2 pi tau)) (inversion.py:2297-2299); omega = frequencies*2*pi */
    double *omega = (double *)malloc(sizeof(double) * (size_t)k);
    for (int n = 0; n < k; ++n) omega[n] = (1.0 / (2.0 * M_PI * tau[n])) * 2.0 * M_PI;
    if (toeplitz) {                                  /* matrices.py:396-405: symmetric toeplitz(c) */
        const double t0 = 1.0 / omega[0];
        double *c = (double *)malloc(sizeof(double) * (size_t)k);
        for (int n = 0; n < k; ++n) c[n] = m_entry(omega[n], t0, eps, coef3);
        for (int n = 0; n < k; ++n)
            for (int j = 0; j < k; ++j) out[(size_t)n * k + j] = c[abs(n - j)];
        free(c);
    } else {
        for (int n = 0; n < k; ++n)
            for (int j = 0; j < k; ++j) out[(size_t)n * k + j] = m_entry(omega[n], 1.0 / omega[j], eps, coef3);
    }
    free(omega);
}
