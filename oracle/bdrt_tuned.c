/*
 * bdrt_tuned.c -- a TUNED CPU evaluator of the headline family (Series / Series_pos, one DRT block, no outlier parameters):
 * test infrastructure like the rest of oracle/ -- it exists for bench.py's `cpu_baseline.tuned` figure ("the
 * same dense algorithm, preallocated workspace, -O3 -march=native") and is checked against the oracle by tests/test_oracle_tuned.py.
 * The product never links or loads it.
 *
 * Same formulas as oracle/bdrt_oracle.c::eval_core for this family (Series_pos_modelcode.txt:37-69; gradient hand-derived, SURVEY
 * 8(a)), organised for a CPU core: one preallocated workspace per handle (the checker mallocs a dozen vectors per call), A and a
 * transposed copy of A walked along contiguous rows in both products (dense: 2 Nf x K each way, as SURVEY 8(d) counts them), the
 * three penalty operators as dense K x K products as well (what "dense formulation" means in 8(d)) or -- `banded` -- through their
 * 2 bw + 1 diagonals when they are banded Toeplitz (what the GPU's structured path executes).  Plain C loops that gcc vectorises
 * (-O3 -march=native -ffp-contract=fast); exp / log from libm.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int nf, K, pos, banded, bw;
    double *A, *At;            /* [2nf][K], [K][2nf] */
    double *L[3], *Lt[3];      /* dense [K][K] and transposes */
    double taps[3][64];        /* banded: T_i[d + bw] */
    double *Z, *w;             /* [2nf], 2 pi f [nf] */
    double sigma_min, ups_alpha, ups_beta, induc_scale;
    /* workspace */
    double *x, *u, *iu, *v[3], *wv[3], *zh, *gz, *tmp;
} tuned_s1;

static double *dup(const double *s, size_t n) { double *d = (double *)malloc(n * sizeof(double)); memcpy(d, s, n * sizeof(double)); return d; }

void tuned_s1_destroy(tuned_s1 *t)
{
    if (!t) return;
    free(t->A); free(t->At); free(t->Z); free(t->w);
    for (int i = 0; i < 3; ++i) { free(t->L[i]); free(t->Lt[i]); free(t->v[i]); free(t->wv[i]); }
    free(t->x); free(t->u); free(t->iu); free(t->zh); free(t->gz); free(t->tmp);
    free(t);
}

/* banded = 1: use the diagonals when all three operators are banded Toeplitz (|d| <= 6 up to 1e-19 of the largest entry, diagonals
 * constant to 1e-12), else dense; banded = 0: dense whatever the structure.  Returns NULL on allocation failure. */
tuned_s1 *tuned_s1_create(int nf, int K, int pos, const double *A, const double *L0, const double *L1, const double *L2, const double *Z,
                          const double *freq, double sigma_min, double ups_alpha, double ups_beta, double induc_scale, int banded)
{
    tuned_s1 *t = (tuned_s1 *)calloc(1, sizeof(tuned_s1));
    if (!t) return NULL;
    const int N = 2 * nf;
    t->nf = nf; t->K = K; t->pos = pos;
    t->A = dup(A, (size_t)N * K);
    t->At = (double *)malloc((size_t)N * K * sizeof(double));
    for (int r = 0; r < N; ++r) for (int k = 0; k < K; ++k) t->At[(size_t)k * N + r] = A[(size_t)r * K + k];
    const double *Ls[3] = {L0, L1, L2};
    int ok = banded;
    const int bw = 6;
    for (int i = 0; i < 3; ++i) {
        t->L[i] = dup(Ls[i], (size_t)K * K);
        t->Lt[i] = (double *)malloc((size_t)K * K * sizeof(double));
        double mx = 0.0;
        for (int r = 0; r < K; ++r) for (int c = 0; c < K; ++c) { t->Lt[i][(size_t)c * K + r] = Ls[i][(size_t)r * K + c]; mx = fmax(mx, fabs(Ls[i][(size_t)r * K + c])); }
        for (int d = -(K - 1); d <= K - 1 && ok; ++d) {
            double lo = INFINITY, hi = -INFINITY;
            for (int r = (d < 0 ? -d : 0); r < (d > 0 ? K - d : K); ++r) { const double e = Ls[i][(size_t)r * K + r + d]; lo = fmin(lo, e); hi = fmax(hi, e); }
            if (abs(d) > bw) { if (fmax(fabs(lo), fabs(hi)) > 1e-19 * mx) ok = 0; }
            else { if (hi - lo > 1e-12 * mx) ok = 0; t->taps[i][d + bw] = Ls[i][(size_t)(K / 2) * K + K / 2 + d]; }
        }
    }
    t->banded = ok && K > 2 * bw + 2; t->bw = bw;
    t->Z = dup(Z, (size_t)N);
    t->w = (double *)malloc((size_t)nf * sizeof(double));
    for (int n = 0; n < nf; ++n) t->w[n] = 2.0 * M_PI * freq[n];
    t->sigma_min = sigma_min; t->ups_alpha = ups_alpha; t->ups_beta = ups_beta; t->induc_scale = induc_scale;
    t->x = (double *)malloc((size_t)K * sizeof(double)); t->u = (double *)malloc((size_t)K * sizeof(double)); t->iu = (double *)malloc((size_t)K * sizeof(double));
    for (int i = 0; i < 3; ++i) { t->v[i] = (double *)malloc((size_t)K * sizeof(double)); t->wv[i] = (double *)malloc((size_t)K * sizeof(double)); }
    t->zh = (double *)malloc((size_t)N * sizeof(double)); t->gz = (double *)malloc((size_t)N * sizeof(double));
    t->tmp = (double *)malloc((size_t)(K > N ? K : N) * sizeof(double));
    return t;
}

int tuned_s1_is_banded(const tuned_s1 *t) { return t->banded; }
int tuned_s1_num_params(const tuned_s1 *t) { return 2 * t->K + 9; }

static inline void gemv_rows(const double *restrict M, int r, int c, const double *restrict x, double *restrict y)
{
    /* eight independent partial sums per row (a fixed-trip inner loop: what the vectoriser turns into one 512-bit FMA) */
    enum { W = 8 };
    for (int i = 0; i < r; ++i) {
        const double *restrict row = M + (size_t)i * c;
        double acc[W] = {0, 0, 0, 0, 0, 0, 0, 0};
        int j = 0;
        for (; j + W <= c; j += W)
            for (int l = 0; l < W; ++l) acc[l] += row[j + l] * x[j + l];
        double s = 0.0;
        for (; j < c; ++j) s += row[j] * x[j];
        y[i] = s + ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
}

static inline void band_apply(const double *restrict taps, int bw, int K, const double *restrict x, double *restrict y, int transpose)
{
    /* y[k] = sum_d T[d] x[k + d] (d = -bw..bw inside [0, K)); transpose: y[k] = sum_d T[d] x[k - d] */
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        const int lo = transpose ? (k + bw > K - 1 ? k - (K - 1) : -bw) : (k - bw < 0 ? -k : -bw);
        const int hi = transpose ? (k - bw < 0 ? k : bw) : (k + bw > K - 1 ? K - 1 - k : bw);
        for (int d = lo; d <= hi; ++d) s += taps[d + bw] * x[transpose ? k - d : k + d];
        y[k] = s;
    }
}

/* theta [2K + 9] in Stan declaration order: Rinf_raw, induc_raw, x[K], sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw, ups_raw[K],
 * d0, d1, d2 (all unconstrained: log of the <lower=0> ones).  Returns lp (with the Jacobian of the transforms if `jacobian`). */
double tuned_s1_logp_grad(tuned_s1 *t, const double *restrict theta, int jacobian, double *restrict grad)
{
    const int nf = t->nf, N = 2 * nf, K = t->K;
    const int ox = 2, oe = 2 + K, ou = 6 + K, od = 6 + 2 * K;
    const double jac = jacobian ? 1.0 : 0.0;
    double lp = 0.0;
    double sc[6], dd[3];
    { const int idx[6] = {0, 1, oe, oe + 1, oe + 2, oe + 3}; for (int j = 0; j < 6; ++j) { sc[j] = exp(theta[idx[j]]); lp += -0.5 * sc[j] * sc[j] + jac * theta[idx[j]]; } }
    for (int i = 0; i < 3; ++i) { dd[i] = exp(theta[od + i]); lp += -6.0 * theta[od + i] - 5.0 / dd[i] + jac * theta[od + i]; }
    double *restrict x = t->x, *restrict u = t->u, *restrict iu = t->iu;
    for (int k = 0; k < K; ++k) {
        x[k] = t->pos ? exp(theta[ox + k]) : theta[ox + k];
        if (t->pos) lp += jac * theta[ox + k];
        u[k] = 0.15 * exp(theta[ou + k]);
        iu[k] = 1.0 / u[k];
    }
    /* prior: v_i = L_i x, q^2 = sum d_i v_i^2 ~ normal(0, ups); ups_raw ~ inv_gamma; dups ~ std_normal */
    for (int i = 0; i < 3; ++i) { if (t->banded) band_apply(t->taps[i], t->bw, K, x, t->v[i], 0); else gemv_rows(t->L[i], K, K, x, t->v[i]); }
    double sv[3] = {0.0, 0.0, 0.0};
    double *restrict gx = grad + ox, *restrict gu = grad + ou;
    for (int k = 0; k < K; ++k) {
        const double tu = theta[ou + k], i1 = iu[k], i2 = i1 * i1;
        const double v0 = t->v[0][k], v1 = t->v[1][k], v2 = t->v[2][k];
        const double q2 = dd[0] * v0 * v0 + dd[1] * v1 * v1 + dd[2] * v2 * v2;
        const double ir = 0.15 * i1;
        lp += -(tu + log(0.15)) - 0.5 * q2 * i2 - (t->ups_alpha + 1.0) * tu - t->ups_beta * ir + jac * tu;
        sv[0] += v0 * v0 * i2; sv[1] += v1 * v1 * i2; sv[2] += v2 * v2 * i2;
        double g = -i1 + q2 * i2 * i1;
        if (k >= 1 && k + 1 < K) { const double du = 0.5 * (u[k] - 0.5 * (u[k - 1] + u[k + 1])) * i1; lp += -0.5 * du * du; g += -du * 0.25 * (u[k - 1] + u[k + 1]) * i2; }
        if (k >= 2) { const double i0 = iu[k - 1]; g += 0.5 * (u[k - 1] - 0.5 * (u[k - 2] + u[k])) * i0 * 0.25 * i0; }
        if (k + 2 < K) { const double i0 = iu[k + 1]; g += 0.5 * (u[k + 1] - 0.5 * (u[k] + u[k + 2])) * i0 * 0.25 * i0; }
        gu[k] = u[k] * g - (t->ups_alpha + 1.0) + t->ups_beta * ir + jac;
        t->wv[0][k] = -dd[0] * v0 * i2; t->wv[1][k] = -dd[1] * v1 * i2; t->wv[2][k] = -dd[2] * v2 * i2;
    }
    for (int k = 0; k < K; ++k) gx[k] = 0.0;
    for (int i = 0; i < 3; ++i) {
        if (t->banded) band_apply(t->taps[i], t->bw, K, t->wv[i], t->tmp, 1); else gemv_rows(t->Lt[i], K, K, t->wv[i], t->tmp);
        for (int k = 0; k < K; ++k) gx[k] += t->tmp[k];
        grad[od + i] = -0.5 * dd[i] * sv[i] - 6.0 + 5.0 / dd[i] + jac;
    }
    /* likelihood */
    gemv_rows(t->A, N, K, x, t->zh);
    const double Rinf = 100.0 * sc[0], induc = sc[1] * t->induc_scale, s_res = 0.05 * sc[2], a_p = 0.05 * sc[3], a_r = 0.05 * sc[4], a_i = 0.05 * sc[5];
    const double c0 = t->sigma_min * t->sigma_min + s_res * s_res, ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
    double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
    for (int n = 0; n < nf; ++n) {
        const double wn = t->w[n], zr = t->zh[n] + Rinf, zi = t->zh[nf + n] + induc * wn;
        const double common = ar2 * zr * zr + ai2 * zi * zi;
        const double s2r = c0 + ap2 * zr * zr + common, s2i = c0 + ap2 * zi * zi + common;
        const double er = t->Z[n] - zr, ei = t->Z[nf + n] - zi;
        const double wr = 1.0 / s2r, wi = 1.0 / s2i;
        lp += -0.5 * log(s2r * s2i) - 0.5 * er * er * wr - 0.5 * ei * ei * wi;
        const double hr = -0.5 * wr + 0.5 * er * er * wr * wr, hi = -0.5 * wi + 0.5 * ei * ei * wi * wi;
        const double gzr = er * wr + 2.0 * zr * (hr * (ap2 + ar2) + hi * ar2), gzi = ei * wi + 2.0 * zi * (hi * (ap2 + ai2) + hr * ai2);
        t->gz[n] = gzr; t->gz[nf + n] = gzi;
        sR += gzr; sL += gzi * wn; sH += hr + hi; sHz2 += hr * zr * zr + hi * zi * zi; sHzr2 += (hr + hi) * zr * zr; sHzi2 += (hr + hi) * zi * zi;
    }
    gemv_rows(t->At, K, N, t->gz, t->tmp);
    for (int k = 0; k < K; ++k) { const double graw = t->tmp[k] + gx[k]; gx[k] = t->pos ? x[k] * graw + jac : graw; }
    const double dl[6] = {100.0 * sR, t->induc_scale * sL, 0.1 * s_res * sH, 0.1 * a_p * sHz2, 0.1 * a_r * sHzr2, 0.1 * a_i * sHzi2};
    { const int idx[6] = {0, 1, oe, oe + 1, oe + 2, oe + 3}; for (int j = 0; j < 6; ++j) grad[idx[j]] = sc[j] * (dl[j] - sc[j]) + jac; }
    return lp;
}

/* n evaluations at theta + small perturbations (timing loop of bench.py's cpu_baseline.tuned): returns the sum of the lp values */
double tuned_s1_bench(tuned_s1 *t, const double *theta0, int n, double *grad)
{
    const int D = 2 * t->K + 9;
    double *th = (double *)malloc((size_t)D * sizeof(double));
    double acc = 0.0;
    for (int it = 0; it < n; ++it) {
        for (int j = 0; j < D; ++j) th[j] = theta0[j] + 1e-3 * ((it * 31 + j * 17) % 13 - 6);
        acc += tuned_s1_logp_grad(t, th, 1, grad);
    }
    free(th);
    return acc;
}
