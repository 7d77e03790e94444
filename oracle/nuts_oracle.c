/*
 * nuts_oracle.c -- CPU ORACLE for the sampler (test infrastructure, NOT the product).
 *
 * A plain recursive restatement of Stan 2.19's adapt_diag_e_nuts (the engine behind
 * StanModel.sampling, reference call site bayes_drt/inversion.py:1218-1221; algorithm as recorded in
 * SURVEY.md Appendix A).  pystan 2.19.1.1 is absent from /root/reference and from this image, so this
 * file is "parity unpinned" against Stan itself; it is pinned instead by
 *   - known-answer targets (Gaussians with known moments) through the generic target callback, and
 *   - the reference's committed HMC results (code_EchemActa/bayes_results) within Monte-Carlo error.
 * It consumes random numbers exactly like the HIP sampler (Philox4x32-10 keyed by (seed, chain) and indexed
 * by (iteration, depth, leaf, purpose)) so that the two can be compared draw by draw on short runs, while
 * being structured differently (recursion + direct sums here, checkpointed iteration on the GPU).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bdrt_oracle.h"

typedef double (*orc_target_fn)(void *ctx, const double *theta, double *grad);   /* returns lp (jacobian on) */

typedef struct {
    double adapt_delta, adapt_t0, adapt_gamma, adapt_kappa;
    int max_treedepth, init_buffer, term_buffer, base_window;
    double init_radius, max_deltaH, stepsize0;
} orc_nuts_control;

typedef struct {
    long long n_leapfrog;
    int n_divergent, n_max_treedepth;
    double stepsize, mean_accept;
} orc_chain_diag;

enum { RNG_INIT = 1, RNG_MOMENTUM = 2, RNG_DIRECTION = 3, RNG_LEAF = 4, RNG_TOP = 5, RNG_EPS_MOMENTUM = 6 };

static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint32_t k0, k1, chain; } rng_t;

static void uniform2(const rng_t *g, uint32_t index, uint32_t purpose, uint32_t depth, uint32_t trial, uint32_t iter,
                     double *u0, double *u1)
{
    uint32_t o[4];
    philox(index, purpose | (depth << 8) | (trial << 16), iter, g->chain, g->k0, g->k1, o);
    uint64_t a = ((uint64_t)o[0] << 32) | o[1], b = ((uint64_t)o[2] << 32) | o[3];
    *u0 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    *u1 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static double uniform1(const rng_t *g, uint32_t index, uint32_t purpose, uint32_t depth, uint32_t trial, uint32_t iter)
{
    double a, b;
    uniform2(g, index, purpose, depth, trial, iter, &a, &b);
    return a;
}
static double normal1(const rng_t *g, uint32_t j, uint32_t purpose, uint32_t trial, uint32_t iter)
{
    double u0, u1;
    uniform2(g, j >> 1, purpose, 0, trial, iter, &u0, &u1);
    double r = sqrt(-2.0 * log(u0)), a = 6.283185307179586476925286766559 * u1;
    return (j & 1) ? r * sin(a) : r * cos(a);
}
static double lse2(double a, double b)
{
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}

typedef struct {
    int D;
    orc_target_fn fn; void *fctx;
    const orc_nuts_control *c;
    rng_t rng;
    double *minv;
    /* moving point */
    double *th, *p, *g; double lp;
    /* per-transition */
    double eps, H0; int dir, iter, depth_top;
    int leaf;              /* leaf counter inside the subtree being built */
    double lsw_sub;        /* log sum of weights of the new subtree */
    double *thq, *gq; double lpq;   /* proposal of the new subtree */
    long long n_leap; double sum_metro; int divergent;
    long long total_evals;
} nctx;

static void leapfrog(nctx *x, double e)
{
    const int D = x->D;
    for (int j = 0; j < D; ++j) { x->p[j] += 0.5 * e * x->g[j]; x->th[j] += e * x->minv[j] * x->p[j]; }
    x->lp = x->fn(x->fctx, x->th, x->g);
    x->total_evals += 1;
    for (int j = 0; j < D; ++j) x->p[j] += 0.5 * e * x->g[j];
}
static double kinetic(const nctx *x)
{
    double s = 0;
    for (int j = 0; j < x->D; ++j) s += x->minv[j] * x->p[j] * x->p[j];
    return 0.5 * s;
}
static int criterion(const nctx *x, const double *psl, const double *psr, const double *rho)
{
    double a = 0, b = 0;
    for (int j = 0; j < x->D; ++j) { a += psl[j] * rho[j]; b += psr[j] * rho[j]; }
    return a > 0 && b > 0;
}

/* Stan base_nuts::build_tree; the proposal inside the subtree is drawn by weighted reservoir sampling over
 * the leaves in generation order (same multinomial distribution as Stan's pairwise merges). */
static int build_tree(nctx *x, int depth, double *rho, double *psl, double *psr)
{
    const int D = x->D;
    if (depth == 0) {
        leapfrog(x, x->dir * x->eps);
        x->n_leap += 1;
        double h = -x->lp + kinetic(x);
        if (isnan(h)) h = INFINITY;
        const double w = x->H0 - h;
        x->sum_metro += w > 0 ? 1.0 : exp(w);
        if ((h - x->H0) > x->c->max_deltaH) { x->divergent = 1; return 0; }
        const double lsw_new = lse2(x->lsw_sub, w);
        const double u = uniform1(&x->rng, (uint32_t)x->leaf, RNG_LEAF, (uint32_t)x->depth_top, 0, (uint32_t)x->iter);
        if (x->leaf == 0 || u < exp(w - lsw_new)) {
            memcpy(x->thq, x->th, sizeof(double) * D); memcpy(x->gq, x->g, sizeof(double) * D); x->lpq = x->lp;
        }
        x->lsw_sub = lsw_new;
        x->leaf += 1;
        for (int j = 0; j < D; ++j) { rho[j] += x->p[j]; psl[j] = psr[j] = x->minv[j] * x->p[j]; }
        return 1;
    }
    double *rho_l = (double *)calloc((size_t)D, sizeof(double)), *rho_r = (double *)calloc((size_t)D, sizeof(double));
    double *dummy = (double *)malloc(sizeof(double) * (size_t)D);
    int ok = build_tree(x, depth - 1, rho_l, psl, dummy);
    if (ok) ok = build_tree(x, depth - 1, rho_r, dummy, psr);
    if (ok) {
        for (int j = 0; j < D; ++j) { rho_l[j] += rho_r[j]; rho[j] += rho_l[j]; }
        ok = criterion(x, psl, psr, rho_l);
    }
    free(rho_l); free(rho_r); free(dummy);
    return ok;
}

/* windowed adaptation bookkeeping (stan/mcmc/windowed_adaptation.hpp) */
typedef struct { int counter, size, next, init_buffer, term_buffer, base_window; } win_t;
static void win_init(win_t *w, int warmup, int ib, int tb, int bw)
{
    /* num_warmup < 20: stan::mcmc::windowed_adaptation::set_window_params returns early, its unsigned next-window index
       stays at UINT_MAX and neither the metric nor the step-size restart is ever triggered */
    const int no_metric = warmup < 20;
    if (no_metric) { ib = warmup; tb = 0; bw = 0; }
    else if (ib + bw + tb > warmup) { ib = (int)(0.15 * warmup); tb = (int)(0.1 * warmup); bw = warmup - (ib + tb); }
    w->init_buffer = ib; w->term_buffer = tb; w->base_window = bw;
    w->counter = 0; w->size = bw; w->next = no_metric ? -1 : ib + bw - 1;
}
static int win_active(const win_t *w, int warmup) { return w->counter >= w->init_buffer && w->counter < warmup - w->term_buffer && w->counter != warmup; }
static int win_end(const win_t *w, int warmup) { return w->counter == w->next && w->counter != warmup; }
static void win_next(win_t *w, int warmup)
{
    if (w->next == warmup - w->term_buffer - 1) return;
    w->size *= 2;
    w->next = w->counter + w->size;
    if (w->next == warmup - w->term_buffer - 1) return;
    int boundary = w->next + 2 * w->size;
    if (boundary >= warmup - w->term_buffer) w->next = warmup - w->term_buffer - 1;
}

void orc_nuts_defaults(orc_nuts_control *c)
{
    c->adapt_delta = 0.9; c->adapt_t0 = 10; c->adapt_gamma = 0.05; c->adapt_kappa = 0.75;
    c->max_treedepth = 10; c->init_buffer = 75; c->term_buffer = 50; c->base_window = 25;
    c->init_radius = 2; c->max_deltaH = 1000; c->stepsize0 = 1;
}

/* Stan base_hmc::init_stepsize from the point (ths, gs, lps); trial momenta from (iter, trial) */
static void init_stepsize(nctx *x, const double *ths, const double *gs, double lps, double *eps, int iter)
{
    const int D = x->D;
    const double thr = log(0.8);
    int dir = 0;
    for (int trial = 0; trial < 10000; ++trial) {
        for (int j = 0; j < D; ++j) {
            x->p[j] = normal1(&x->rng, (uint32_t)j, RNG_EPS_MOMENTUM, (uint32_t)trial, (uint32_t)iter) / sqrt(x->minv[j]);
            x->th[j] = ths[j]; x->g[j] = gs[j];
        }
        const double H0 = -lps + kinetic(x);
        leapfrog(x, *eps);
        double h = -x->lp + kinetic(x);
        if (isnan(h)) h = INFINITY;
        const double dH = H0 - h;
        if (trial == 0) { dir = dH > thr ? 1 : -1; continue; }
        if (dir == 1 && !(dH > thr)) break;
        if (dir == -1 && !(dH < thr)) break;
        *eps = dir == 1 ? 2.0 * (*eps) : 0.5 * (*eps);
        if (!(*eps > 1e-300) || *eps > 1e7) break;
    }
}

int orc_nuts_sample_generic(int D, orc_target_fn fn, void *fctx, int chain_id, uint64_t seed, int warmup, int n_draws,
                            const double *init_theta, const orc_nuts_control *ctrl, double *draws, double *lp_draws,
                            orc_chain_diag *diag)
{
    orc_nuts_control cdef;
    if (!ctrl) { orc_nuts_defaults(&cdef); ctrl = &cdef; }
    nctx x;
    memset(&x, 0, sizeof(x));
    x.D = D; x.fn = fn; x.fctx = fctx; x.c = ctrl;
    x.rng.k0 = (uint32_t)seed; x.rng.k1 = (uint32_t)(seed >> 32); x.rng.chain = (uint32_t)chain_id;
    const size_t nb = sizeof(double) * (size_t)D;
    x.minv = (double *)malloc(nb); x.th = (double *)malloc(nb); x.p = (double *)malloc(nb); x.g = (double *)malloc(nb);
    x.thq = (double *)malloc(nb); x.gq = (double *)malloc(nb);
    double *ths = (double *)malloc(nb), *gs = (double *)malloc(nb);
    double *thm = (double *)malloc(nb), *pm = (double *)malloc(nb), *gm = (double *)malloc(nb);
    double *thp = (double *)malloc(nb), *pp = (double *)malloc(nb), *gp = (double *)malloc(nb);
    double *rho = (double *)malloc(nb), *rho_sub = (double *)malloc(nb), *psl = (double *)malloc(nb), *psr = (double *)malloc(nb);
    double *wmean = (double *)calloc((size_t)D, sizeof(double)), *wm2 = (double *)calloc((size_t)D, sizeof(double));
    for (int j = 0; j < D; ++j) x.minv[j] = 1.0;
    double lps = 0;
    int rc = 0;

    /* initial point: given, or U(-r, r) retried until lp and gradient are finite */
    int ok = 0;
    for (int att = 0; att < 100 && !ok; ++att) {
        for (int j = 0; j < D; ++j)
            ths[j] = (att == 0 && init_theta) ? init_theta[j]
                                              : ctrl->init_radius * (2.0 * uniform1(&x.rng, (uint32_t)j, RNG_INIT, 0, (uint32_t)att, 0) - 1.0);
        lps = fn(fctx, ths, gs);
        x.total_evals += 1;
        ok = isfinite(lps);
        for (int j = 0; j < D && ok; ++j) ok = isfinite(gs[j]);
    }
    if (!ok) { rc = -1; goto done; }

    {
        double eps = ctrl->stepsize0;
        /* dual averaging state */
        int da_counter = 0; double da_sbar = 0, da_xbar = 0, da_mu = log(10.0 * ctrl->stepsize0);
        win_t win; win_init(&win, warmup, ctrl->init_buffer, ctrl->term_buffer, ctrl->base_window);
        int win_n = 0;
        long long leap_total = 0; int n_div = 0, n_maxd = 0, n_post = 0; double sum_acc = 0;

        init_stepsize(&x, ths, gs, lps, &eps, 0);

        const int total = warmup + n_draws;
        for (int iter = 0; iter < total; ++iter) {
            x.iter = iter; x.eps = eps;
            for (int j = 0; j < D; ++j) {
                x.p[j] = normal1(&x.rng, (uint32_t)j, RNG_MOMENTUM, 0, (uint32_t)iter) / sqrt(x.minv[j]);
                x.th[j] = ths[j]; x.g[j] = gs[j];
            }
            x.lp = lps;
            x.H0 = -lps + kinetic(&x);
            memcpy(thm, x.th, nb); memcpy(thp, x.th, nb); memcpy(pm, x.p, nb); memcpy(pp, x.p, nb);
            memcpy(gm, x.g, nb); memcpy(gp, x.g, nb); memcpy(rho, x.p, nb);
            double lsw = 0.0;
            int depth = 0;
            x.n_leap = 0; x.sum_metro = 0; x.divergent = 0;
            while (depth < ctrl->max_treedepth) {
                x.dir = uniform1(&x.rng, 0, RNG_DIRECTION, (uint32_t)depth, 0, (uint32_t)iter) > 0.5 ? 1 : -1;
                x.depth_top = depth; x.leaf = 0; x.lsw_sub = -INFINITY;
                if (x.dir > 0) { memcpy(x.th, thp, nb); memcpy(x.p, pp, nb); memcpy(x.g, gp, nb); }
                else { memcpy(x.th, thm, nb); memcpy(x.p, pm, nb); memcpy(x.g, gm, nb); }
                memset(rho_sub, 0, nb);
                const int valid = build_tree(&x, depth, rho_sub, psl, psr);
                if (!valid) break;
                if (x.dir > 0) { memcpy(thp, x.th, nb); memcpy(pp, x.p, nb); memcpy(gp, x.g, nb); }
                else { memcpy(thm, x.th, nb); memcpy(pm, x.p, nb); memcpy(gm, x.g, nb); }
                depth += 1;
                int take;
                if (x.lsw_sub > lsw) take = 1;
                else take = uniform1(&x.rng, 0, RNG_TOP, (uint32_t)depth, 0, (uint32_t)iter) < exp(x.lsw_sub - lsw);
                if (take) { memcpy(ths, x.thq, nb); memcpy(gs, x.gq, nb); lps = x.lpq; }
                lsw = lse2(lsw, x.lsw_sub);
                for (int j = 0; j < D; ++j) rho[j] += rho_sub[j];
                double a = 0, b = 0;
                for (int j = 0; j < D; ++j) { a += x.minv[j] * pm[j] * rho[j]; b += x.minv[j] * pp[j] * rho[j]; }
                if (!(a > 0 && b > 0)) break;
            }
            const double accept = x.sum_metro / (double)(x.n_leap > 0 ? x.n_leap : 1);
            leap_total += x.n_leap;
            const int warm = iter < warmup;
            if (!warm) {
                n_post += 1; sum_acc += accept;
                if (x.divergent) n_div += 1;
                if (depth >= ctrl->max_treedepth) n_maxd += 1;
                if (draws) memcpy(draws + (size_t)(iter - warmup) * D, ths, nb);
                if (lp_draws) lp_draws[iter - warmup] = lps;
            } else {
                /* stepsize_adaptation::learn_stepsize */
                da_counter += 1;
                const double acc1 = accept > 1.0 ? 1.0 : accept;
                const double eta = 1.0 / (da_counter + ctrl->adapt_t0);
                da_sbar = (1.0 - eta) * da_sbar + eta * (ctrl->adapt_delta - acc1);
                const double xx = da_mu - da_sbar * sqrt((double)da_counter) / ctrl->adapt_gamma;
                const double x_eta = pow((double)da_counter, -ctrl->adapt_kappa);
                da_xbar = (1.0 - x_eta) * da_xbar + x_eta * xx;
                eps = exp(xx);
                /* var_adaptation::learn_variance */
                int update = 0;
                if (win_active(&win, warmup)) {
                    win_n += 1;
                    for (int j = 0; j < D; ++j) {
                        const double delta = ths[j] - wmean[j];
                        wmean[j] += delta / win_n;
                        wm2[j] += (ths[j] - wmean[j]) * delta;
                    }
                }
                if (win_end(&win, warmup)) {
                    win_next(&win, warmup);
                    const double n = win_n;
                    for (int j = 0; j < D; ++j) {
                        const double var = n > 1.0 ? wm2[j] / (n - 1.0) : 0.0;
                        x.minv[j] = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
                        wmean[j] = 0; wm2[j] = 0;
                    }
                    win_n = 0;
                    update = 1;
                }
                win.counter += 1;
                if (iter + 1 == warmup) eps = exp(da_xbar);
                else if (update) {
                    init_stepsize(&x, ths, gs, lps, &eps, iter + 1);
                    da_mu = log(10.0 * eps); da_counter = 0; da_sbar = 0; da_xbar = 0;
                }
            }
        }
        if (diag) {
            diag->n_leapfrog = leap_total; diag->n_divergent = n_div; diag->n_max_treedepth = n_maxd;
            diag->stepsize = eps; diag->mean_accept = n_post ? sum_acc / n_post : 0.0;
        }
    }
done:
    free(x.minv); free(x.th); free(x.p); free(x.g); free(x.thq); free(x.gq); free(ths); free(gs);
    free(thm); free(pm); free(gm); free(thp); free(pp); free(gp); free(rho); free(rho_sub); free(psl); free(psr);
    free(wmean); free(wm2);
    return rc;
}

/* the DRT posterior as a target */
static double model_target(void *ctx, const double *theta, double *grad)
{
    double lp;
    orc_logp_grad((const orc_model *)ctx, theta, 1, &lp, grad);
    return lp;
}

int orc_nuts_sample(const orc_model *m, int chain_id, uint64_t seed, int warmup, int n_draws, const double *init_theta,
                    const orc_nuts_control *ctrl, double *draws, double *lp_draws, orc_chain_diag *diag)
{
    return orc_nuts_sample_generic(orc_num_params(m), model_target, (void *)m, chain_id, seed, warmup, n_draws,
                                   init_theta, ctrl, draws, lp_draws, diag);
}

/* known-answer target: independent Gaussians N(mu_j, sd_j^2); ctx = [D, mu[D], sd[D]] as doubles */
static double gauss_target(void *ctx, const double *theta, double *grad)
{
    const double *c = (const double *)ctx;
    const int D = (int)c[0];
    double lp = 0;
    for (int j = 0; j < D; ++j) {
        const double z = (theta[j] - c[1 + j]) / c[1 + D + j];
        lp += -0.5 * z * z;
        grad[j] = -z / c[1 + D + j];
    }
    return lp;
}
int orc_nuts_sample_gauss(const double *desc, int chain_id, uint64_t seed, int warmup, int n_draws,
                          const orc_nuts_control *ctrl, double *draws, orc_chain_diag *diag)
{
    return orc_nuts_sample_generic((int)desc[0], gauss_target, (void *)desc, chain_id, seed, warmup, n_draws, 0, ctrl,
                                   draws, 0, diag);
}

/* evaluation-rate helper for bench.py's cpu_baseline leg: n evaluations of log_prob+grad at jittered points;
 * returns a checksum so the loop cannot be optimised away */
double orc_eval_loop(const orc_model *m, const double *theta0, int n, int jacobian)
{
    const int D = orc_num_params(m);
    double *th = (double *)malloc(sizeof(double) * (size_t)D), *g = (double *)malloc(sizeof(double) * (size_t)D);
    memcpy(th, theta0, sizeof(double) * (size_t)D);
    double acc = 0, lp;
    for (int i = 0; i < n; ++i) {
        th[i % D] += 1e-9;
        orc_logp_grad(m, th, jacobian, &lp, g);
        acc += lp + g[i % D];
    }
    free(th); free(g);
    return acc;
}
