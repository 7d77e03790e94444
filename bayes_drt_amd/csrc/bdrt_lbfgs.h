// bdrt_lbfgs.h -- host-side L-BFGS state machine (no HIP dependency: also compiled by tests/host/ on the CPU
// to test the optimiser logic on analytic functions; the product only ever feeds it GPU evaluations).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/bdrt.h"

namespace bdrt {

struct LbfgsFit {
    enum Phase { START, BRACKET, ZOOM, DONE };
    int D = 0;
    const bdrt_opt_options *opt = nullptr;
    Phase phase = START;
    int iters = 0, n_evals = 0, rc = 1;
    bool reset_dir = true;            // next direction is steepest descent
    int ls_restarts = 0;
    std::vector<double> x, g, xt, gt, p, x_prev, g_prev;
    double f = 0.0, ft = 0.0;
    // history
    std::deque<std::vector<double>> S, Y;
    std::deque<double> RHO;
    // line search
    double phi0 = 0, dphi0 = 0, alpha = 0, alpha_prev = 0, phi_prev = 0, dphi_prev = 0;
    double a_lo = 0, a_hi = 0, phi_lo = 0, phi_hi = 0, dphi_lo = 0, dphi_hi = 0;
    int ls_iter = 0;
    double last_alpha = 0, last_dphi0 = 0, last_df = 0;
    bool have_last = false;

    static double dot(const std::vector<double> &a, const std::vector<double> &b)
    {
        double s = 0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }

    void init(int D_, const double *x0, const bdrt_opt_options *o)
    {
        D = D_; opt = o;
        x.assign(x0, x0 + D); g.assign(D, 0.0); xt = x; gt = g; p = g; x_prev = x; g_prev = g;
        phase = START;
    }
    const double *trial() const { return phase == START ? x.data() : xt.data(); }

    void two_loop()     // p = -H g
    {
        std::vector<double> q = g;
        const int m = (int)S.size();
        std::vector<double> al(m);
        for (int i = m - 1; i >= 0; --i) {
            al[i] = RHO[i] * dot(S[i], q);
            for (int j = 0; j < D; ++j) q[j] -= al[i] * Y[i][j];
        }
        double gamma = 1.0;
        if (m > 0) gamma = dot(S[m - 1], Y[m - 1]) / dot(Y[m - 1], Y[m - 1]);
        for (int j = 0; j < D; ++j) q[j] *= gamma;
        for (int i = 0; i < m; ++i) {
            const double be = RHO[i] * dot(Y[i], q);
            for (int j = 0; j < D; ++j) q[j] += S[i][j] * (al[i] - be);
        }
        for (int j = 0; j < D; ++j) p[j] = -q[j];
    }

    static double cubic_min(double a, double fa, double dfa, double b, double fb, double dfb)
    {
        // minimiser of the cubic interpolating (a, fa, dfa), (b, fb, dfb) (Nocedal & Wright eq. 3.59)
        const double d1 = dfa + dfb - 3.0 * (fa - fb) / (a - b);
        const double rad = d1 * d1 - dfa * dfb;
        if (!(rad >= 0.0) || !std::isfinite(d1)) return 0.5 * (a + b);
        const double d2 = (b > a ? 1.0 : -1.0) * std::sqrt(rad);
        const double den = dfb - dfa + 2.0 * d2;
        if (den == 0.0 || !std::isfinite(den)) return 0.5 * (a + b);
        return b - (b - a) * (dfb + d2 - d1) / den;
    }

    void start_linesearch()
    {
        if (reset_dir || S.empty()) for (int j = 0; j < D; ++j) p[j] = -g[j];
        else two_loop();
        dphi0 = dot(g, p);
        if (!(dphi0 < 0.0)) {               // not a descent direction: fall back to steepest descent
            for (int j = 0; j < D; ++j) p[j] = -g[j];
            dphi0 = -dot(g, g);
            S.clear(); Y.clear(); RHO.clear();
        }
        phi0 = f;
        double a0;
        // first trial step: Stan uses init_alpha on the first iteration (and after a reset), then
        // min(1, 1.01*cubic-interpolant of the previous step); the unit step of the textbook L-BFGS measured
        // equal or better on the tests/host functions and is used for iterations after the first.
        a0 = (have_last && !reset_dir) ? 1.0 : opt->init_alpha;
        alpha = a0; alpha_prev = 0.0; phi_prev = phi0; dphi_prev = dphi0;
        ls_iter = 0;
        phase = BRACKET;
        set_trial();
    }
    void set_trial()
    {
        for (int j = 0; j < D; ++j) xt[j] = x[j] + alpha * p[j];
    }

    bool finish_iteration()   // accept (xt, ft, gt); returns true if the fit terminated
    {
        last_alpha = alpha; last_dphi0 = dphi0; last_df = ft - f; have_last = true;
        std::vector<double> s(D), y(D);
        double snorm2 = 0;
        for (int j = 0; j < D; ++j) { s[j] = xt[j] - x[j]; y[j] = gt[j] - g[j]; snorm2 += s[j] * s[j]; }
        const double f_old = f;
        x_prev = x; g_prev = g;
        x = xt; g = gt; f = ft;
        iters += 1;
        const double sy = dot(s, y);
        if (sy > 1e-300 * std::max(1.0, dot(y, y))) {
            S.push_back(s); Y.push_back(y); RHO.push_back(1.0 / sy);
            if ((int)S.size() > opt->history) { S.pop_front(); Y.pop_front(); RHO.pop_front(); }
        }
        reset_dir = false;
        ls_restarts = 0;
        // termination (Stan BFGSMinimizer::step)
        const double eps = DBL_EPSILON;
        const double gnorm = std::sqrt(dot(g, g));
        if (std::fabs(f - f_old) < opt->tol_obj) { rc = 0; return true; }
        if (std::fabs(f - f_old) / std::max(std::max(std::fabs(f), std::fabs(f_old)), 1.0) < opt->tol_rel_obj * eps) { rc = 0; return true; }
        if (gnorm < opt->tol_grad) { rc = 0; return true; }
        {
            two_loop();                                   // p = -H g  ->  g^T H g = -g.p
            const double gHg = -dot(g, p);
            if (gHg / std::max(std::fabs(f), 1.0) < opt->tol_rel_grad * eps) { rc = 0; return true; }
        }
        if (std::sqrt(snorm2) < opt->tol_param) { rc = 0; return true; }
        if (iters >= opt->max_iter) { rc = 1; return true; }
        return false;
    }

    void linesearch_failed()
    {
        if (!reset_dir && ls_restarts < 2) {     // retry once from steepest descent with a cleared history
            S.clear(); Y.clear(); RHO.clear();
            reset_dir = true; have_last = false; ls_restarts += 1;
            start_linesearch();
        } else {
            rc = -2;                            // line search failed to make progress
            phase = DONE;
        }
    }

    // feed the evaluation at trial(): fval = -lp, grad = -grad lp
    void feed(double lp, const double *grad_lp)
    {
        n_evals += 1;
        const double c1 = 1e-4, c2 = 0.9;
        if (phase == START) {
            f = -lp;
            for (int j = 0; j < D; ++j) g[j] = -grad_lp[j];
            if (!std::isfinite(f)) { rc = -1; phase = DONE; return; }
            if (std::sqrt(dot(g, g)) < opt->tol_grad) { rc = 0; phase = DONE; return; }
            reset_dir = true;
            start_linesearch();
            return;
        }
        ft = -lp;
        bool finite = std::isfinite(ft);
        for (int j = 0; j < D; ++j) { gt[j] = -grad_lp[j]; finite = finite && std::isfinite(gt[j]); }
        const double dphi = finite ? dot(gt, p) : 0.0;
        ls_iter += 1;
        if (phase == BRACKET) {
            if (!finite || ft > phi0 + c1 * alpha * dphi0 || (ls_iter > 1 && ft >= phi_prev)) {
                a_lo = alpha_prev; phi_lo = phi_prev; dphi_lo = dphi_prev;
                a_hi = alpha; phi_hi = finite ? ft : INFINITY; dphi_hi = dphi;
                phase = ZOOM;
            } else if (std::fabs(dphi) <= -c2 * dphi0) {
                if (finish_iteration()) { phase = DONE; return; }
                start_linesearch();
                return;
            } else if (dphi >= 0.0) {
                a_lo = alpha; phi_lo = ft; dphi_lo = dphi;
                a_hi = alpha_prev; phi_hi = phi_prev; dphi_hi = dphi_prev;
                phase = ZOOM;
            } else {
                alpha_prev = alpha; phi_prev = ft; dphi_prev = dphi;
                alpha *= 4.0;
                if (ls_iter > 40 || alpha > 1e10) { linesearch_failed(); return; }
                set_trial();
                return;
            }
        } else {   // ZOOM
            if (!finite || ft > phi0 + c1 * alpha * dphi0 || ft >= phi_lo) {
                a_hi = alpha; phi_hi = finite ? ft : INFINITY; dphi_hi = dphi;
            } else {
                if (std::fabs(dphi) <= -c2 * dphi0) {
                    if (finish_iteration()) { phase = DONE; return; }
                    start_linesearch();
                    return;
                }
                if (dphi * (a_hi - a_lo) >= 0.0) { a_hi = a_lo; phi_hi = phi_lo; dphi_hi = dphi_lo; }
                a_lo = alpha; phi_lo = ft; dphi_lo = dphi;
            }
        }
        // next zoom trial
        if (ls_iter > 60 || std::fabs(a_hi - a_lo) <= 1e-16 * std::max(1.0, std::fabs(a_lo))) {
            // accept the best point found if it decreases f at all
            if (a_lo > 0.0 && phi_lo < phi0) {
                alpha = a_lo;
                set_trial();
                // re-evaluation needed to have the gradient at a_lo: mark and wait for it
                phase = BRACKET; ls_iter = 1000;     // sentinel: accept whatever comes back
                force_accept = true;
                return;
            }
            linesearch_failed();
            return;
        }
        double a_new;
        if (std::isfinite(phi_hi) && std::isfinite(dphi_hi)) a_new = cubic_min(a_lo, phi_lo, dphi_lo, a_hi, phi_hi, dphi_hi);
        else a_new = 0.5 * (a_lo + a_hi);
        const double lo = std::min(a_lo, a_hi), hi = std::max(a_lo, a_hi), w = hi - lo;
        if (!(a_new > lo + 0.1 * w) || !(a_new < hi - 0.1 * w) || !std::isfinite(a_new)) a_new = 0.5 * (a_lo + a_hi);
        alpha = a_new;
        set_trial();
    }
    bool force_accept = false;

    // entry point for callers: routes the evaluation at trial() to the state machine
    void feed_any(double lp, const double *grad_lp)
    {
        if (force_accept) {
            // gradient at the best point of an exhausted zoom: accept it if it decreases f at all
            force_accept = false;
            n_evals += 1;
            ft = -lp;
            for (int j = 0; j < D; ++j) gt[j] = -grad_lp[j];
            if (!std::isfinite(ft) || !(ft < f)) { linesearch_failed(); return; }
            if (finish_iteration()) { phase = DONE; return; }
            start_linesearch();
            return;
        }
        feed(lp, grad_lp);
    }
};

}  // namespace bdrt
