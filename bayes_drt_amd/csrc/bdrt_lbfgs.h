// bdrt_lbfgs.h -- host-side L-BFGS state machine (no HIP dependency: also compiled by tests/host/ on the CPU
// to test the optimiser logic on analytic functions; the product only ever feeds it GPU evaluations).
//
// What it restates: the optimiser behind `StanModel.optimizing` (reference bayes_drt/inversion.py:1216), i.e. Stan 2.19.1's
// BFGSMinimizer<..., LBFGSUpdate<5>> -- pystan==2.19.1.1 is a third-party dependency that is absent from /root/reference, so
// this follows the published algorithm, step for step:
//   * step(): the first iteration and every iteration after a failed line search start from steepest descent with the trial
//     step `init_alpha` (1e-3); every other iteration starts its line search at min(1, 1.01 * the minimiser of the cubic through
//     the previous step) -- usually far below the textbook unit step on this posterior;
//   * WolfeLineSearch: strong Wolfe conditions (c1 = 1e-4, c2 = 0.9), bracket growth x10, at most 20 bracketing steps, a
//     non-finite evaluation halves the step towards the last good one (at most 10 times); zoom by cubic interpolation kept
//     1 % inside the bracket, bisection every fifth step, bracket width floor 1e-16;
//   * a failed line search resets the history and retries from steepest descent; failing again ends the run (Stan:
//     TERM_LSFAIL);
//   * after a reset the first pair rescales the initial Hessian (B0fact = y.y / s.y) and the remembered step length;
//   * termination tests in Stan's order: |df| < tol_obj; |df| < tol_rel_obj * eps * max(|f|, |f_prev|, 1); |g| < tol_grad;
//     g^T H g / max(|f|, 1) < tol_rel_grad * eps (H = the L-BFGS inverse Hessian); |dx| < tol_param; iteration cap.
// Stan's exact iterate path is not reproducible (summation order of its autodiff gradient): the contract is the
// same kind of iterate -- one that stops by a tolerance test long before a stationary point on this ill-conditioned posterior.
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
    // return codes: 0 a convergence test fired, 1 iteration cap, -1 no finite start, -2 line search failed twice (TERM_LSFAIL)
    static constexpr double C1 = 1e-4, C2 = 0.9, MIN_ALPHA = 1e-12, MIN_RANGE = 1e-16;
    static constexpr int MAX_LS_ITS = 20, MAX_LS_RESTARTS = 10;

    int D = 0;
    const bdrt_opt_options *opt = nullptr;
    Phase phase = START;
    int iters = 0, n_evals = 0, rc = 1;
    int ls_fail_resets = 0;           // how often a failed line search reset the history (diagnostic)
    std::vector<double> x, g, xt, gt, p;
    double f = 0.0, ft = 0.0;
    // history (newest at the back)
    std::deque<std::vector<double>> S, Y;
    std::deque<double> RHO;
    double gammak = 1.0;
    // the previous accepted step, for the first trial step of the next line search
    double alpha = 0.0;               // Stan's _alpha: last accepted step length (rescaled after a reset)
    double prev_dfp = 0.0;            // g_{k-1} . p_{k-1} (rescaled likewise)
    double f_prev = 0.0;
    int reset = 1;                    // Stan's resetB: 1 first iteration, 2 after a failed line search, 0 otherwise
    // line search
    double dfp = 0, c1dfp = 0, c2dfp = 0;
    double ls_alpha0 = 0, ls_prevF = 0, ls_prevDFp = 0;
    int ls_its = 0, ls_restarts = 0;
    double alo = 0, aloF = 0, aloDFp = 0, ahi = 0, ahiF = 0, ahiDFp = 0;
    int zoom_its = 0;

    static double dot(const std::vector<double> &a, const std::vector<double> &b)
    {
        double s = 0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }

    void init(int D_, const double *x0, const bdrt_opt_options *o)
    {
        D = D_; opt = o;
        x.assign(x0, x0 + D); g.assign(D, 0.0); xt = x; gt = g; p = g;
        phase = START;
    }
    const double *trial() const { return phase == START ? x.data() : xt.data(); }

    void search_direction()     // p = -H g (two-loop recursion, initial scaling gammak)
    {
        std::vector<double> q = g;
        const int m = (int)S.size();
        std::vector<double> al(m);
        for (int i = m - 1; i >= 0; --i) {
            al[i] = RHO[i] * dot(S[i], q);
            for (int j = 0; j < D; ++j) q[j] -= al[i] * Y[i][j];
        }
        for (int j = 0; j < D; ++j) q[j] *= gammak;
        for (int i = 0; i < m; ++i) {
            const double be = RHO[i] * dot(Y[i], q);
            for (int j = 0; j < D; ++j) q[j] += S[i][j] * (al[i] - be);
        }
        for (int j = 0; j < D; ++j) p[j] = -q[j];
    }

    // Minimiser over [loX, hiX] of the cubic c(t) with c(0) = 0, c'(0) = df0, c(x1) = f1, c'(x1) = df1 (Stan's CubicInterp)
    static double cubic_interp(double df0, double x1, double f1, double df1, double loX, double hiX)
    {
        const double c3 = (-12.0 * f1 + 6.0 * x1 * (df0 + df1)) / (x1 * x1 * x1);
        const double c2 = -(4.0 * df0 + 2.0 * df1) / x1 + 6.0 * f1 / (x1 * x1);
        const double c1 = df0;
        const double t_s = std::sqrt(c2 * c2 - 2.0 * c1 * c3);
        const double s1 = -(c2 + t_s) / c3, s2 = -(c2 - t_s) / c3;
        auto val = [&](double t) { return t * (t * (t * c3 / 3.0 + c2) / 2.0 + c1); };
        double minF = val(loX), minX = loX;
        double tmp = val(hiX);
        if (tmp < minF) { minF = tmp; minX = hiX; }
        if (loX < s1 && s1 < hiX) { tmp = val(s1); if (tmp < minF) { minF = tmp; minX = s1; } }
        if (loX < s2 && s2 < hiX) { tmp = val(s2); if (tmp < minF) { minF = tmp; minX = s2; } }
        return minX;
    }

    void set_trial(double a)
    {
        alpha = a;
        for (int j = 0; j < D; ++j) xt[j] = x[j] + a * p[j];
    }

    // one iteration of Stan's step(): choose the direction / first trial step and enter the line search
    void begin_iteration(int reset_code)
    {
        reset = reset_code;
        if (reset) for (int j = 0; j < D; ++j) p[j] = -g[j];
        dfp = dot(g, p);
        double a0;
        if (iters > 0 && reset != 2) {
            a0 = std::min(1.0, 1.01 * cubic_interp(prev_dfp, alpha, f - f_prev, dfp, MIN_ALPHA, 1.0));
            if (!(a0 > 0.0) || !std::isfinite(a0)) a0 = opt->init_alpha;      // (a NaN cubic: Stan would fail the search and reset)
        } else {
            a0 = opt->init_alpha;
        }
        c1dfp = C1 * dfp; c2dfp = C2 * dfp;
        ls_alpha0 = MIN_ALPHA; ls_prevF = f; ls_prevDFp = dfp;
        ls_its = 0; ls_restarts = 0;
        phase = BRACKET;
        set_trial(a0);
    }

    void linesearch_failed()
    {
        if (reset) { rc = -2; phase = DONE; return; }       // already from steepest descent: nothing else to try
        S.clear(); Y.clear(); RHO.clear(); gammak = 1.0;
        ls_fail_resets += 1;
        begin_iteration(2);
    }

    bool accept()      // (xt, ft, gt) satisfies the Wolfe conditions; returns true when the fit terminates
    {
        std::vector<double> s(D), y(D);
        double snorm2 = 0.0;
        for (int j = 0; j < D; ++j) { s[j] = xt[j] - x[j]; y[j] = gt[j] - g[j]; snorm2 += s[j] * s[j]; }
        f_prev = f; prev_dfp = dfp;
        x = xt; g = gt; f = ft;
        iters += 1;
        const double sy = dot(s, y), yy = dot(y, y);
        if (reset) {
            // the first pair after a reset scales the initial Hessian; the remembered step is rescaled with it
            const double B0fact = yy / sy;
            S.clear(); Y.clear(); RHO.clear();
            if (std::isfinite(B0fact) && B0fact > 0.0) { prev_dfp /= B0fact; alpha *= B0fact; }
        }
        if (sy > 0.0 && std::isfinite(sy)) {                // (always true under the strong Wolfe conditions)
            S.push_back(s); Y.push_back(y); RHO.push_back(1.0 / sy);
            gammak = sy / yy;
            if ((int)S.size() > opt->history) { S.pop_front(); Y.pop_front(); RHO.pop_front(); }
        }
        search_direction();                                 // direction of the next iteration; also H g for the tol_rel_grad test
        const double eps = DBL_EPSILON;
        const double df = std::fabs(f_prev - f);
        if (df < opt->tol_obj) { rc = 0; return true; }
        if (df < opt->tol_rel_obj * eps * std::max(std::fabs(f_prev), std::max(std::fabs(f), 1.0))) { rc = 0; return true; }
        if (std::sqrt(dot(g, g)) < opt->tol_grad) { rc = 0; return true; }
        if (-dot(g, p) / std::max(std::fabs(f), 1.0) < opt->tol_rel_grad * eps) { rc = 0; return true; }
        if (std::sqrt(snorm2) < opt->tol_param) { rc = 0; return true; }
        if (iters >= opt->max_iter) { rc = 1; return true; }
        return false;
    }

    void next_zoom_trial()
    {
        zoom_its += 1;
        if (std::fabs(alo - ahi) < MIN_RANGE) { linesearch_failed(); return; }
        double a;
        if (zoom_its % 5 == 0) {
            a = 0.5 * (alo + ahi);
        } else {
            const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
            double d2 = std::sqrt(d1 * d1 - aloDFp * ahiDFp);
            if (ahi < alo) d2 = -d2;
            a = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
            const double lo = std::min(alo, ahi), hi = std::max(alo, ahi), w = std::fabs(alo - ahi);
            if (!std::isfinite(a) || a < lo + 0.01 * w || a > hi - 0.01 * w) a = 0.5 * (alo + ahi);
        }
        phase = ZOOM;
        set_trial(a);
    }

    void enter_zoom(double lo, double loF, double loD, double hi, double hiF, double hiD)
    {
        alo = lo; aloF = loF; aloDFp = loD; ahi = hi; ahiF = hiF; ahiDFp = hiD;
        zoom_its = 0;
        next_zoom_trial();
    }

    // feed the evaluation at trial(): fval = -lp, grad = -grad lp
    void feed(double lp, const double *grad_lp)
    {
        n_evals += 1;
        if (phase == START) {
            f = -lp;
            for (int j = 0; j < D; ++j) g[j] = -grad_lp[j];
            if (!std::isfinite(f)) { rc = -1; phase = DONE; return; }
            if (std::sqrt(dot(g, g)) < opt->tol_grad) { rc = 0; phase = DONE; return; }
            if (opt->max_iter <= 0) { rc = 1; phase = DONE; return; }
            begin_iteration(1);
            return;
        }
        ft = -lp;
        bool finite = std::isfinite(ft);
        for (int j = 0; j < D; ++j) { gt[j] = -grad_lp[j]; finite = finite && std::isfinite(gt[j]); }
        if (phase == BRACKET) {
            if (!finite) {                                  // Stan: func() != 0 -> pull the step back towards the last good one
                if (ls_restarts >= MAX_LS_RESTARTS) { linesearch_failed(); return; }
                ls_restarts += 1;
                set_trial(0.5 * (ls_alpha0 + alpha));
                return;
            }
            ls_restarts = 0;
            const double newDFp = dot(gt, p);
            if (ft > f + alpha * c1dfp || (ft >= ls_prevF && ls_its > 0)) {
                enter_zoom(ls_alpha0, ls_prevF, ls_prevDFp, alpha, ft, newDFp);
                return;
            }
            if (std::fabs(newDFp) <= -c2dfp) {
                if (accept()) { phase = DONE; return; }
                begin_iteration(0);
                return;
            }
            if (newDFp >= 0.0) {
                enter_zoom(alpha, ft, newDFp, ls_alpha0, ls_prevF, ls_prevDFp);
                return;
            }
            ls_alpha0 = alpha; ls_prevF = ft; ls_prevDFp = newDFp;
            ls_its += 1;
            if (ls_its >= MAX_LS_ITS) { linesearch_failed(); return; }
            set_trial(alpha * 10.0);
            return;
        }
        // ZOOM
        if (!finite) {
            const double lo = std::min(alo, ahi);
            const double a = 0.5 * (alpha + lo);
            if (std::fabs(lo - a) < MIN_RANGE) { linesearch_failed(); return; }
            set_trial(a);
            return;
        }
        const double newDFp = dot(gt, p);
        if (ft > f + alpha * c1dfp || ft >= aloF) {
            ahi = alpha; ahiF = ft; ahiDFp = newDFp;
        } else {
            if (std::fabs(newDFp) <= -c2dfp) {
                if (accept()) { phase = DONE; return; }
                begin_iteration(0);
                return;
            }
            if (newDFp * (ahi - alo) >= 0.0) { ahi = alo; ahiF = aloF; ahiDFp = aloDFp; }
            alo = alpha; aloF = ft; aloDFp = newDFp;
        }
        next_zoom_trial();
    }

    // entry point for callers (kept from the first version of this file, which had a second path)
    void feed_any(double lp, const double *grad_lp) { feed(lp, grad_lp); }
};

}  // namespace bdrt
