// bdrt_lbfgs.h -- the L-BFGS state machine behind `algorithm='LBFGS'` (no HIP dependency in the logic: it is also compiled by
// tests/host/ on the CPU and run on analytic functions).
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
//
// ONE logic, two vector back ends: `LbfgsCore<V>` holds the scalars and every decision; `V` holds the vectors and provides
// the handful of vector operations.  `HostVecs` (std::vector; the host-driven path of bdrt_lbfgs.hip feeds it GPU evaluations,
// tests/host feeds it analytic functions) and the device back end of bdrt_lbfgs_dev.h (element j in thread j, history rows in
// LDS / HBM, dot products as workgroup reductions) -- so the device-resident optimiser takes the decisions this file takes.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../include/bdrt.h"

#if defined(__HIPCC__)
// (inlined into the device kernel: a call would force the whole state machine into scratch memory)
#define BDRT_HD __host__ __device__ __forceinline__
#else
#define BDRT_HD
#endif

namespace bdrt {

using std::isfinite;

constexpr int LBFGS_MAX_HISTORY = 8;

// Vector back end contract (all methods uniform across the threads that share a fit on the device):
//   double dot_g_p(), dot_g_g();           g.p, g.g at the current iterate
//   double dot_gt_p(bool &finite);         gt.p at the trial point; finite = every component of gt finite
//   void p_minus_g();                      p = -g
//   void set_trial(double a);              xt = x + a p
//   void accept(int slot, double &sy, double &yy, double &ss);   S[slot] = xt - x, Y[slot] = gt - g, then x = xt, g = gt
//   void q_from_g(); double dot_S_q(int slot), dot_Y_q(int slot); void q_axpy_Y(int slot, double c), q_axpy_S(int slot, double c);
//   void q_scale(double c); void p_minus_q();
template <class V>
struct LbfgsCore {
    enum Phase { START, BRACKET, ZOOM, DONE };
    // return codes: 0 a convergence test fired, 1 iteration cap, -1 no finite start, -2 line search failed twice (TERM_LSFAIL)
    static constexpr double C1 = 1e-4, C2 = 0.9, MIN_ALPHA = 1e-12, MIN_RANGE = 1e-16;
    static constexpr int MAX_LS_ITS = 20, MAX_LS_RESTARTS = 10;

    V v;
    bdrt_opt_options opt;
    Phase phase = START;
    int iters = 0, n_evals = 0, rc = 1;
    int ls_fail_resets = 0;           // how often a failed line search reset the history (diagnostic)
    double f = 0.0, ft = 0.0;
    // history: circular buffer of `hist_n` pairs, newest at slot (hist_head + hist_n - 1) % history
    int hist_head = 0, hist_n = 0;
    double rho[LBFGS_MAX_HISTORY];
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

    BDRT_HD int history() const { return opt.history < 1 ? 1 : (opt.history > LBFGS_MAX_HISTORY ? LBFGS_MAX_HISTORY : opt.history); }
    BDRT_HD int slot(int i) const { return (hist_head + i) % history(); }          // i = 0 oldest .. hist_n - 1 newest

    // rho[] by a run-time slot through compile-time indices only: a dynamically indexed member would pin the whole state
    // machine in scratch memory on the device
    BDRT_HD double rho_get(int sl) const
    {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < LBFGS_MAX_HISTORY; ++k) r = k == sl ? rho[k] : r;
        return r;
    }
    BDRT_HD void rho_set(int sl, double val)
    {
#pragma unroll
        for (int k = 0; k < LBFGS_MAX_HISTORY; ++k) rho[k] = k == sl ? val : rho[k];
    }

    BDRT_HD void search_direction()     // p = -H g (two-loop recursion, initial scaling gammak)
    {
        double al[LBFGS_MAX_HISTORY];
        v.q_from_g();
#pragma unroll
        for (int i = LBFGS_MAX_HISTORY - 1; i >= 0; --i) {
            al[i] = 0.0;
            if (i < hist_n) {
                al[i] = rho_get(slot(i)) * v.dot_S_q(slot(i));
                v.q_axpy_Y(slot(i), -al[i]);
            }
        }
        v.q_scale(gammak);
#pragma unroll
        for (int i = 0; i < LBFGS_MAX_HISTORY; ++i) {
            if (i < hist_n) {
                const double be = rho_get(slot(i)) * v.dot_Y_q(slot(i));
                v.q_axpy_S(slot(i), al[i] - be);
            }
        }
        v.p_minus_q();
    }

    // Minimiser over [loX, hiX] of the cubic c(t) with c(0) = 0, c'(0) = df0, c(x1) = f1, c'(x1) = df1 (Stan's CubicInterp)
    BDRT_HD static double cubic_interp(double df0, double x1, double f1, double df1, double loX, double hiX)
    {
        const double c3 = (-12.0 * f1 + 6.0 * x1 * (df0 + df1)) / (x1 * x1 * x1);
        const double c2 = -(4.0 * df0 + 2.0 * df1) / x1 + 6.0 * f1 / (x1 * x1);
        const double c1 = df0;
        const double t_s = sqrt(c2 * c2 - 2.0 * c1 * c3);
        const double s1 = -(c2 + t_s) / c3, s2 = -(c2 - t_s) / c3;
        double minF = loX * (loX * (loX * c3 / 3.0 + c2) / 2.0 + c1), minX = loX;
        double tmp = hiX * (hiX * (hiX * c3 / 3.0 + c2) / 2.0 + c1);
        if (tmp < minF) { minF = tmp; minX = hiX; }
        if (loX < s1 && s1 < hiX) { tmp = s1 * (s1 * (s1 * c3 / 3.0 + c2) / 2.0 + c1); if (tmp < minF) { minF = tmp; minX = s1; } }
        if (loX < s2 && s2 < hiX) { tmp = s2 * (s2 * (s2 * c3 / 3.0 + c2) / 2.0 + c1); if (tmp < minF) { minF = tmp; minX = s2; } }
        return minX;
    }

    BDRT_HD void set_trial(double a) { alpha = a; v.set_trial(a); }

    // one iteration of Stan's step(): choose the direction / first trial step and enter the line search
    BDRT_HD void begin_iteration(int reset_code)
    {
        reset = reset_code;
        if (reset) v.p_minus_g();
        dfp = v.dot_g_p();
        double a0;
        if (iters > 0 && reset != 2) {
            const double ci = cubic_interp(prev_dfp, alpha, f - f_prev, dfp, MIN_ALPHA, 1.0);
            a0 = 1.01 * ci < 1.0 ? 1.01 * ci : 1.0;
            if (!(a0 > 0.0) || !isfinite(a0)) a0 = opt.init_alpha;      // (a NaN cubic: Stan would fail the search and reset)
        } else {
            a0 = opt.init_alpha;
        }
        c1dfp = C1 * dfp; c2dfp = C2 * dfp;
        ls_alpha0 = MIN_ALPHA; ls_prevF = f; ls_prevDFp = dfp;
        ls_its = 0; ls_restarts = 0;
        phase = BRACKET;
        set_trial(a0);
    }

    BDRT_HD void linesearch_failed()
    {
        if (reset) { rc = -2; phase = DONE; return; }       // already from steepest descent: nothing else to try
        hist_n = 0; hist_head = 0; gammak = 1.0;
        ls_fail_resets += 1;
        begin_iteration(2);
    }

    BDRT_HD bool accept()      // the trial point satisfies the Wolfe conditions; returns true when the fit terminates
    {
        const int H = history();
        if (reset) { hist_n = 0; hist_head = 0; }
        int sl;
        if (hist_n < H) sl = slot(hist_n);
        else { sl = hist_head; hist_head = (hist_head + 1) % H; hist_n = H - 1; }      // the new pair overwrites the oldest one
        double sy, yy, ss;
        v.accept(sl, sy, yy, ss);
        f_prev = f; prev_dfp = dfp; f = ft;
        iters += 1;
        if (reset) {
            // the first pair after a reset scales the initial Hessian; the remembered step is rescaled with it
            const double B0fact = yy / sy;
            if (isfinite(B0fact) && B0fact > 0.0) { prev_dfp /= B0fact; alpha *= B0fact; }
        }
        if (sy > 0.0 && isfinite(sy)) {                     // (always true under the strong Wolfe conditions)
            rho_set(sl, 1.0 / sy);
            gammak = sy / yy;
            hist_n += 1;
        }
        search_direction();                                 // direction of the next iteration; also H g for the tol_rel_grad test
        const double eps = DBL_EPSILON;
        const double df = fabs(f_prev - f);
        const double fm = fabs(f_prev) > fabs(f) ? fabs(f_prev) : fabs(f);
        if (df < opt.tol_obj) { rc = 0; return true; }
        if (df < opt.tol_rel_obj * eps * (fm > 1.0 ? fm : 1.0)) { rc = 0; return true; }
        if (sqrt(v.dot_g_g()) < opt.tol_grad) { rc = 0; return true; }
        if (-v.dot_g_p() / (fabs(f) > 1.0 ? fabs(f) : 1.0) < opt.tol_rel_grad * eps) { rc = 0; return true; }
        if (sqrt(ss) < opt.tol_param) { rc = 0; return true; }
        if (iters >= opt.max_iter) { rc = 1; return true; }
        return false;
    }

    BDRT_HD void next_zoom_trial()
    {
        zoom_its += 1;
        if (fabs(alo - ahi) < MIN_RANGE) { linesearch_failed(); return; }
        double a;
        if (zoom_its % 5 == 0) {
            a = 0.5 * (alo + ahi);
        } else {
            const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
            double d2 = sqrt(d1 * d1 - aloDFp * ahiDFp);
            if (ahi < alo) d2 = -d2;
            a = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
            const double lo = alo < ahi ? alo : ahi, hi = alo < ahi ? ahi : alo, w = fabs(alo - ahi);
            if (!isfinite(a) || a < lo + 0.01 * w || a > hi - 0.01 * w) a = 0.5 * (alo + ahi);
        }
        phase = ZOOM;
        set_trial(a);
    }

    BDRT_HD void enter_zoom(double lo, double loF, double loD, double hi, double hiF, double hiD)
    {
        alo = lo; aloF = loF; aloDFp = loD; ahi = hi; ahiF = hiF; ahiDFp = hiD;
        zoom_its = 0;
        next_zoom_trial();
    }

    // the start point has been evaluated (gradient in the back end): fval = -lp
    BDRT_HD void start(double fval)
    {
        n_evals += 1;
        f = fval;
        if (!isfinite(f)) { rc = -1; phase = DONE; return; }
        if (sqrt(v.dot_g_g()) < opt.tol_grad) { rc = 0; phase = DONE; return; }
        if (opt.max_iter <= 0) { rc = 1; phase = DONE; return; }
        begin_iteration(1);
    }

    // the trial point has been evaluated (its gradient in the back end): fval = -lp
    BDRT_HD void feed_trial(double fval)
    {
        n_evals += 1;
        ft = fval;
        bool gfinite = true;
        const double newDFp = v.dot_gt_p(gfinite);
        const bool finite = isfinite(ft) && gfinite;
        if (phase == BRACKET) {
            if (!finite) {                                  // Stan: func() != 0 -> pull the step back towards the last good one
                if (ls_restarts >= MAX_LS_RESTARTS) { linesearch_failed(); return; }
                ls_restarts += 1;
                set_trial(0.5 * (ls_alpha0 + alpha));
                return;
            }
            ls_restarts = 0;
            if (ft > f + alpha * c1dfp || (ft >= ls_prevF && ls_its > 0)) {
                enter_zoom(ls_alpha0, ls_prevF, ls_prevDFp, alpha, ft, newDFp);
                return;
            }
            if (fabs(newDFp) <= -c2dfp) {
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
            const double lo = alo < ahi ? alo : ahi;
            const double a = 0.5 * (alpha + lo);
            if (fabs(lo - a) < MIN_RANGE) { linesearch_failed(); return; }
            set_trial(a);
            return;
        }
        if (ft > f + alpha * c1dfp || ft >= aloF) {
            ahi = alpha; ahiF = ft; ahiDFp = newDFp;
        } else {
            if (fabs(newDFp) <= -c2dfp) {
                if (accept()) { phase = DONE; return; }
                begin_iteration(0);
                return;
            }
            if (newDFp * (ahi - alo) >= 0.0) { ahi = alo; ahiF = aloF; ahiDFp = aloDFp; }
            alo = alpha; aloF = ft; aloDFp = newDFp;
        }
        next_zoom_trial();
    }
};

// ---- host back end --------------------------------------------------------------------------------------------------------
struct HostVecs {
    int D = 0;
    std::vector<double> x, g, xt, gt, p, q;
    std::vector<std::vector<double>> S, Y;
    static double dot(const std::vector<double> &a, const std::vector<double> &b)
    {
        double s = 0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }
    double dot_g_p() { return dot(g, p); }
    double dot_g_g() { return dot(g, g); }
    double dot_gt_p(bool &finite)
    {
        finite = true;
        for (double t : gt) finite = finite && std::isfinite(t);
        return finite ? dot(gt, p) : 0.0;
    }
    void p_minus_g() { for (int j = 0; j < D; ++j) p[j] = -g[j]; }
    void set_trial(double a) { for (int j = 0; j < D; ++j) xt[j] = x[j] + a * p[j]; }
    void accept(int slot, double &sy, double &yy, double &ss)
    {
        std::vector<double> &s = S[slot], &y = Y[slot];
        for (int j = 0; j < D; ++j) { s[j] = xt[j] - x[j]; y[j] = gt[j] - g[j]; }
        sy = dot(s, y); yy = dot(y, y); ss = dot(s, s);
        x = xt; g = gt;
    }
    void q_from_g() { q = g; }
    double dot_S_q(int slot) { return dot(S[slot], q); }
    double dot_Y_q(int slot) { return dot(Y[slot], q); }
    void q_axpy_Y(int slot, double c) { for (int j = 0; j < D; ++j) q[j] += c * Y[slot][j]; }
    void q_axpy_S(int slot, double c) { for (int j = 0; j < D; ++j) q[j] += c * S[slot][j]; }
    void q_scale(double c) { for (int j = 0; j < D; ++j) q[j] *= c; }
    void p_minus_q() { for (int j = 0; j < D; ++j) p[j] = -q[j]; }
};

// the host-driven fit: asks for one evaluation at a time (`trial()`), is fed lp and the gradient of lp
struct LbfgsFit {
    typedef LbfgsCore<HostVecs> Core;
    enum Phase { START = Core::START, BRACKET = Core::BRACKET, ZOOM = Core::ZOOM, DONE = Core::DONE };
    Core c;
    int D = 0;
    // views the callers read
    std::vector<double> &x = c.v.x, &g = c.v.g;
    double &f = c.f;
    int &iters = c.iters, &n_evals = c.n_evals, &rc = c.rc;
    int phase = START;

    static double dot(const std::vector<double> &a, const std::vector<double> &b) { return HostVecs::dot(a, b); }

    void init(int D_, const double *x0, const bdrt_opt_options *o)
    {
        D = D_;
        c.opt = *o;
        HostVecs &v = c.v;
        v.D = D;
        v.x.assign(x0, x0 + D); v.g.assign(D, 0.0); v.xt = v.x; v.gt = v.g; v.p = v.g; v.q = v.g;
        v.S.assign(LBFGS_MAX_HISTORY, std::vector<double>(D, 0.0));
        v.Y = v.S;
        c.phase = Core::START;
        phase = START;
    }
    const double *trial() const { return c.phase == Core::START ? c.v.x.data() : c.v.xt.data(); }

    // feed the evaluation at trial(): lp and the gradient of lp
    void feed_any(double lp, const double *grad_lp)
    {
        if (c.phase == Core::START) {
            for (int j = 0; j < D; ++j) c.v.g[j] = -grad_lp[j];
            c.start(-lp);
        } else {
            for (int j = 0; j < D; ++j) c.v.gt[j] = -grad_lp[j];
            c.feed_trial(-lp);
        }
        phase = (int)c.phase;
    }
    // result of a fit that ran elsewhere (the device-resident kernel): last iterate, gradient of -lp there, counters
    void load(const double *x_, const double *g_, double f_, int iters_, int n_evals_, int rc_)
    {
        c.v.x.assign(x_, x_ + D); c.v.g.assign(g_, g_ + D);
        c.f = f_; c.iters = iters_; c.n_evals = n_evals_; c.rc = rc_;
        c.phase = Core::DONE; phase = DONE;
    }
    LbfgsFit() = default;
    LbfgsFit(const LbfgsFit &o) : c(o.c), D(o.D), x(c.v.x), g(c.v.g), f(c.f), iters(c.iters), n_evals(c.n_evals), rc(c.rc), phase(o.phase) {}
    LbfgsFit &operator=(const LbfgsFit &o) { c = o.c; D = o.D; phase = o.phase; return *this; }
};

}  // namespace bdrt
