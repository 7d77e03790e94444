// CPU harness for the host-side L-BFGS state machine (bayes_drt_amd/csrc/bdrt_lbfgs.h) on analytic functions.
// Build: g++ -O2 -std=c++17 tests/host/lbfgs_harness.cpp -o /tmp/lbfgs_harness ; prints "name iters evals f |g| rc"
#include <cstdio>
#include <functional>
#include "../../bayes_drt_amd/csrc/bdrt_lbfgs.h"
using namespace bdrt;
typedef std::function<double(const double *, double *)> Fn;   // returns lp (= -f), writes grad of lp

static void run(const char *name, int D, const std::vector<double> &x0, Fn fn, int max_iter)
{
    bdrt_opt_options o;
    o.max_iter = max_iter; o.history = 5; o.init_alpha = 1e-3; o.tol_obj = 1e-12; o.tol_rel_obj = 1e4;
    o.tol_grad = 1e-8; o.tol_rel_grad = 1e7; o.tol_param = 1e-8;
    LbfgsFit F;
    F.init(D, x0.data(), &o);
    std::vector<double> g(D);
    long guard = 0;
    while (F.phase != LbfgsFit::DONE && guard++ < 10000000) {
        const double lp = fn(F.trial(), g.data());
        F.feed_any(lp, g.data());
    }
    double gn = 0; for (double v : F.g) gn += v * v;
    printf("%s %d %d %.17g %.6e %d", name, F.iters, F.n_evals, F.f, std::sqrt(gn), F.rc);
    for (int j = 0; j < D && j < 6; ++j) printf(" %.12g", F.x[j]);
    printf("\n");
}

int main()
{
    // Rosenbrock (2-D and 10-D chained)
    auto rosen = [](int D) {
        return [D](const double *x, double *g) {
            double f = 0; for (int j = 0; j < D; ++j) g[j] = 0;
            for (int j = 0; j + 1 < D; ++j) {
                const double a = x[j + 1] - x[j] * x[j], b = 1 - x[j];
                f += 100 * a * a + b * b;
                g[j] += -400 * a * x[j] - 2 * b; g[j + 1] += 200 * a;
            }
            for (int j = 0; j < D; ++j) g[j] = -g[j];
            return -f;
        };
    };
    run("rosen2", 2, {-1.2, 1.0}, rosen(2), 5000);
    run("rosen10", 10, std::vector<double>(10, -1.0), rosen(10), 5000);
    // ill-conditioned quadratic, condition 1e8
    run("quad", 20, std::vector<double>(20, 1.0), [](const double *x, double *g) {
        double f = 0; for (int j = 0; j < 20; ++j) { const double c = std::pow(10.0, 8.0 * j / 19.0); f += 0.5 * c * x[j] * x[j]; g[j] = -c * x[j]; }
        return -f; }, 5000);
    // log-barrier: -inf outside the domain (tests non-finite handling): f = x - log(x) on x>0, min at 1
    run("barrier", 1, {5.0}, [](const double *x, double *g) -> double {
        if (x[0] <= 0) { g[0] = NAN; return -(double)INFINITY; }
        g[0] = -(1 - 1 / x[0]); return -(x[0] - std::log(x[0])); }, 500);
    return 0;
}
