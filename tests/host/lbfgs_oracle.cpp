// Test harness: the product's host-side L-BFGS state machine (bayes_drt_amd/csrc/bdrt_lbfgs.h) driven by the CPU
// ORACLE's log-posterior (oracle/bdrt_oracle.c).  Used only by tests to check "same optimiser, same start:
// GPU evaluations vs oracle evaluations give the same MAP" (SURVEY H1 ladder step (b)).
// Build: g++ -O2 -std=c++17 -shared -fPIC tests/host/lbfgs_oracle.cpp -Loracle -loracle -o <out.so>
#include "../../bayes_drt_amd/csrc/bdrt_lbfgs.h"
#include "../../bayes_drt_amd/csrc/bdrt_newton.h"
#include "../../oracle/bdrt_oracle.h"

extern "C" int harness_optimize(const orc_model *m, const double *init, int max_iter, double *out, int *iters,
                                int *n_evals, double *lp_out)
{
    using namespace bdrt;
    bdrt_opt_options o;
    o.max_iter = max_iter; o.history = 5; o.init_alpha = 1e-3; o.tol_obj = 1e-12; o.tol_rel_obj = 1e4;
    o.tol_grad = 1e-8; o.tol_rel_grad = 1e7; o.tol_param = 1e-8;
    const int D = orc_num_params(m);
    LbfgsFit F;
    F.init(D, init, &o);
    std::vector<double> g(D);
    long long guard = 0;
    while (F.phase != LbfgsFit::DONE && guard++ < (long long)max_iter * 70 + 100) {
        double lp;
        orc_logp_grad(m, F.trial(), 0, &lp, g.data());
        F.feed_any(lp, g.data());
    }
    memcpy(out, F.x.data(), sizeof(double) * D);
    *iters = F.iters; *n_evals = F.n_evals; *lp_out = -F.f;
    return F.rc;
}

// L-BFGS (lbfgs_iter iterations) followed by the Newton polish, all evaluations by the CPU oracle
extern "C" int harness_optimize_newton(const orc_model *m, const double *init, int lbfgs_iter, int newton_iter, double tol,
                                       double *out, int *newton_iters, double *lp_out, double *ginf_out)
{
    using namespace bdrt;
    const int D = orc_num_params(m);
    std::vector<double> x0(D);
    int it, ne; double lp;
    if (lbfgs_iter > 0) harness_optimize(m, init, lbfgs_iter, x0.data(), &it, &ne, &lp);
    else memcpy(x0.data(), init, sizeof(double) * D);
    NewtonFit N;
    N.init(D, x0.data(), newton_iter, tol);
    std::vector<double> lps, grs;
    while (N.phase != NewtonFit::DONE) {
        const int n = N.n_requests();
        lps.resize(n); grs.resize((size_t)n * D);
        for (int i = 0; i < n; ++i) orc_logp_grad(m, N.request(i), 0, &lps[i], &grs[(size_t)i * D]);
        N.consume(lps.data(), grs.data());
    }
    memcpy(out, N.x.data(), sizeof(double) * D);
    *newton_iters = N.iters; *lp_out = N.lp; *ginf_out = N.grad_inf();
    return N.rc;
}
