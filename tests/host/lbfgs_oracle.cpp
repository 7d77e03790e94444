// Test harness: the product's host-side L-BFGS state machine (bayes_drt_amd/csrc/bdrt_lbfgs.h) driven by the CPU
// ORACLE's log-posterior (oracle/bdrt_oracle.c).  Used only by tests to check "same optimiser, same start:
// GPU evaluations vs oracle evaluations give the same MAP" (SURVEY H1 ladder step (b)).
// Build: g++ -O2 -std=c++17 -shared -fPIC tests/host/lbfgs_oracle.cpp -Loracle -loracle -o <out.so>
#include "../../bayes_drt_amd/csrc/bdrt_lbfgs.h"
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
