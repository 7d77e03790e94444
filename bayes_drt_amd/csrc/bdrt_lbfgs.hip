// bdrt_lbfgs.hip -- MAP estimation: replaces StanModel.optimizing (reference bayes_drt/inversion.py:1216).
//
// L-BFGS on the unconstrained scale, log-density without Jacobian (Stan's `optimizing` semantics),
// history 5, strong-Wolfe line search (c1 = 1e-4, c2 = 0.9) with cubic interpolation, first step alpha0 = 1e-3 and
// Stan's five termination tests (tol_obj, tol_rel_obj, tol_grad, tol_rel_grad, tol_param).
// Stan's exact iterate path is not reproducible (its source is not in the reference; H1), only its contract.
//
// MI355X design: the optimiser is a per-fit state machine on the host; every log_prob+gradient it asks for is
// evaluated on the GPU, and all fits (spectra / restarts) advance in lock-step so that one kernel launch serves
// the whole batch (one 16-column MFMA tile per 16 fits).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <deque>

#include "bdrt_host.h"
#include "bdrt_lbfgs.h"

using namespace bdrt;

extern "C" {

void bdrt_opt_defaults(bdrt_opt_options *o)
{
    o->max_iter = 50000; o->history = 5; o->init_alpha = 1e-3; o->tol_obj = 1e-12; o->tol_rel_obj = 1e4;
    o->tol_grad = 1e-8; o->tol_rel_grad = 1e7; o->tol_param = 1e-8;
}

int bdrt_optimize(bdrt_problem *p, const double *init_theta, const int *spec, int n_fits, const bdrt_opt_options *opts,
                  double *theta_out, bdrt_opt_report *reports)
{
    if (!p || !init_theta || !theta_out || n_fits < 1) { set_error("bdrt_optimize: bad arguments"); return -1; }
    bdrt_opt_options o;
    if (opts) o = *opts; else bdrt_opt_defaults(&o);
    Problem &P = p->impl;
    const int D = P.dev.D;
    for (int i = 0; i < n_fits; ++i)
        if (spec && (spec[i] < 0 || spec[i] >= P.dev.n_spectra)) { set_error("bdrt_optimize: spectrum index out of range"); return -1; }
    std::vector<LbfgsFit> fits((size_t)n_fits);
    for (int i = 0; i < n_fits; ++i) fits[i].init(D, init_theta + (size_t)i * D, &o);
    int rc;
    if ((rc = P.ensure_scratch((size_t)n_fits))) return rc;
    std::vector<double> h_theta((size_t)n_fits * D), h_grad((size_t)n_fits * D), h_lp((size_t)n_fits);
    std::vector<int> h_spec((size_t)n_fits), active;
    const long long max_rounds = (long long)o.max_iter * 70 + 100;
    for (long long round = 0; round < max_rounds; ++round) {
        active.clear();
        for (int i = 0; i < n_fits; ++i)
            if (fits[i].phase != LbfgsFit::DONE) active.push_back(i);
        if (active.empty()) break;
        const int B = (int)active.size();
        for (int k = 0; k < B; ++k) {
            memcpy(&h_theta[(size_t)k * D], fits[active[k]].trial(), D * sizeof(double));
            h_spec[k] = spec ? spec[active[k]] : 0;
        }
        BDRT_HIP(hipMemcpyAsync(P.d_theta, h_theta.data(), (size_t)B * D * sizeof(double), hipMemcpyHostToDevice, P.stream));
        BDRT_HIP(hipMemcpyAsync(P.d_spec, h_spec.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, P.stream));
        if ((rc = launch_logp_grad(&P, P.d_theta, P.d_spec, B, /*jacobian=*/0, P.d_lp, P.d_grad, nullptr, nullptr,
                                   nullptr, P.stream)))
            return rc;
        BDRT_HIP(hipMemcpyAsync(h_lp.data(), P.d_lp, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, P.stream));
        BDRT_HIP(hipMemcpyAsync(h_grad.data(), P.d_grad, (size_t)B * D * sizeof(double), hipMemcpyDeviceToHost, P.stream));
        BDRT_HIP(hipStreamSynchronize(P.stream));
        for (int k = 0; k < B; ++k) {
            LbfgsFit &F = fits[active[k]];
            F.feed_any(h_lp[k], &h_grad[(size_t)k * D]);
        }
    }
    for (int i = 0; i < n_fits; ++i) {
        LbfgsFit &F = fits[i];
        memcpy(theta_out + (size_t)i * D, F.x.data(), D * sizeof(double));
        if (reports) {
            reports[i].iterations = F.iters;
            reports[i].n_evals = F.n_evals;
            reports[i].return_code = F.phase == LbfgsFit::DONE ? F.rc : 1;
            reports[i].lp = -F.f;
            reports[i].grad_norm = std::sqrt(LbfgsFit::dot(F.g, F.g));
        }
    }
    return 0;
}

}  // extern "C"
