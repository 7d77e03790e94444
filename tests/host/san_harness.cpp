// Sanitizer harness (CPU): a small synthetic Series_pos problem through the oracle's matrix builders, log-posterior,
// NUTS, and through the product's host-side L-BFGS / Newton / QP-free state machines driven by the oracle.
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../bayes_drt_amd/csrc/bdrt_lbfgs.h"
#include "../../bayes_drt_amd/csrc/bdrt_newton.h"
extern "C" {
#include "../../oracle/bdrt_oracle.h"
typedef struct { double adapt_delta, adapt_t0, adapt_gamma, adapt_kappa; int max_treedepth, init_buffer, term_buffer, base_window;
                 double init_radius, max_deltaH, stepsize0; } orc_nuts_control;
typedef struct { long long n_leapfrog; int n_divergent, n_max_treedepth; double stepsize, mean_accept; } orc_chain_diag;
void orc_nuts_defaults(orc_nuts_control *c);
int orc_nuts_sample(const orc_model *m, int chain_id, unsigned long long seed, int warmup, int n_draws, const double *init_theta,
                    const orc_nuts_control *ctrl, double *draws, double *lp_draws, orc_chain_diag *diag);
}
using namespace bdrt;

int main()
{
    const int nf = 14, K = 11;
    std::vector<double> f(nf), tau(K);
    for (int i = 0; i < nf; ++i) f[i] = std::pow(10.0, 4.0 - 0.5 * i);
    for (int k = 0; k < K; ++k) tau[k] = std::pow(10.0, -5.0 + 0.6 * k);
    const double eps = 1.0 / (0.6 * std::log(10.0));
    std::vector<double> A(2 * nf * K), L0(K * K), L1(K * K), L2(K * K), Z(2 * nf);
    orc_build_A(f.data(), nf, tau.data(), K, eps, ORC_KERNEL_DRT, 0, 1, 0, 0.0, 0, A.data());
    orc_build_A(f.data(), nf, tau.data(), K, eps, ORC_KERNEL_DRT, 1, 1, 0, 0.0, 0, A.data() + nf * K);
    const double c0[4] = {1, 0, 0, 0}, c1[4] = {0, 1, 0, 0}, c2[4] = {0, 0, 0.75, 0};
    orc_build_L(tau.data(), K, eps, c0, L0.data());
    orc_build_L(tau.data(), K, eps, c1, L1.data());
    orc_build_L(tau.data(), K, eps, c2, L2.data());
    for (int i = 0; i < 2 * nf; ++i) { Z[i] = (i < nf ? 1.0 : 0.0); for (int k = 0; k < K; ++k) Z[i] += 0.2 * A[i * K + k]; }
    orc_model m; memset(&m, 0, sizeof(m));
    m.nf = nf; m.nblocks = 1; m.K[0] = K; m.nonneg[0] = 1; m.x_scale[0] = 1.0;
    m.A[0] = A.data(); m.L0[0] = L0.data(); m.L1[0] = L1.data(); m.L2[0] = L2.data(); m.Z = Z.data(); m.freq = f.data();
    m.sigma_min = 0.002; m.ups_alpha = 1.0; m.ups_beta = 0.1; m.induc_scale = 1.0;
    const int D = orc_num_params(&m);
    std::vector<double> th(D, -0.5), g(D), draws(6 * D), lps(6);
    double lp;
    orc_logp_grad(&m, th.data(), 1, &lp, g.data());
    if (!std::isfinite(lp)) return 2;
    orc_nuts_control c; orc_nuts_defaults(&c); c.max_treedepth = 5;
    orc_chain_diag dg;
    if (orc_nuts_sample(&m, 0, 42ull, 12, 6, nullptr, &c, draws.data(), lps.data(), &dg) != 0) return 3;
    bdrt_opt_options o; o.max_iter = 150; o.history = 5; o.init_alpha = 1e-3; o.tol_obj = 1e-12; o.tol_rel_obj = 1e4;
    o.tol_grad = 1e-8; o.tol_rel_grad = 1e7; o.tol_param = 1e-8; o.newton_max_iter = 30; o.lbfgs_before_newton = 150; o.newton_tol = 1e-8;
    LbfgsFit F; F.init(D, th.data(), &o);
    long guard = 0;
    while (F.phase != LbfgsFit::DONE && guard++ < 100000) { orc_logp_grad(&m, F.trial(), 0, &lp, g.data()); F.feed_any(lp, g.data()); }
    NewtonFit N; N.init(D, F.x.data(), 30, 1e-8);
    std::vector<double> lpv, gv;
    while (N.phase != NewtonFit::DONE) {
        const int n = N.n_requests();
        lpv.resize(n); gv.resize((size_t)n * D);
        for (int i = 0; i < n; ++i) orc_logp_grad(&m, N.request(i), 0, &lpv[i], &gv[(size_t)i * D]);
        N.consume(lpv.data(), gv.data());
    }
    printf("SAN_OK lp=%.6f newton_iters=%d ginf=%.2e leapfrogs=%lld\n", N.lp, N.iters, N.grad_inf(), dg.n_leapfrog);
    return 0;
}
