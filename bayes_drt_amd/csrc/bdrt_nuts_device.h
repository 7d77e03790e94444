// bdrt_nuts_device.h -- counter-based RNG and the per-chain NUTS bookkeeping shared by the NUTS kernel.
//
// The sampler being replaced is Stan 2.19.1's adapt_diag_e_nuts as driven by pystan's `sampling`
// (reference call site bayes_drt/inversion.py:1218-1221; behaviour: SURVEY.md Appendix A).  Stan's own RNG
// stream cannot be reproduced (boost::ecuyer1988 inside the absent pystan), so randomness is a counter-based
// Philox4x32-10 keyed by (seed, chain id) and indexed by (iteration, tree depth, leaf, purpose): every draw is
// a pure function of those, which makes results independent of how chains are packed into workgroups / GPUs.
#pragma once
#include <stdint.h>

namespace bdrt {

// purposes (counter word 1, low byte)
enum { RNG_INIT = 1, RNG_MOMENTUM = 2, RNG_DIRECTION = 3, RNG_LEAF = 4, RNG_TOP = 5, RNG_EPS_MOMENTUM = 6 };

struct Philox {
    uint32_t k0, k1, chain;
};

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two uniforms in (0,1) from one Philox block; counter = (index, purpose | depth<<8 | trial<<16, iteration, chain)
__host__ __device__ inline void rng_uniform2(const Philox &g, uint32_t index, uint32_t purpose, uint32_t depth,
                                             uint32_t trial, uint32_t iter, double &u0, double &u1)
{
    uint32_t o[4];
    philox4x32_10(index, purpose | (depth << 8) | (trial << 16), iter, g.chain, g.k0, g.k1, o);
    const uint64_t a = ((uint64_t)o[0] << 32) | o[1], b = ((uint64_t)o[2] << 32) | o[3];
    u0 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    u1 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline double rng_uniform(const Philox &g, uint32_t index, uint32_t purpose, uint32_t depth,
                                              uint32_t trial, uint32_t iter)
{
    double u0, u1;
    rng_uniform2(g, index, purpose, depth, trial, iter, u0, u1);
    return u0;
}

// standard normal number `j` of a stream: Box-Muller on block j/2, cosine branch for even j, sine for odd j
__host__ __device__ inline double rng_normal(const Philox &g, uint32_t j, uint32_t purpose, uint32_t trial,
                                             uint32_t iter)
{
    double u0, u1;
    rng_uniform2(g, j >> 1, purpose, 0, trial, iter, u0, u1);
    const double r = sqrt(-2.0 * log(u0));
    const double a = 6.283185307179586476925286766559 * u1;
    return (j & 1) ? r * sin(a) : r * cos(a);
}

#if defined(__HIPCC__)
// normals 2i (cosine branch) and 2i+1 (sine branch) of a stream from ONE Philox block and one Box-Muller transform;
// same values as rng_normal(2i), rng_normal(2i+1) up to the rounding of sin/cos(2 pi u) vs sinpi/cospi(2 u)
__device__ inline void rng_normal_pair(const Philox &g, uint32_t i, uint32_t purpose, uint32_t trial, uint32_t iter,
                                       double &z0, double &z1)
{
    double u0, u1;
    rng_uniform2(g, i, purpose, 0, trial, iter, u0, u1);
    const double r = sqrt(-2.0 * log(u0));
    double sn, cs;
    sincospi(2.0 * u1, &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
}
#endif

// The per-leaf weights of the device kernels go through the lean exp / log of bdrt_device.h (their arguments are finite and <= 0, or the
// comparison they feed is false either way); host code -- the window logic's unit test -- keeps the C library.
#if defined(__HIP_DEVICE_COMPILE__)
#define BDRT_NUTS_EXP(x) ::bdrt::lean_exp(x)
#define BDRT_NUTS_LOG1P_EXP(d) ::bdrt::lean_log(1.0 + ::bdrt::lean_exp(d))
#else
#define BDRT_NUTS_EXP(x) exp(x)
#define BDRT_NUTS_LOG1P_EXP(d) log1p(exp(d))
#endif

__host__ __device__ inline double log_sum_exp2(double a, double b)
{
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    return a > b ? a + BDRT_NUTS_LOG1P_EXP(b - a) : b + BDRT_NUTS_LOG1P_EXP(a - b);
}

// A leaf joins a subtree of log-weight lsw_sub with log-weight w: the subtree's new log-weight (log_sum_exp2(lsw_sub, w): its own formula)
// and whether the leaf replaces the subtree's proposal, u < exp(w - lsw_new) (Stan base_nuts::build_tree: multinomial sampling inside
// the subtree).  Device code takes both from ONE exponential, t = exp(-|lsw_sub - w|): lsw_new = max + log(1 + t), and
// exp(w - lsw_new) = (w >= lsw_sub ? 1 : t) / (1 + t) -- the second exponential and the logarithm are off the path to the decision
// (three dependent transcendentals before; the decision can differ from the other form's only where u meets the probability to the
// last bit).  Host code (the logic's unit tests) keeps the textbook form.
__host__ __device__ inline bool nuts_leaf_joins(double lsw_sub, double w, double u, double &lsw_new)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double dd = lsw_sub - w;
    const double t = BDRT_NUTS_EXP(-fabs(dd));
    lsw_new = lsw_sub == -INFINITY ? w : fmax(lsw_sub, w) + ::bdrt::lean_log(1.0 + t);
    // (an empty subtree -- lsw_sub = -inf, the lean exponential's argument is not finite -- takes its first leaf whatever u is)
    return lsw_sub == -INFINITY || u * (1.0 + t) < (dd <= 0.0 ? 1.0 : t);
#else
    lsw_new = log_sum_exp2(lsw_sub, w);
    return u < BDRT_NUTS_EXP(w - lsw_new);
#endif
}

// chain phases
enum { PH_INIT = 0, PH_EPS = 1, PH_TREE = 2, PH_DONE = 3, PH_FAILED = 4 };

// per-chain scalar state, one struct per chain in global memory (host reads it back for diagnostics)
struct ChainState {
    int phase;
    int iter;             // completed iterations (warm-up + sampling)
    int depth;            // current tree depth (doublings completed)
    int leaf, nleaves;    // position inside the subtree being built
    int dir;              // +1 / -1
    int init_attempt;
    int eps_dir, eps_trials;
    int n_leap_iter;      // leapfrogs in the current iteration
    int da_counter;
    int win_counter, win_size, next_window, win_n;
    int init_buffer, term_buffer, base_window;
    int n_div, n_maxdepth, n_post;
    int spec, chain_id;
    int kicked;           // the half kick + drift of the upcoming evaluation has already been applied
    int z_iter;           // iteration whose momentum normals are already stored in the chain's V_ZN row (-1: none)
    int thsel;            // which of the two theta rows (V_TH / V_TH2) is live (wide-vector path: speculative drift, see stage C)
    double eps;           // nominal step size
    double H0, lsw, lsw_sub, lps, lpq, sum_metro;
    double da_sbar, da_xbar, da_mu;
    double sum_accept;
    long long n_leap_total;
};

// member-by-member copy: `a = b` on this 200-byte struct between LDS / global memory and a local is lowered to a block copy through a
// private (scratch) buffer that the optimiser does not split afterwards -- the 112 bytes of scratch every one-chain-per-wave kernel
// carried in round 5.  (Template: source and destination keep their address spaces.)
template <class DST, class SRC>
__host__ __device__ inline void chain_state_copy(DST &d, const SRC &s)
{
    d.phase = s.phase; d.iter = s.iter; d.depth = s.depth; d.leaf = s.leaf; d.nleaves = s.nleaves; d.dir = s.dir;
    d.init_attempt = s.init_attempt; d.eps_dir = s.eps_dir; d.eps_trials = s.eps_trials; d.n_leap_iter = s.n_leap_iter;
    d.da_counter = s.da_counter; d.win_counter = s.win_counter; d.win_size = s.win_size; d.next_window = s.next_window; d.win_n = s.win_n;
    d.init_buffer = s.init_buffer; d.term_buffer = s.term_buffer; d.base_window = s.base_window;
    d.n_div = s.n_div; d.n_maxdepth = s.n_maxdepth; d.n_post = s.n_post; d.spec = s.spec; d.chain_id = s.chain_id;
    d.kicked = s.kicked; d.z_iter = s.z_iter; d.thsel = s.thsel;
    d.eps = s.eps; d.H0 = s.H0; d.lsw = s.lsw; d.lsw_sub = s.lsw_sub; d.lps = s.lps; d.lpq = s.lpq; d.sum_metro = s.sum_metro;
    d.da_sbar = s.da_sbar; d.da_xbar = s.da_xbar; d.da_mu = s.da_mu; d.sum_accept = s.sum_accept; d.n_leap_total = s.n_leap_total;
}

struct NutsParams {
    int warmup, n_draws, max_depth;
    double delta, gamma, t0, kappa, init_radius, max_deltaH, stepsize0;
    uint32_t seed_lo, seed_hi;
    int has_init;
};

// Stan's windowed_adaptation schedule (stan/mcmc/windowed_adaptation.hpp)
__host__ __device__ inline void window_init(ChainState &s, int warmup, int init_buffer, int term_buffer, int base_window)
{
    const bool no_metric = warmup < 20;      // Stan: set_window_params returns early, the (unsigned) next window is never reached
    if (no_metric) {
        init_buffer = warmup; term_buffer = 0; base_window = 0;
    } else if (init_buffer + base_window + term_buffer > warmup) {
        init_buffer = (int)(0.15 * warmup);
        term_buffer = (int)(0.1 * warmup);
        base_window = warmup - (init_buffer + term_buffer);
    }
    s.init_buffer = init_buffer; s.term_buffer = term_buffer; s.base_window = base_window;
    s.win_counter = 0;
    s.win_size = base_window;
    s.next_window = no_metric ? -1 : init_buffer + base_window - 1;
    s.win_n = 0;
}
template <class S>
__host__ __device__ inline bool window_active(const S &s, int warmup)
{
    return s.win_counter >= s.init_buffer && s.win_counter < warmup - s.term_buffer && s.win_counter != warmup;
}
template <class S>
__host__ __device__ inline bool window_end(const S &s, int warmup)
{
    return s.win_counter == s.next_window && s.win_counter != warmup;
}
template <class S>
__host__ __device__ inline void window_next(S &s, int warmup)
{
    if (s.next_window == warmup - s.term_buffer - 1) return;
    s.win_size *= 2;
    s.next_window = s.win_counter + s.win_size;
    if (s.next_window == warmup - s.term_buffer - 1) return;
    const int boundary = s.next_window + 2 * s.win_size;
    if (boundary >= warmup - s.term_buffer) s.next_window = warmup - s.term_buffer - 1;
}

// Stan's stepsize_adaptation::learn_stepsize (dual averaging)
template <class S>
__host__ __device__ inline void da_restart(S &s) { s.da_counter = 0; s.da_sbar = 0.0; s.da_xbar = 0.0; }
template <class S>
__host__ __device__ inline void da_learn(S &s, const NutsParams &np, double accept)
{
    s.da_counter += 1;
    accept = accept > 1.0 ? 1.0 : accept;
    const double eta = 1.0 / (s.da_counter + np.t0);
    s.da_sbar = (1.0 - eta) * s.da_sbar + eta * (np.delta - accept);
    const double x = s.da_mu - s.da_sbar * sqrt((double)s.da_counter) / np.gamma;
    const double x_eta = pow((double)s.da_counter, -np.kappa);
    s.da_xbar = (1.0 - x_eta) * s.da_xbar + x_eta * x;
    s.eps = exp(x);
}


// ---- scalar logic shared by the three samplers (16-chain kernel, one-chain-per-workgroup kernel, cooperative wide-vector
//      path): the same statements on a chain state that lives in LDS or in registers ---------------------------------------

// Stan base_hmc::init_stepsize, one trial: lp, kin = log-posterior / kinetic energy after the trial leapfrog.
// Returns the kind of next start point: 1 new transition (search finished), 3 another trial.
template <class S>
__host__ __device__ inline int nuts_stepsize_trial(S &s, const NutsParams &np, double lp, double kin)
{
    double h = -lp + kin;
    if (isnan(h)) h = INFINITY;
    const double dH = s.H0 - h;
    const double thr = -0.2231435513142097557662950903;   // log(0.8)
    bool finished = false;
    const int trials = s.eps_trials;
    const int edir = s.eps_dir;
    double eps = s.eps;
    if (trials == 0) {
        s.eps_dir = dH > thr ? 1 : -1;
    } else {
        if (edir == 1 && !(dH > thr)) finished = true;
        else if (edir == -1 && !(dH < thr)) finished = true;
        else eps = edir == 1 ? 2.0 * eps : 0.5 * eps;
        if (!(eps > 1e-300) || eps > 1e7) finished = true;   // Stan throws here; we stop adapting
        s.eps = eps;
    }
    s.eps_trials = trials + 1;
    if (!finished) return 3;
    // services::sample::hmc_nuts_diag_e_adapt sets mu = log(10*stepsize) from the CONFIGURED step size before the first
    // init_stepsize; after a metric update mu = log(10*eps) (adapt_diag_e_nuts)
    s.da_mu = s.iter == 0 ? log(10.0 * np.stepsize0) : log(10.0 * eps);
    da_restart(s);
    s.phase = PH_TREE;
    return 1;
}

// One new leaf of the subtree under construction (Stan base_nuts::build_tree at depth 0, seen in time order): energy error,
// divergence test, acceptance statistic, and the uniform sampling inside the subtree (keep leaf i with probability
// w_i / W_i).  Outputs: endt = 2 on divergence; otherwise tree = true, copyq (this leaf becomes the subtree's proposal),
// nm (trailing one bits of the leaf index = sub-subtrees that end here), last (the subtree is complete).
// (u_pre: the leaf's uniform when the caller has drawn it ahead -- rng_uniform(rng, leaf_now, RNG_LEAF, depth, 0, iter) --, else nullptr)
template <class S>
__host__ __device__ inline void nuts_tree_leaf(S &s, const NutsParams &np, const Philox &rng, double lp, double kin, int leaf_now,
                                               bool &copyq, bool &tree, int &nm, bool &last, int &endt, const double *u_pre = nullptr)
{
    s.n_leap_iter = s.n_leap_iter + 1;
    double h = -lp + kin;
    if (isnan(h)) h = INFINITY;
    const double H0 = s.H0;
    const bool divergent = (h - H0) > np.max_deltaH;
    const double w = H0 - h;
    s.sum_metro = s.sum_metro + (w > 0.0 ? 1.0 : BDRT_NUTS_EXP(fmax(w, -746.0)));
    if (divergent) {
        endt = 2;               // transition ends, subtree discarded, divergent
        return;
    }
    const double u = u_pre ? *u_pre : rng_uniform(rng, (uint32_t)leaf_now, RNG_LEAF, (uint32_t)s.depth, 0, (uint32_t)s.iter);
    double lsw_new;
    const bool joins = nuts_leaf_joins(s.lsw_sub, w, u, lsw_new);
    if (leaf_now == 0 || joins) { copyq = true; s.lpq = lp; }
    s.lsw_sub = lsw_new;
    tree = true;
    while ((leaf_now >> nm) & 1) ++nm;              // trailing ones = sub-subtrees ending here
    last = leaf_now == s.nleaves - 1;
}

// End of a transition (Stan adapt_diag_e_nuts::transition): statistics, dual averaging, metric-window bookkeeping.
// endt: 1 U-turn / depth limit, 2 divergence.  Outputs: draw (index of the sampling draw to store, -1 in warm-up),
// welf / wend (Welford update / end of a metric window, with wn samples), and the kind of next start point (return value:
// 0 chain finished, 1 new transition, 3 step-size search after a metric update).
template <class S>
__host__ __device__ inline int nuts_transition_end(S &s, const NutsParams &np, int endt, int &draw, bool &welf, bool &wend, double &wn)
{
    const int nli = s.n_leap_iter;
    const double accept = s.sum_metro / (double)(nli > 0 ? nli : 1);
    const int iter = s.iter;
    const bool warm = iter < np.warmup;
    s.n_leap_total = s.n_leap_total + nli;
    if (!warm) {
        s.n_post = s.n_post + 1;
        s.sum_accept = s.sum_accept + accept;
        if (endt == 2) s.n_div = s.n_div + 1;
        if (s.depth >= np.max_depth) s.n_maxdepth = s.n_maxdepth + 1;
        draw = iter - np.warmup;
    }
    bool redo_eps = false;
    if (warm) {
        da_learn(s, np, accept);                                 // stepsize_adaptation::learn_stepsize (dual averaging)
        // var_adaptation::learn_variance bookkeeping (windowed_adaptation)
        const int wc = s.win_counter;
        const bool w_act = window_active(s, np.warmup), w_end = window_end(s, np.warmup);
        int win_n = s.win_n;
        if (w_act) { win_n += 1; welf = true; wn = win_n; }
        if (w_end) {
            window_next(s, np.warmup);                           // compute_next_window
            wend = true; wn = win_n;
            win_n = 0;
            redo_eps = true;
        }
        s.win_n = win_n;
        s.win_counter = wc + 1;
    }
    s.iter = iter + 1;
    if (warm && iter + 1 == np.warmup) s.eps = exp(s.da_xbar);       // complete_adaptation
    if (iter + 1 >= np.warmup + np.n_draws) { s.phase = PH_DONE; return 0; }
    if (redo_eps && iter + 1 < np.warmup) { s.phase = PH_EPS; s.eps_dir = 0; s.eps_trials = 0; return 3; }
    return 1;
}

}  // namespace bdrt
