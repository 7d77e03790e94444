// bdrt_nuts.hip -- device-resident NUTS (replaces StanModel.sampling, reference bayes_drt/inversion.py:1218-1221).
//
// MI355X design: one workgroup owns 16 chains for the whole run.  Every loop iteration is one leapfrog for those
// 16 chains: kick/drift (vector pass), the MFMA log-posterior+gradient tile (bdrt_device.h), second kick, then the
// NUTS bookkeeping (multinomial sampling, U-turn checks, tree doubling, step-size / metric adaptation) -- all on
// the device, so there is no host round trip per gradient evaluation.  Chains are asynchronous: each advances
// through its own iterations / tree depths; only the leapfrog itself is lock-step inside a workgroup.  The host
// relaunches the kernel in bounded slices (`rounds` leapfrogs per launch) and never reads anything back until
// the end; workgroups never communicate, so there is no grid barrier and no inter-workgroup hand-off.
//
// Algorithm: Stan 2.19 multinomial NUTS with diagonal metric:
//   * tree doubling with uniform direction, biased progressive sampling between the old trajectory and the new
//     subtree, uniform (multinomial) sampling inside the new subtree -- realised as weighted reservoir sampling
//     over the leaves in generation order (same distribution as Stan's pairwise merging);
//   * generalised U-turn criterion p#_left.rho > 0 && p#_right.rho > 0 on every completed sub-subtree; the
//     sub-subtree rho's come from a running sum and <= max_depth checkpoints (stored at even leaves only);
//   * divergence when H - H0 > 1000; max tree depth 10;
//   * warm-up: step-size heuristic + dual averaging (delta, gamma, t0, kappa), windowed diagonal metric
//     (init_buffer 75 / base_window 25 doubling / term_buffer 50, regularised variance).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bdrt_host.h"
#include "bdrt_nuts_device.h"

namespace bdrt {

constexpr int MAXD = 10;            // checkpoint slots (>= max_treedepth)
constexpr int NQ_CHK = 2 * MAXD + 2;
static_assert(NW * NQ_CHK * NC <= MIN_LR * NC, "Lr buffer too small for the NUTS reduction scratch");

// state vectors per workgroup, each [D][16]
enum { V_TH = 0, V_P, V_G, V_THM, V_PM, V_GM, V_THP, V_PP, V_GP, V_THS, V_GS, V_THQ, V_GQ, V_RHO, V_RHOC, V_MINV,
       V_WMEAN, V_WM2, V_CKC /* MAXD */, V_CKP = V_CKC + MAXD /* MAXD */, V_COUNT = V_CKP + MAXD };

// flag / value slots broadcast through LDS: [slot][16]
enum { F_EPS = 0, F_ACT, F_LPN, F_KIN, F_NONFIN, F_COPYQ, F_CUR2S, F_TREE, F_EVEN, F_CKIDX, F_NM, F_LAST, F_DIR,
       F_UPDS, F_ENDT, F_WELF, F_WN, F_WEND, F_DRAW, F_NEXT, F_ITER, F_TRIAL, F_ATT, F_KIN0, F_RED /* NQ_CHK */,
       F_COUNT = F_RED + NQ_CHK };

template <int NQ>
__device__ __forceinline__ void chain_reduce_big(double (&v)[NQ], double *red, double *out, int tid)
{
    // same as chain_reduce but with a caller-provided scratch of NW*NQ*NC doubles
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double x = v[q];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        if (lane < NC) red[(wave * NQ + q) * NC + lane] = x;
    }
    __syncthreads();
    for (int i = tid; i < NQ * NC; i += NT) {
        const int q = i / NC, c = i % NC;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[(w * NQ + q) * NC + c];
        out[q * NC + c] = s;
    }
    __syncthreads();
}

struct NutsArgs {
    double *vecs;          // [n_wg][V_COUNT][D][16]
    ChainState *states;    // [n_units]
    double *draws;         // [n_units][n_draws][D]
    double *lp_draws;      // [n_units][n_draws]
    unsigned long long *leap_counter;   // total leapfrogs (all chains)
    int *done_counter;     // workgroups whose chains are all finished
    int n_units;
    int rounds;
};

__global__ __launch_bounds__(NT) void nuts_kernel(DevProblem P, NutsParams np, NutsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int c = tid & (NC - 1), g = tid >> 4;
    const int D = P.D;
    const int wg = blockIdx.x;
    const int c0 = wg * NC;
    const int nvalid = min(NC, a.n_units - c0);
    const bool valid = c < nvalid;

    // LDS carve-up: [tile region | flags | chain states | spec]
    const size_t tile_doubles = lds_doubles(P);
    double *fl = smem + tile_doubles;
    ChainState *sts = reinterpret_cast<ChainState *>(fl + F_COUNT * NC);
    int *spec = reinterpret_cast<int *>(sts + NC);
    // scratch for the big reduction: the tile's Lr buffer is free outside logp_grad_tile
    double *bigred = smem + (size_t)P.XR * NC + (size_t)P.ZR * NC * (1 + P.npar);

    double *V = a.vecs + (size_t)wg * V_COUNT * D * NC;
    auto vec = [&](int v) -> double * { return V + (size_t)v * D * NC; };

    if (tid < NC) {
        if (valid) sts[c] = a.states[c0 + c];
        else { memset(&sts[c], 0, sizeof(ChainState)); sts[c].phase = PH_DONE; }
        spec[c] = valid ? sts[c].spec : 0;
    }
    __syncthreads();
    // (re)derive the leapfrog inputs of the first round from the stored state
    if (tid < NC) {
        ChainState &s = sts[c];
        const bool act = s.phase == PH_INIT || s.phase == PH_EPS || s.phase == PH_TREE;
        fl[F_ACT * NC + c] = act ? 1.0 : 0.0;
        double e = 0.0;
        if (s.phase == PH_EPS) e = s.eps;
        else if (s.phase == PH_TREE) e = s.dir * s.eps;
        fl[F_EPS * NC + c] = e;
    }
    __syncthreads();

    TileIO io;
    io.theta = vec(V_TH); io.t_sc = 1; io.t_sj = NC;
    io.grad = vec(V_G); io.g_sc = 1; io.g_sj = NC;
    io.lp = fl + F_LPN * NC;
    io.spec = spec;
    io.nvalid = NC;            // padded columns carry a DONE state and finite vectors
    io.jacobian = 1;
    io.Z_hat = nullptr; io.sigma_tot = nullptr; io.params = nullptr;

    unsigned long long my_leaps = 0;

    for (int round = 0; round < a.rounds; ++round) {
        // ---- any chain still working? ---------------------------------------------------------------
        {
            const double act = fl[F_ACT * NC + c];
            if (!__syncthreads_or(act != 0.0)) break;
        }
        const double e = fl[F_EPS * NC + c];
        const bool act = fl[F_ACT * NC + c] != 0.0;

        // ---- A: half kick + drift --------------------------------------------------------------------
        if (act) {
            double *TH = vec(V_TH), *Pm = vec(V_P);
            const double *G = vec(V_G), *MI = vec(V_MINV);
            for (int j = g; j < D; j += NG) {
                const int i = j * NC + c;
                const double p = Pm[i] + 0.5 * e * G[i];
                Pm[i] = p;
                TH[i] += e * MI[i] * p;
            }
        }
        __syncthreads();

        // ---- B: log-posterior + gradient at the new point (MFMA tile) ------------------------------------
        logp_grad_tile(P, io, smem);

        // ---- C: second half kick, kinetic energy, finiteness of the gradient -----------------------------
        {
            double acc[2] = {0.0, 0.0};
            if (act) {
                double *Pm = vec(V_P);
                const double *G = vec(V_G), *MI = vec(V_MINV);
                for (int j = g; j < D; j += NG) {
                    const int i = j * NC + c;
                    const double gj = G[i];
                    const double p = Pm[i] + 0.5 * e * gj;
                    Pm[i] = p;
                    acc[0] += MI[i] * p * p;
                    acc[1] += isfinite(gj) ? 0.0 : 1.0;
                }
            }
            chain_reduce<2>(acc, smem + (size_t)P.XR * NC + (size_t)P.ZR * NC * (1 + P.npar) + (size_t)P.LR * NC,
                            fl + F_KIN * NC, tid);   // F_KIN, F_NONFIN are consecutive slots
        }

        // ---- S1: per-chain scalar logic after the evaluation -----------------------------------------------
        if (tid < NC) {
            ChainState &s = sts[c];
            const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};
            fl[F_COPYQ * NC + c] = 0; fl[F_CUR2S * NC + c] = 0; fl[F_TREE * NC + c] = 0; fl[F_NM * NC + c] = 0;
            fl[F_LAST * NC + c] = 0; fl[F_UPDS * NC + c] = 0; fl[F_ENDT * NC + c] = 0; fl[F_WELF * NC + c] = 0;
            fl[F_WEND * NC + c] = 0; fl[F_DRAW * NC + c] = -1; fl[F_NEXT * NC + c] = 0; fl[F_EVEN * NC + c] = 0;
            fl[F_DIR * NC + c] = s.dir;
            if (act) {
                const double lp = fl[F_LPN * NC + c];
                const double kin = 0.5 * fl[F_KIN * NC + c];
                const bool finite_pt = isfinite(lp) && fl[F_NONFIN * NC + c] == 0.0;
                if (s.phase == PH_INIT) {
                    if (finite_pt) {
                        s.lps = lp;
                        fl[F_CUR2S * NC + c] = 1;
                        s.phase = PH_EPS; s.eps_dir = 0; s.eps_trials = 0;
                        fl[F_NEXT * NC + c] = 3;
                    } else {
                        s.init_attempt += 1;
                        if (s.init_attempt >= 100) { s.phase = PH_FAILED; }
                        else fl[F_NEXT * NC + c] = 4;
                    }
                } else if (s.phase == PH_EPS) {
                    // Stan base_hmc::init_stepsize
                    my_leaps += 1;
                    double h = -lp + kin;
                    if (isnan(h)) h = INFINITY;
                    const double dH = s.H0 - h;
                    const double thr = -0.2231435513142097557662950903;   // log(0.8)
                    bool finished = false;
                    if (s.eps_trials == 0) {
                        s.eps_dir = dH > thr ? 1 : -1;
                    } else {
                        if (s.eps_dir == 1 && !(dH > thr)) finished = true;
                        else if (s.eps_dir == -1 && !(dH < thr)) finished = true;
                        else s.eps = s.eps_dir == 1 ? 2.0 * s.eps : 0.5 * s.eps;
                        if (!(s.eps > 1e-300) || s.eps > 1e7) finished = true;   // Stan throws here; we stop adapting
                    }
                    s.eps_trials += 1;
                    if (finished) {
                        // services::sample::hmc_nuts_diag_e_adapt sets mu = log(10*stepsize) from the CONFIGURED step
                        // size before the first init_stepsize; after a metric update mu = log(10*eps) (adapt_diag_e_nuts)
                        s.da_mu = s.iter == 0 ? log(10.0 * np.stepsize0) : log(10.0 * s.eps);
                        da_restart(s);
                        s.phase = PH_TREE;
                        fl[F_NEXT * NC + c] = 1;
                    } else {
                        fl[F_NEXT * NC + c] = 3;
                    }
                } else {   // PH_TREE: one new leaf
                    my_leaps += 1;
                    s.n_leap_iter += 1;
                    double h = -lp + kin;
                    if (isnan(h)) h = INFINITY;
                    const bool divergent = (h - s.H0) > np.max_deltaH;
                    const double w = s.H0 - h;
                    s.sum_metro += w > 0.0 ? 1.0 : exp(w);
                    if (divergent) {
                        fl[F_ENDT * NC + c] = 2;               // transition ends, subtree discarded, divergent
                    } else {
                        const double lsw_new = log_sum_exp2(s.lsw_sub, w);
                        // uniform sampling inside the new subtree: keep leaf i with probability w_i / W_i
                        const double u = rng_uniform(rng, (uint32_t)s.leaf, RNG_LEAF, (uint32_t)s.depth, 0, (uint32_t)s.iter);
                        if (s.leaf == 0 || u < exp(w - lsw_new)) { fl[F_COPYQ * NC + c] = 1; s.lpq = lp; }
                        s.lsw_sub = lsw_new;
                        fl[F_TREE * NC + c] = 1;
                        const int i = s.leaf;
                        if ((i & 1) == 0) {
                            fl[F_EVEN * NC + c] = 1;
                            fl[F_CKIDX * NC + c] = __popc((unsigned)(i >> 1));
                        } else {
                            int t = 0;
                            while ((i >> t) & 1) ++t;          // trailing ones = completed sub-subtrees ending here
                            fl[F_NM * NC + c] = t;
                            fl[F_CKIDX * NC + c] = __popc((unsigned)(i >> 1));
                        }
                        if (i == s.nleaves - 1) fl[F_LAST * NC + c] = 1;
                    }
                }
            }
        }
        __syncthreads();

        // ---- D: proposal copy, checkpoints, running rho, U-turn dot products, speculative subtree close ------
        {
            double acc[NQ_CHK];
#pragma unroll
            for (int q = 0; q < NQ_CHK; ++q) acc[q] = 0.0;
            const bool copyq = fl[F_COPYQ * NC + c] != 0.0, cur2s = fl[F_CUR2S * NC + c] != 0.0;
            const bool tree = fl[F_TREE * NC + c] != 0.0, even = fl[F_EVEN * NC + c] != 0.0;
            const bool last = fl[F_LAST * NC + c] != 0.0;
            const int nm = (int)fl[F_NM * NC + c], ck = (int)fl[F_CKIDX * NC + c];
            const int dir = (int)fl[F_DIR * NC + c];
            const int leaf0 = tree && (sts[c].leaf == 0);
            if (copyq || cur2s || tree) {
                const double *TH = vec(V_TH), *Pm = vec(V_P), *G = vec(V_G), *MI = vec(V_MINV);
                double *THQ = vec(V_THQ), *GQ = vec(V_GQ), *THS = vec(V_THS), *GS = vec(V_GS);
                double *RHOC = vec(V_RHOC), *RHO = vec(V_RHO);
                double *CKC = vec(V_CKC + (ck < MAXD ? ck : 0)), *CKP = vec(V_CKP + (ck < MAXD ? ck : 0));
                double *THE = vec(dir > 0 ? V_THP : V_THM), *PE = vec(dir > 0 ? V_PP : V_PM), *GE = vec(dir > 0 ? V_GP : V_GM);
                const double *PO = vec(dir > 0 ? V_PM : V_PP);     // momentum at the other end
                for (int j = g; j < D; j += NG) {
                    const int i = j * NC + c;
                    const double th = TH[i], p = Pm[i], gj = G[i];
                    if (copyq) { THQ[i] = th; GQ[i] = gj; }
                    if (cur2s) { THS[i] = th; GS[i] = gj; }
                    if (tree) {
                        const double mi = MI[i];
                        const double before = leaf0 ? 0.0 : RHOC[i];
                        if (even) { CKC[i] = before; CKP[i] = p; }
                        const double rc = before + p;
                        RHOC[i] = rc;
#pragma unroll
                        for (int l = 0; l < MAXD; ++l) {
                            if (l < nm) {
                                const int idx = ck - l;
                                const double rho = rc - V[((size_t)(V_CKC + idx) * D) * NC + i];
                                acc[2 * l] += mi * V[((size_t)(V_CKP + idx) * D) * NC + i] * rho;
                                acc[2 * l + 1] += mi * p * rho;
                            }
                        }
                        if (last) {
                            // speculative close of the subtree: extend the trajectory end and test the whole trajectory
                            const double rt = RHO[i] + rc;
                            RHO[i] = rt;
                            THE[i] = th; PE[i] = p; GE[i] = gj;
                            acc[2 * MAXD] += mi * PO[i] * rt;
                            acc[2 * MAXD + 1] += mi * p * rt;
                        }
                    }
                }
            }
            chain_reduce_big<NQ_CHK>(acc, bigred, fl + F_RED * NC, tid);
        }

        // ---- S2: validity of the new subtree, trajectory-level decisions, adaptation scalars ---------------------
        if (tid < NC) {
            ChainState &s = sts[c];
            const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};
            int endt = (int)fl[F_ENDT * NC + c];
            if (fl[F_TREE * NC + c] != 0.0) {
                const int nm = (int)fl[F_NM * NC + c];
                bool ok = true;
                for (int l = 0; l < nm; ++l)
                    ok = ok && (fl[(F_RED + 2 * l) * NC + c] > 0.0) && (fl[(F_RED + 2 * l + 1) * NC + c] > 0.0);
                if (!ok) {
                    endt = 1;                                   // U-turn inside the new subtree: discard it, stop
                } else if (fl[F_LAST * NC + c] != 0.0) {
                    // subtree complete and valid (Stan base_nuts::transition after build_tree)
                    s.depth += 1;
                    bool take;
                    if (s.lsw_sub > s.lsw) take = true;
                    else take = rng_uniform(rng, 0, RNG_TOP, (uint32_t)s.depth, 0, (uint32_t)s.iter) < exp(s.lsw_sub - s.lsw);
                    if (take) { fl[F_UPDS * NC + c] = 1; s.lps = s.lpq; }
                    s.lsw = log_sum_exp2(s.lsw, s.lsw_sub);
                    const bool keep_going = (fl[(F_RED + 2 * MAXD) * NC + c] > 0.0) && (fl[(F_RED + 2 * MAXD + 1) * NC + c] > 0.0);
                    if (!keep_going || s.depth >= np.max_depth) endt = 1;
                    else {
                        // next doubling
                        s.dir = rng_uniform(rng, 0, RNG_DIRECTION, (uint32_t)s.depth, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                        s.leaf = 0; s.nleaves = 1 << s.depth; s.lsw_sub = -INFINITY;
                        fl[F_NEXT * NC + c] = 2;
                    }
                } else {
                    s.leaf += 1;
                }
            }
            if (endt) {
                // end of transition (Stan adapt_diag_e_nuts::transition)
                const double accept = s.sum_metro / (double)(s.n_leap_iter > 0 ? s.n_leap_iter : 1);
                const bool warm = s.iter < np.warmup;
                s.n_leap_total += s.n_leap_iter;
                if (!warm) {
                    s.n_post += 1;
                    s.sum_accept += accept;
                    if (endt == 2) s.n_div += 1;
                    if (s.depth >= np.max_depth) s.n_maxdepth += 1;
                    fl[F_DRAW * NC + c] = s.iter - np.warmup;
                    if (a.lp_draws) a.lp_draws[(size_t)(c0 + c) * np.n_draws + (s.iter - np.warmup)] = s.lps;
                }
                bool redo_eps = false;
                if (warm) {
                    da_learn(s, np, accept);
                    if (window_active(s, np.warmup)) { s.win_n += 1; fl[F_WELF * NC + c] = 1; fl[F_WN * NC + c] = s.win_n; }
                    if (window_end(s, np.warmup)) {
                        window_next(s, np.warmup);
                        fl[F_WEND * NC + c] = 1; fl[F_WN * NC + c] = s.win_n;
                        s.win_n = 0;
                        redo_eps = true;
                    }
                    s.win_counter += 1;
                }
                s.iter += 1;
                if (warm && s.iter == np.warmup) s.eps = exp(s.da_xbar);       // complete_adaptation
                fl[F_ENDT * NC + c] = 1;
                if (s.iter >= np.warmup + np.n_draws) {
                    s.phase = PH_DONE;
                    fl[F_NEXT * NC + c] = 0;
                } else if (redo_eps && s.iter < np.warmup) {
                    s.phase = PH_EPS; s.eps_dir = 0; s.eps_trials = 0;
                    fl[F_NEXT * NC + c] = 3;
                } else {
                    fl[F_NEXT * NC + c] = 1;
                }
            }
            fl[F_ITER * NC + c] = s.iter;
            fl[F_TRIAL * NC + c] = s.eps_trials;
            fl[F_ATT * NC + c] = s.init_attempt;
            fl[F_DIR * NC + c] = s.dir;
        }
        __syncthreads();

        // ---- E: sample update, metric adaptation, draw output, preparation of the next leapfrog -----------------
        {
            double acc[1] = {0.0};
            const bool upds = fl[F_UPDS * NC + c] != 0.0, welf = fl[F_WELF * NC + c] != 0.0;
            const bool wend = fl[F_WEND * NC + c] != 0.0;
            const int draw = (int)fl[F_DRAW * NC + c], next = (int)fl[F_NEXT * NC + c];
            const int dir = (int)fl[F_DIR * NC + c];
            if (upds || welf || wend || draw >= 0 || next) {
                const ChainState &s = sts[c];
                const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};
                const uint32_t iter = (uint32_t)fl[F_ITER * NC + c], trial = (uint32_t)fl[F_TRIAL * NC + c];
                const uint32_t att = (uint32_t)fl[F_ATT * NC + c];
                const double wn = fl[F_WN * NC + c];
                double *TH = vec(V_TH), *Pm = vec(V_P), *G = vec(V_G), *MI = vec(V_MINV);
                double *THS = vec(V_THS), *GS = vec(V_GS);
                const double *THQ = vec(V_THQ), *GQ = vec(V_GQ);
                double *WM = vec(V_WMEAN), *W2 = vec(V_WM2);
                double *dr = (draw >= 0 && valid) ? a.draws + ((size_t)(c0 + c) * np.n_draws + draw) * D : nullptr;
                for (int j = g; j < D; j += NG) {
                    const int i = j * NC + c;
                    double ths = THS[i], gs = GS[i];
                    if (upds) { ths = THQ[i]; gs = GQ[i]; THS[i] = ths; GS[i] = gs; }
                    double mi = MI[i];
                    if (welf) {            // Welford (stan::math::welford_var_estimator)
                        const double delta = ths - WM[i];
                        const double mean = WM[i] + delta / wn;
                        WM[i] = mean;
                        W2[i] += (ths - mean) * delta;
                    }
                    if (wend) {            // var_adaptation::learn_variance
                        const double var = wn > 1.0 ? W2[i] / (wn - 1.0) : 0.0;
                        mi = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
                        MI[i] = mi;
                        WM[i] = 0.0; W2[i] = 0.0;
                    }
                    if (dr) dr[j] = ths;
                    if (next == 1 || next == 3) {
                        // fresh momentum p ~ N(0, M), M = diag(1/Minv); restart from the current sample
                        const double z = next == 1 ? rng_normal(rng, (uint32_t)j, RNG_MOMENTUM, 0, iter)
                                                   : rng_normal(rng, (uint32_t)j, RNG_EPS_MOMENTUM, trial, iter);
                        const double p = z / sqrt(mi);
                        Pm[i] = p; TH[i] = ths; G[i] = gs;
                        acc[0] += mi * p * p;
                        if (next == 1) {
                            V[((size_t)V_THM * D) * NC + i] = ths; V[((size_t)V_THP * D) * NC + i] = ths;
                            V[((size_t)V_PM * D) * NC + i] = p; V[((size_t)V_PP * D) * NC + i] = p;
                            V[((size_t)V_GM * D) * NC + i] = gs; V[((size_t)V_GP * D) * NC + i] = gs;
                            V[((size_t)V_RHO * D) * NC + i] = p;
                        }
                    } else if (next == 2) {
                        // continue from the trajectory end in direction dir
                        TH[i] = V[((size_t)(dir > 0 ? V_THP : V_THM) * D) * NC + i];
                        Pm[i] = V[((size_t)(dir > 0 ? V_PP : V_PM) * D) * NC + i];
                        G[i] = V[((size_t)(dir > 0 ? V_GP : V_GM) * D) * NC + i];
                    } else if (next == 4) {
                        TH[i] = np.init_radius * (2.0 * rng_uniform(rng, (uint32_t)j, RNG_INIT, 0, att, 0) - 1.0);
                        Pm[i] = 0.0; G[i] = 0.0;
                    }
                }
            }
            chain_reduce<1>(acc, smem + (size_t)P.XR * NC + (size_t)P.ZR * NC * (1 + P.npar) + (size_t)P.LR * NC,
                            fl + F_KIN0 * NC, tid);
        }

        // ---- S3: Hamiltonian at the start point, step for the next round -------------------------------------------
        if (tid < NC) {
            ChainState &s = sts[c];
            const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};
            const int next = (int)fl[F_NEXT * NC + c];
            if (next == 1) {
                s.H0 = -s.lps + 0.5 * fl[F_KIN0 * NC + c];
                s.lsw = 0.0; s.lsw_sub = -INFINITY; s.depth = 0; s.leaf = 0; s.nleaves = 1;
                s.n_leap_iter = 0; s.sum_metro = 0.0;
                s.dir = rng_uniform(rng, 0, RNG_DIRECTION, 0, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
            } else if (next == 3) {
                s.H0 = -s.lps + 0.5 * fl[F_KIN0 * NC + c];
            }
            const bool on = s.phase == PH_INIT || s.phase == PH_EPS || s.phase == PH_TREE;
            fl[F_ACT * NC + c] = on ? 1.0 : 0.0;
            double en = 0.0;
            if (s.phase == PH_EPS) en = s.eps;
            else if (s.phase == PH_TREE) en = s.dir * s.eps;
            fl[F_EPS * NC + c] = en;
        }
        __syncthreads();
    }

    // ---- write the chain states back -----------------------------------------------------------------------------
    __syncthreads();
    if (tid < NC && valid) a.states[c0 + c] = sts[c];
    if (tid < NC) {
        unsigned long long x = my_leaps;
        for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (tid == 0 && x) atomicAdd(a.leap_counter, x);
    }
    {
        const int ph = (tid < NC) ? sts[c].phase : PH_DONE;
        const int busy = __syncthreads_or(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE);
        if (tid == 0 && !busy) atomicAdd(a.done_counter, 1);
    }
}

struct Sampler {
    Problem *prob = nullptr;
    NutsParams np;
    NutsArgs args;
    int n_units = 0, n_wg = 0, D = 0;
    size_t lds_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms_total = 0.0;
    int64_t n_launch = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int *d_done = nullptr;
    unsigned long long *d_leaps = nullptr;
    int rounds_default = 256;
};

static size_t nuts_lds_bytes(const DevProblem &P)
{
    return (lds_doubles(P) + (size_t)F_COUNT * NC) * sizeof(double) + NC * sizeof(ChainState) + NC * sizeof(int) + 16;
}

}  // namespace bdrt

using namespace bdrt;

struct bdrt_sampler {
    bdrt::Sampler impl;
};

extern "C" {

void bdrt_nuts_defaults(bdrt_nuts_control *c)
{
    c->adapt_delta = 0.9; c->adapt_t0 = 10; c->adapt_gamma = 0.05; c->adapt_kappa = 0.75;
    c->max_treedepth = 10; c->init_buffer = 75; c->term_buffer = 50; c->base_window = 25;
    c->init_radius = 2; c->max_deltaH = 1000; c->stepsize0 = 1;
}

void bdrt_sampler_destroy(bdrt_sampler *s)
{
    if (!s) return;
    Sampler &S = s->impl;
    if (S.stream) hipStreamSynchronize(S.stream);
    for (auto &pr : S.pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (S.args.vecs) hipFree(S.args.vecs);
    if (S.args.states) hipFree(S.args.states);
    if (S.args.draws) hipFree(S.args.draws);
    if (S.args.lp_draws) hipFree(S.args.lp_draws);
    if (S.d_done) hipFree(S.d_done);
    if (S.d_leaps) hipFree(S.d_leaps);
    if (S.stream) hipStreamDestroy(S.stream);
    delete s;
}

bdrt_sampler *bdrt_sampler_create(bdrt_problem *p, int n_units, const int *spec, const int *chain_id, int warmup,
                                  int n_draws, uint64_t seed, const double *init_theta, const bdrt_nuts_control *ctrl)
{
    if (!p || n_units < 1 || warmup < 0 || n_draws < 0) { set_error("bdrt_sampler_create: bad arguments"); return nullptr; }
    bdrt_nuts_control c;
    if (ctrl) c = *ctrl; else bdrt_nuts_defaults(&c);
    if (c.max_treedepth < 1 || c.max_treedepth > MAXD) { set_error("max_treedepth must be in [1,%d]", MAXD); return nullptr; }
    Problem &P = p->impl;
    bdrt_sampler *s = new bdrt_sampler();
    Sampler &S = s->impl;
    memset(&S.args, 0, sizeof(S.args));
    S.prob = &P;
    S.n_units = n_units;
    S.n_wg = (n_units + NC - 1) / NC;
    S.D = P.dev.D;
    S.np.warmup = warmup; S.np.n_draws = n_draws; S.np.max_depth = c.max_treedepth;
    S.np.delta = c.adapt_delta; S.np.gamma = c.adapt_gamma; S.np.t0 = c.adapt_t0; S.np.kappa = c.adapt_kappa;
    S.np.init_radius = c.init_radius; S.np.max_deltaH = c.max_deltaH; S.np.stepsize0 = c.stepsize0;
    S.np.seed_lo = (uint32_t)seed; S.np.seed_hi = (uint32_t)(seed >> 32);
    S.np.has_init = init_theta != nullptr;
    S.lds_bytes = nuts_lds_bytes(P.dev);
    auto fail = [&](const char *msg) -> bdrt_sampler * { set_error("%s", msg); bdrt_sampler_destroy(s); return nullptr; };
    if (S.lds_bytes > 160 * 1024) return fail("bdrt_sampler_create: problem too large for the 160 KiB LDS budget");
    for (int u = 0; u < n_units; ++u)
        if (spec && (spec[u] < 0 || spec[u] >= P.dev.n_spectra)) return fail("bdrt_sampler_create: spectrum index out of range");

    const size_t nvec = (size_t)S.n_wg * V_COUNT * S.D * NC;
    std::vector<double> hv(nvec, 0.0);
    std::vector<ChainState> hs((size_t)n_units);
    for (int u = 0; u < n_units; ++u) {
        ChainState &st = hs[u];
        memset(&st, 0, sizeof(st));
        st.phase = PH_INIT;
        st.spec = spec ? spec[u] : 0;
        st.chain_id = chain_id ? chain_id[u] : u;
        st.eps = c.stepsize0;
        st.dir = 1;
        st.lsw_sub = -INFINITY;
        window_init(st, warmup, c.init_buffer, c.term_buffer, c.base_window);
        const int wg = u / NC, cc = u % NC;
        double *V = hv.data() + (size_t)wg * V_COUNT * S.D * NC;
        const Philox rng = {S.np.seed_lo, S.np.seed_hi, (uint32_t)st.chain_id};
        for (int j = 0; j < S.D; ++j) {
            V[((size_t)V_MINV * S.D + j) * NC + cc] = 1.0;
            V[((size_t)V_TH * S.D + j) * NC + cc] =
                init_theta ? init_theta[(size_t)u * S.D + j]
                           : c.init_radius * (2.0 * rng_uniform(rng, (uint32_t)j, RNG_INIT, 0, 0, 0) - 1.0);
        }
        if (!init_theta) st.init_attempt = 0;
    }
    // padded columns of the last workgroup: finite placeholders
    for (int u = n_units; u < S.n_wg * NC; ++u) {
        const int wg = u / NC, cc = u % NC;
        double *V = hv.data() + (size_t)wg * V_COUNT * S.D * NC;
        for (int j = 0; j < S.D; ++j) V[((size_t)V_MINV * S.D + j) * NC + cc] = 1.0;
    }
    if (hipMalloc((void **)&S.args.vecs, nvec * sizeof(double)) != hipSuccess) return fail("hipMalloc(vecs) failed");
    if (hipMalloc((void **)&S.args.states, hs.size() * sizeof(ChainState)) != hipSuccess) return fail("hipMalloc(states) failed");
    const size_t nd = (size_t)n_units * std::max(n_draws, 1) * S.D;
    if (hipMalloc((void **)&S.args.draws, nd * sizeof(double)) != hipSuccess) return fail("hipMalloc(draws) failed");
    if (hipMalloc((void **)&S.args.lp_draws, (size_t)n_units * std::max(n_draws, 1) * sizeof(double)) != hipSuccess)
        return fail("hipMalloc(lp) failed");
    if (hipMalloc((void **)&S.d_done, sizeof(int)) != hipSuccess) return fail("hipMalloc failed");
    if (hipMalloc((void **)&S.d_leaps, sizeof(unsigned long long)) != hipSuccess) return fail("hipMalloc failed");
    hipMemcpy(S.args.vecs, hv.data(), nvec * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(S.args.states, hs.data(), hs.size() * sizeof(ChainState), hipMemcpyHostToDevice);
    hipMemset(S.args.draws, 0, nd * sizeof(double));
    hipMemset(S.d_leaps, 0, sizeof(unsigned long long));
    S.args.leap_counter = S.d_leaps;
    S.args.done_counter = S.d_done;
    S.args.n_units = n_units;
    if (hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    static size_t attr_bytes = 0;
    if (S.lds_bytes > attr_bytes) {
        hipError_t e = hipFuncSetAttribute((const void *)nuts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)S.lds_bytes);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(nuts_kernel, %zu B dynamic LDS) failed: %s", S.lds_bytes, hipGetErrorString(e));
            bdrt_sampler_destroy(s);
            return nullptr;
        }
        attr_bytes = S.lds_bytes;
    }
    return s;
}

static int harvest_events(Sampler &S, bool wait)
{
    size_t k = 0;
    for (; k < S.pending.size(); ++k) {
        auto &pr = S.pending[k];
        if (!wait && hipEventQuery(pr.second) != hipSuccess) break;
        if (wait) BDRT_HIP(hipEventSynchronize(pr.second));
        float ms = 0.f;
        BDRT_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        S.ms_total += ms;
        hipEventDestroy(pr.first); hipEventDestroy(pr.second);
    }
    S.pending.erase(S.pending.begin(), S.pending.begin() + k);
    return 0;
}

int bdrt_sampler_advance(bdrt_sampler *s, int rounds, int *all_done)
{
    if (!s || rounds < 1) { set_error("bdrt_sampler_advance: bad arguments"); return -1; }
    Sampler &S = s->impl;
    S.args.rounds = rounds;
    BDRT_HIP(hipMemsetAsync(S.d_done, 0, sizeof(int), S.stream));
    hipEvent_t e0, e1;
    BDRT_HIP(hipEventCreate(&e0));
    BDRT_HIP(hipEventCreate(&e1));
    BDRT_HIP(hipEventRecord(e0, S.stream));
    hipLaunchKernelGGL(nuts_kernel, dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, S.prob->dev, S.np, S.args);
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipEventRecord(e1, S.stream));
    S.pending.emplace_back(e0, e1);
    S.n_launch += 1;
    if (all_done) {
        int done = 0;
        BDRT_HIP(hipMemcpyAsync(&done, S.d_done, sizeof(int), hipMemcpyDeviceToHost, S.stream));
        BDRT_HIP(hipStreamSynchronize(S.stream));
        *all_done = done >= S.n_wg;
        return harvest_events(S, true);
    }
    return harvest_events(S, false);
}

int bdrt_sampler_sync(bdrt_sampler *s)
{
    if (!s) return -1;
    BDRT_HIP(hipStreamSynchronize(s->impl.stream));
    return harvest_events(s->impl, true);
}

int bdrt_sampler_run(bdrt_sampler *s)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    // upper bound on the leapfrogs one chain can need: (2^depth - 1 + step-size trials) per iteration
    const long long per_iter = (1LL << S.np.max_depth) + 64;
    const long long bound = per_iter * (S.np.warmup + S.np.n_draws + 2) + 200;
    long long spent = 0;
    int done = 0;
    while (!done && spent <= bound) {
        int rc = bdrt_sampler_advance(s, S.rounds_default, &done);
        if (rc) return rc;
        spent += S.rounds_default;
    }
    if (!done) { set_error("bdrt_sampler_run: chains did not finish within the leapfrog bound"); return -3; }
    return 0;
}

int bdrt_sampler_results(bdrt_sampler *s, double *draws, double *lp, bdrt_chain_diag *diag)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    BDRT_HIP(hipStreamSynchronize(S.stream));
    const size_t nd = (size_t)S.n_units * S.np.n_draws;
    if (draws && nd) BDRT_HIP(hipMemcpy(draws, S.args.draws, nd * S.D * sizeof(double), hipMemcpyDeviceToHost));
    if (lp && nd) BDRT_HIP(hipMemcpy(lp, S.args.lp_draws, nd * sizeof(double), hipMemcpyDeviceToHost));
    if (diag) {
        std::vector<ChainState> hs((size_t)S.n_units);
        BDRT_HIP(hipMemcpy(hs.data(), S.args.states, hs.size() * sizeof(ChainState), hipMemcpyDeviceToHost));
        for (int u = 0; u < S.n_units; ++u) {
            diag[u].n_leapfrog = hs[u].n_leap_total;
            diag[u].n_divergent = hs[u].n_div;
            diag[u].n_max_treedepth = hs[u].n_maxdepth;
            diag[u].stepsize = hs[u].eps;
            diag[u].mean_accept = hs[u].n_post ? hs[u].sum_accept / hs[u].n_post : 0.0;
            if (hs[u].phase == PH_FAILED) diag[u].n_leapfrog = -1;
        }
    }
    return 0;
}

int64_t bdrt_sampler_total_leapfrogs(bdrt_sampler *s)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    unsigned long long v = 0;
    if (hipStreamSynchronize(S.stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, S.d_leaps, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

int bdrt_sampler_kernel_time(bdrt_sampler *s, double *ms_total, int64_t *n_launches, int reset)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    BDRT_HIP(hipStreamSynchronize(S.stream));
    int rc = harvest_events(S, true);
    if (rc) return rc;
    if (ms_total) *ms_total = S.ms_total;
    if (n_launches) *n_launches = S.n_launch;
    if (reset) { S.ms_total = 0.0; S.n_launch = 0; }
    return 0;
}

int bdrt_sample(bdrt_problem *p, int n_units, const int *spec, const int *chain_id, int warmup, int n_draws,
                uint64_t seed, const double *init_theta, const bdrt_nuts_control *ctrl, double *draws, double *lp,
                bdrt_chain_diag *diag)
{
    bdrt_sampler *s = bdrt_sampler_create(p, n_units, spec, chain_id, warmup, n_draws, seed, init_theta, ctrl);
    if (!s) return -1;
    int rc = bdrt_sampler_run(s);
    if (rc == 0) rc = bdrt_sampler_results(s, draws, lp, diag);
    bdrt_sampler_destroy(s);
    return rc;
}

}  // extern "C"
