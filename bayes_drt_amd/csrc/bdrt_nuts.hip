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
//     sub-subtree rho's are kept per level (binary counter over the leaf index), summed in the order of the recursion;
//   * divergence when H - H0 > 1000; max tree depth 10;
//   * warm-up: step-size heuristic + dual averaging (delta, gamma, t0, kappa), windowed diagonal metric
//     (init_buffer 75 / base_window 25 doubling / term_buffer 50, regularised variance).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "bdrt_host.h"
#include "bdrt_lbfgs.h"
#include "bdrt_nuts_device.h"
#include "bdrt_solo.h"
#include "bdrt_wave.h"
#include "bdrt_nuts_args.h"
#include "bdrt_big.h"

namespace bdrt {

// one-chain-per-wave sampler / evaluator (bdrt_wave.hip)
size_t wave_lds_request(const WaveGeom &g, int n_wg, int n_cu, int *nhot, int max_per_cu);
int launch_wave_nuts(const DevProblem *dp, const NutsParams &np, const NutsArgs &args, const WaveGeom &g, int nhot, int n_wg, size_t lds,
                     hipStream_t stream, int outlier_model);
int launch_wave_eval(const DevProblem *dp, const WaveGeom &g, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                     double *d_grad, int n_wg, size_t lds, hipStream_t stream, int outlier_model);

}  // namespace bdrt
#include "bdrt_nuts16.h"
namespace bdrt {
// (defined in bdrt_nuts_k0.hip .. bdrt_nuts_k4.hip)
BDRT_NUTS16_G0(BDRT_NUTS16_DECLARE) BDRT_NUTS16_G1(BDRT_NUTS16_DECLARE) BDRT_NUTS16_G2(BDRT_NUTS16_DECLARE)
BDRT_NUTS16_G3(BDRT_NUTS16_DECLARE) BDRT_NUTS16_G4(BDRT_NUTS16_DECLARE) BDRT_NUTS16_G5(BDRT_NUTS16_DECLARE_PROF)



// ---------------------------------------------------------------------------------------------------------------------------
// One chain per workgroup (bdrt_solo.h): the same transition logic as nuts_kernel, element j of every vector in thread j,
// all vectors in LDS.  Global state layout: vecs [n_units][SG_COUNT][ds]; states [n_units].
// ---------------------------------------------------------------------------------------------------------------------------

template <int WPE, bool PROF = false>   // waves per SIMD the register budget allows: 2 = one workgroup per CU, 4 = two (when their LDS fits); PROF: fills the phase profile
__global__ __launch_bounds__(SOLO_NT, WPE) void nuts_solo_kernel(const DevProblem *__restrict__ Pp, NutsParams np, NutsArgs a, SoloGeom g)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = a.unit_map ? a.unit_map[blockIdx.x] : blockIdx.x;
    const int D = g.D, DS = a.ds, j = tid;
    const bool own = j < D;                                // this thread owns element j of the D-vectors
    double *Vg = a.vecs + (size_t)blockIdx.x * SG_COUNT * DS;    // global rows
    double *V = smem + g.o_vec;                            // LDS rows
    // WPE 4 (two workgroups per CU): SOLO_NHOT rows in LDS, the others where they are in HBM (each thread touches its own element)
    constexpr bool TRIM = WPE == 4;
    auto row = [&](int v) -> double * {
        if constexpr (!TRIM) return V + (size_t)v * g.DSS;
        else { const int h = solo_hot_slot(v); return h >= 0 ? V + (size_t)h * g.DSS : Vg + (size_t)v * DS; }
    };
    double *red = smem + g.o_red;
    double *zrow = smem + g.o_z;
    double *lps_l = smem + g.o_scv + 12;                   // lp of the last evaluation
    int slot = 0;

    // the chain's scalar state lives in LDS; every thread keeps its hot part in registers (identical updates)
    ChainState *cold = reinterpret_cast<ChainState *>(smem + g.o_state);
    if (tid == 0) chain_state_copy(*cold, a.states[unit]);          // (member by member: a struct assignment goes through scratch)
#pragma unroll 7
    for (int v = 0; v < SV_COUNT; ++v) {
        if (TRIM && solo_hot_slot(v) < 0) continue;
        const double x = own ? Vg[(size_t)v * DS + j] : 0.0;
        if (j < g.DSS) row(v)[j] = x;
    }
    solo_eval_init(P, g, smem, tid);
    __syncthreads();
    // one workgroup per CU: the whole state in registers (256 VGPRs to spend); two per CU: the hot part only
    typedef typename std::conditional<WPE == 2, ChainState, SoloHot>::type State;
    State s;
    if constexpr (WPE == 2) chain_state_copy(s, *cold); else s.from(*cold);
    const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)cold->chain_id};
    const SoloEvalRegs er = solo_eval_setup(P, g, cold->spec, tid);
    double *TH = row(SV_TH), *Pm = row(SV_P), *G = row(SV_G), *MI = row(SV_MINV);
    // a statement of the shared scalar logic that needs the whole state: assembled from LDS + registers, run identically by
    // every thread, written back by one (all threads take this path together: the state is uniform)
    auto with_full_state = [&](auto fn) {
        if constexpr (WPE == 2) return fn(s);
        else {
            ChainState full;
            chain_state_copy(full, *cold);
            s.to(full);
            const int r = fn(full);
            s.from(full);
            __syncthreads();
            if (tid == 0) chain_state_copy(*cold, full);
            __syncthreads();
            return r;
        }
    };

    if (!cold->kicked) {
        const int ph = s.phase;
        const double e = ph == PH_EPS ? s.eps : (ph == PH_TREE ? s.dir * s.eps : 0.0);
        if ((ph == PH_INIT || ph == PH_EPS || ph == PH_TREE) && own) {
            const double p = Pm[j] + 0.5 * e * G[j];
            Pm[j] = p;
            TH[j] += e * MI[j] * p;
        }
        __syncthreads();
        if (tid == 0) cold->kicked = 1;
        __syncthreads();
    }
    unsigned long long my_leaps = 0;

    for (int round = 0; round < a.rounds; ++round) {
        const int ph0 = s.phase;
        const bool act = ph0 == PH_INIT || ph0 == PH_EPS || ph0 == PH_TREE;
        if (!act) break;
        const double e = ph0 == PH_EPS ? s.eps : (ph0 == PH_TREE ? s.dir * s.eps : 0.0);

        // ---- B: log-posterior + gradient at the new point ------------------------------------------------------------------
        long long *prof = (PROF && a.prof) ? a.prof + (size_t)unit * 32 : nullptr;
        // the uniform that decides whether this leaf replaces the subtree's proposal depends on (leaf, depth, iteration)
        // only: wave 7, which has no per-element work in the first phases of the evaluation, draws it now (Philox, ~200
        // integer instructions) and publishes it through LDS, off the other waves' critical path
        if (ph0 == PH_TREE && wave == SOLO_NW - 1) {
            const double u = rng_uniform(rng, (uint32_t)s.leaf, RNG_LEAF, (uint32_t)s.depth, 0, (uint32_t)s.iter);
            if (lane == 0) lps_l[1] = u;
        }
        solo_eval<WPE == 2 ? 16 : 8>(P, g, smem, TH, G, lps_l, er, 1, tid, prof);
        long long tsp = (prof && tid == 0) ? clock64() : 0;
#define BDRT_SOLO_NPROF(slot) do { if (prof && tid == 0) { const long long t_ = clock64(); prof[slot] += t_ - tsp; tsp = t_; } } while (0)

        // ---- C: second half kick, kinetic energy, finiteness of the gradient ----------------------------------------------------
        double p = 0.0, gj = 0.0, mi = 1.0;
        double kin = 0.0, nonfin = 0.0;
        if (own) {
            gj = G[j]; mi = MI[j];
            p = Pm[j] + 0.5 * e * gj;
            kin = mi * p * p;
            nonfin = isfinite(gj) ? 0.0 : 1.0;
        }
        // (measured and dropped: the U-turn products of the first three merge levels -- known from the leaf index -- in this same
        // block reduction: stages C + D 3.2 k -> 2.9 k cycles, the evaluation + 0.5 k from the registers it takes; 6.12 -> 6.32 us)
        solo_block_sum2(kin, nonfin, red, slot, wave, lane);
        kin *= 0.5;
        BDRT_SOLO_NPROF(5);

        // ---- S1: scalar logic after the evaluation (identical in every thread) ---------------------------------------------------
        bool copyq = false, cur2s = false, tree = false, last = false;
        bool upds = false, welf = false, wend = false;
        int nm = 0, endt = 0, next = 0, draw = -1;
        double wn = 0.0;
        const int dir_now = s.dir;
        const int leaf_now = s.leaf;
        {
            const double lp = *lps_l;
            const bool finite_pt = isfinite(lp) && nonfin == 0.0;
            if (ph0 == PH_INIT) {
                if (finite_pt) {
                    s.lps = lp;
                    cur2s = true;
                    s.phase = PH_EPS; s.eps_dir = 0; s.eps_trials = 0;
                    next = 3;
                } else {
                    const int att = s.init_attempt + 1;
                    s.init_attempt = att;
                    if (att >= 100) s.phase = PH_FAILED;
                    else next = 4;
                }
            } else if (ph0 == PH_EPS) {
                // Stan base_hmc::init_stepsize
                my_leaps += 1;
                next = with_full_state([&](ChainState &f) { return nuts_stepsize_trial(f, np, lp, kin); });
            } else {   // PH_TREE: one new leaf
                my_leaps += 1;
                s.n_leap_iter = s.n_leap_iter + 1;
                double h = -lp + kin;
                if (isnan(h)) h = INFINITY;
                const double H0 = s.H0;
                const bool divergent = (h - H0) > np.max_deltaH;
                const double w = H0 - h;
                s.sum_metro = s.sum_metro + (w > 0.0 ? 1.0 : BDRT_NUTS_EXP(fmax(w, -746.0)));       // (w = -inf on a non-finite energy)
                if (divergent) {
                    endt = 2;
                } else {
                    const double u = lps_l[1];                          // drawn by wave 7 before the evaluation
                    double lsw_new;
                    const bool joins = nuts_leaf_joins(s.lsw_sub, w, u, lsw_new);           // (one exponential: bdrt_nuts_device.h)
                    if (leaf_now == 0 || joins) { copyq = true; s.lpq = lp; }
                    s.lsw_sub = lsw_new;
                    tree = true;
                    while ((leaf_now >> nm) & 1) ++nm;
                    last = leaf_now == s.nleaves - 1;
                }
            }
        }

        BDRT_SOLO_NPROF(6);
        // ---- D: proposal copy, checkpoints, U-turn tests, subtree close ----------------------------------------------------------
        if ((copyq || cur2s) && own) {
            const double th = TH[j];
            if (copyq) { row(SV_THQ)[j] = th; row(SV_GQ)[j] = gj; }
            if (cur2s) { row(SV_THS)[j] = th; row(SV_GS)[j] = gj; }
        }
        if (tree) {
            // binary-counter bookkeeping of the new subtree: see nuts_kernel (level l: rho / first momentum of the completed
            // left sub-subtree of 2^l leaves that waits for its sibling; level 0 keeps only the momentum)
            double rc = p, cpl = p;
            bool ok = true;
            for (int l = 0; l < nm; ++l) {
                double a0 = 0.0, a1 = 0.0;
                if (own) {
                    const double lpv = row(SV_CKP + l)[j];
                    const double lr = l == 0 ? lpv : row(SV_CKC + l)[j];
                    const double rho = lr + rc;
                    a0 = mi * lpv * rho;
                    a1 = mi * p * rho;
                    rc = rho;
                    cpl = lpv;
                }
                solo_block_sum2(a0, a1, red, slot, wave, lane);
                ok = ok && (a0 > 0.0) && (a1 > 0.0);
            }
            if (ok && !last && own) {
                row(SV_CKP + nm)[j] = cpl;
                if (nm > 0) row(SV_CKC + nm)[j] = rc;
            }
            if (!ok) {
                endt = 1;
            } else if (last) {
                double t0 = 0.0, t1 = 0.0;
                if (own) {
                    const double po = row(dir_now > 0 ? SV_PM : SV_PP)[j];     // momentum at the other end
                    const double rt = row(SV_RHO)[j] + rc;
                    row(SV_RHO)[j] = rt;
                    row(dir_now > 0 ? SV_THP : SV_THM)[j] = TH[j];
                    row(dir_now > 0 ? SV_PP : SV_PM)[j] = p;
                    row(dir_now > 0 ? SV_GP : SV_GM)[j] = gj;
                    t0 = mi * po * rt;
                    t1 = mi * p * rt;
                }
                solo_block_sum2(t0, t1, red, slot, wave, lane);
                const int depth = s.depth + 1;
                s.depth = depth;
                const double lsw = s.lsw, lsw_sub = s.lsw_sub;
                bool take;
                if (lsw_sub > lsw) take = true;
                else take = rng_uniform(rng, 0, RNG_TOP, (uint32_t)depth, 0, (uint32_t)s.iter) < BDRT_NUTS_EXP(lsw_sub - lsw);
                if (take) { upds = true; s.lps = s.lpq; }
                s.lsw = log_sum_exp2(lsw, lsw_sub);
                const bool keep_going = (t0 > 0.0) && (t1 > 0.0);
                if (!keep_going || depth >= np.max_depth) endt = 1;
                else {
                    s.dir = rng_uniform(rng, 0, RNG_DIRECTION, (uint32_t)depth, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                    s.leaf = 0; s.nleaves = 1 << depth; s.lsw_sub = -INFINITY;
                    next = 2;
                }
            } else {
                s.leaf = leaf_now + 1;
            }
        }
        if (endt) {
            next = with_full_state([&](ChainState &f) { return nuts_transition_end(f, np, endt, draw, welf, wend, wn); });     // (bdrt_nuts_device.h)
            if (draw >= 0 && a.lp_draws && tid == 0) a.lp_draws[(size_t)unit * np.n_draws + draw] = s.lps;
        }

        BDRT_SOLO_NPROF(7);
        // ---- A': the trajectory continues from the point just evaluated: half kick + drift of the next leapfrog ---------------------
        if (next == 0 && s.phase == PH_TREE) {
            const double e1 = s.dir * s.eps;
            if (own) {
                const double pk = p + 0.5 * e1 * gj;
                Pm[j] = pk;
                TH[j] = TH[j] + e1 * mi * pk;
            }
        }
        // ---- E: sample update, metric adaptation, draw output, start of the next leapfrog when the trajectory does not simply
        //      continue (new transition, next doubling, step-size search, re-initialisation) ------------------------------------------
        if (upds || welf || wend || draw >= 0 || next) {
            const uint32_t iter = (uint32_t)s.iter, trial = (uint32_t)s.eps_trials, att = (uint32_t)s.init_attempt;
            double ths = 0.0, gs = 0.0;
            if (own && (upds || welf || wend || draw >= 0 || next == 1 || next == 3)) {
                ths = row(upds ? SV_THQ : SV_THS)[j]; gs = row(upds ? SV_GQ : SV_GS)[j];
            }
            if (upds && own) { row(SV_THS)[j] = ths; row(SV_GS)[j] = gs; }
            if ((welf || wend) && own) {
                double *WM = Vg + (size_t)SG_WMEAN * DS, *W2 = Vg + (size_t)SG_WM2 * DS;
                double mean = WM[j], m2 = W2[j];
                if (welf) {            // Welford (stan::math::welford_var_estimator)
                    const double delta = ths - mean;
                    mean += delta / wn;
                    m2 += (ths - mean) * delta;
                }
                if (wend) {            // var_adaptation::learn_variance
                    const double var = wn > 1.0 ? m2 / (wn - 1.0) : 0.0;
                    mi = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
                    MI[j] = mi;
                    mean = 0.0; m2 = 0.0;
                }
                WM[j] = mean; W2[j] = m2;
            }
            if (draw >= 0 && own) a.draws[((size_t)unit * np.n_draws + draw) * D + j] = ths;
            if (next == 1 || next == 3) {
                // fresh momentum p ~ N(0, M): normals 2i, 2i+1 from one Philox block (same streams as nuts_kernel)
                if (2 * tid < D) {
                    double z0, z1;
                    rng_normal_pair(rng, (uint32_t)tid, next == 1 ? RNG_MOMENTUM : RNG_EPS_MOMENTUM, next == 1 ? 0u : trial, iter, z0, z1);
                    zrow[2 * tid] = z0; zrow[2 * tid + 1] = z1;
                }
                __syncthreads();
                double pn = 0.0, kin0 = 0.0, dummy = 0.0;
                if (own) { pn = zrow[j] / sqrt(mi); kin0 = mi * pn * pn; }
                solo_block_sum2(kin0, dummy, red, slot, wave, lane);
                s.H0 = -s.lps + 0.5 * kin0;
                if (next == 1) {
                    s.lsw = 0.0; s.lsw_sub = -INFINITY; s.depth = 0; s.leaf = 0; s.nleaves = 1;
                    s.n_leap_iter = 0; s.sum_metro = 0.0;
                    s.dir = rng_uniform(rng, 0, RNG_DIRECTION, 0, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                }
                const double e1 = next == 1 ? s.dir * s.eps : s.eps;
                if (own) {
                    if (next == 1) {
                        row(SV_THM)[j] = ths; row(SV_THP)[j] = ths;
                        row(SV_PM)[j] = pn; row(SV_PP)[j] = pn;
                        row(SV_GM)[j] = gs; row(SV_GP)[j] = gs;
                        row(SV_RHO)[j] = pn;
                    }
                    const double pk = pn + 0.5 * e1 * gs;
                    Pm[j] = pk;
                    TH[j] = ths + e1 * mi * pk;
                }
            } else if (next == 2) {
                // continue from the trajectory end in the new direction
                const int dir = s.dir;
                const double e1 = dir * s.eps;
                if (own) {
                    double et, ep, eg;
                    if (dir == dir_now) { et = TH[j]; ep = p; eg = gj; }
                    else { et = row(dir > 0 ? SV_THP : SV_THM)[j]; ep = row(dir > 0 ? SV_PP : SV_PM)[j]; eg = row(dir > 0 ? SV_GP : SV_GM)[j]; }
                    const double pk = ep + 0.5 * e1 * eg;
                    Pm[j] = pk;
                    TH[j] = et + e1 * mi * pk;
                }
            } else if (next == 4) {
                if (own) {
                    TH[j] = np.init_radius * (2.0 * rng_uniform(rng, (uint32_t)j, RNG_INIT, 0, att, 0) - 1.0);
                    Pm[j] = 0.0;
                }
            }
        }
        __syncthreads();                                   // theta / momentum rows complete before the next evaluation
        BDRT_SOLO_NPROF(8);
#undef BDRT_SOLO_NPROF
    }

    // ---- write the chain back ---------------------------------------------------------------------------------------------------
    __syncthreads();
    for (int v = 0; v < SV_COUNT; ++v)
        if (own && !(TRIM && solo_hot_slot(v) < 0)) Vg[(size_t)v * DS + j] = row(v)[j];
    if (tid == 0) {
        ChainState full;
        chain_state_copy(full, *cold);
        if constexpr (WPE == 2) { const int k = full.kicked; full = s; full.kicked = k; } else s.to(full);
        chain_state_copy(a.states[unit], full);
        if (my_leaps) atomicAdd(a.leap_counter, my_leaps);
        const int ph = s.phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) atomicAdd(a.done_counter, 1);
    }
}

// evaluator of the solo path on its own (few-point batches of bdrt_logp_grad, parity tests): a workgroup per point -- or, with
// fewer workgroups than points, a grid-stride loop over the points --, theta / grad [B x D] in global memory.  Needs only the
// evaluator's share of the LDS (solo_eval_lds_bytes), so that three workgroups fit a CU.
__host__ __device__ inline size_t solo_eval_lds_bytes(const SoloGeom &g) { return ((size_t)g.o_vec + 2 * (size_t)g.DSS) * sizeof(double) + 64; }

// one point per workgroup, nothing else: the form that measured 8.0 us per launch at B = 1 (the grid-stride form below: 9.5)
__global__ __launch_bounds__(SOLO_NT) void solo_eval_one_kernel(const DevProblem *__restrict__ Pp, SoloGeom g, const double *theta,
                                                                const int *spec, int jacobian, double *lp, double *grad)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x, b = blockIdx.x;
    double *TH = smem + g.o_vec, *GR = TH + g.DSS, *lps = smem + g.o_scv + 12;
    if (tid < g.DSS) { TH[tid] = tid < g.D ? theta[(size_t)b * g.D + tid] : 0.0; GR[tid] = 0.0; }
    solo_eval_init(P, g, smem, tid);
    const SoloEvalRegs er = solo_eval_setup(P, g, spec ? spec[b] : 0, tid);
    __syncthreads();
    solo_eval(P, g, smem, TH, GR, lps, er, jacobian, tid);
    if (tid < g.D && grad) grad[(size_t)b * g.D + tid] = GR[tid];
    if (tid == 0 && lp) lp[b] = *lps;
}

// TIGHT: 80 VGPRs (six waves per SIMD: three workgroups share a CU) for batches of more points than CUs
template <bool TIGHT>
__global__ __launch_bounds__(SOLO_NT, TIGHT ? 6 : 2) void solo_eval_kernel(const DevProblem *__restrict__ Pp, SoloGeom g, const double *theta,
                                                                          const int *spec, int B, int jacobian, double *lp, double *grad)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x;
    double *TH = smem + g.o_vec, *GR = TH + g.DSS, *lps = smem + g.o_scv + 12;
    solo_eval_init(P, g, smem, tid);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        if (tid < g.DSS) { TH[tid] = tid < g.D ? theta[(size_t)b * g.D + tid] : 0.0; GR[tid] = 0.0; }
        const SoloEvalRegs er = solo_eval_setup(P, g, spec ? spec[b] : 0, tid);
        __syncthreads();
        solo_eval<TIGHT ? 8 : 16>(P, g, smem, TH, GR, lps, er, jacobian, tid);
        if (tid < g.D && grad) grad[(size_t)b * g.D + tid] = GR[tid];
        if (tid == 0 && lp) lp[b] = *lps;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One chain per workgroup, general block model (bdrt_solo_wide.h): the evaluation by 512 threads, everything after it by the
// cooperative stage of bdrt_nuts_wide.h.  Global state layout: vecs [n_units][V_COUNT][ds] (the rows of nuts_kernel, one column).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int W1_SCRATCH = 1600;                   // doubles of LDS for the cooperative stage (reductions, momentum normals)
// rows of the chain kept in LDS for the launch, most used first (as many as fit: `nhot`): a plain leaf then touches HBM only
// for the trajectory ends / higher checkpoint levels it rarely needs
__device__ __constant__ signed char W1_HOT_ORDER[12] = {V_TH, V_P, V_G, V_MINV, V_CKP, V_THQ, V_GQ, V_CKP + 1, V_CKC + 1, V_CKP + 2, V_CKC + 2, V_RHO};
constexpr int W1_HOT_MAX = 12;

__host__ __device__ inline size_t wide1_lds_bytes(const Wide1Geom &G, int ds, int nhot)
{
    return ((size_t)G.total + W1_SCRATCH + 2 + (size_t)nhot * ds) * sizeof(double) + sizeof(ChainState) + 64 + 64;
}

static_assert((W1_SCRATCH + 2) * sizeof(double) + sizeof(ChainState) + 128 <= 16384, "wide1_capable (bdrt_solo_wide.h) leaves 16 KiB beside the evaluator");

template <bool PROF = false>
__global__ __launch_bounds__(SOLO_NT) void nuts_wide1_kernel(const DevProblem *__restrict__ Pp, NutsParams np, NutsArgs a, Wide1Geom G, int nhot)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x;
    const int wg = blockIdx.x, unit = a.unit_map ? a.unit_map[wg] : wg;
    const int D = P.D, DS = a.ds;
    double *scr = smem + G.total;
    double *lpn = scr + W1_SCRATCH;
    double *hot = lpn + 2;                                 // [nhot][ds]
    ChainState *sts = reinterpret_cast<ChainState *>(hot + (size_t)nhot * DS);
    signed char *hslot = reinterpret_cast<signed char *>(sts + 1);      // [V_COUNT] (64 bytes)
    double *V = a.vecs + (size_t)wg * V_COUNT * DS;       // this chain's rows [V_COUNT][ds]
    auto grow = [&](int v) -> double * { return V + (size_t)v * DS; };                                    // the row in HBM
    auto row = [&](int v) -> double * { const int h = hslot[v]; return h >= 0 ? hot + (size_t)h * DS : grow(v); };
    if (tid == 0) chain_state_copy(sts[0], a.states[unit]);
    if (tid < V_COUNT) {
        int h = -1;
        for (int k = 0; k < nhot; ++k) if (W1_HOT_ORDER[k] == tid) h = k;
        hslot[tid] = (signed char)h;
    }
    wide1_init(P, G, smem, tid);
    for (int k = 0; k < nhot; ++k) {
        const double *src = grow(W1_HOT_ORDER[k]);
        for (int j = tid; j < DS; j += SOLO_NT) hot[(size_t)k * DS + j] = src[j];
    }
    __syncthreads();
    const Wide1Regs er = wide1_setup(P, G, sts[0].spec, tid);
    if (!sts[0].kicked) {
        // half kick + drift of the first evaluation of a freshly created sampler
        const int ph = sts[0].phase;
        const double e = ph == PH_EPS ? sts[0].eps : (ph == PH_TREE ? sts[0].dir * sts[0].eps : 0.0);
        double *TH = row(V_TH), *Pm = row(V_P), *Gr = row(V_G), *MI = row(V_MINV);
        if (ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)
            for (int j = tid; j < D; j += SOLO_NT) {
                const double p = Pm[j] + 0.5 * e * Gr[j];
                Pm[j] = p;
                TH[j] += e * MI[j] * p;
            }
        __syncthreads();
        if (tid == 0) sts[0].kicked = 1;
        __syncthreads();
    }
    WideCtx wx;
    wx.P = Pp; wx.np = &np; wx.a = &a; wx.V = V; wx.smem = scr; wx.sts = sts; wx.lpn = lpn; wx.hvy = nullptr; wx.hvk = nullptr;
    wx.prof = nullptr; wx.D = D; wx.DS = DS; wx.TH2OFF = 0; wx.c0 = unit; wx.nvalid = 1; wx.slot_unit = nullptr; wx.ncol = 1;
    wx.hot_base = hot; wx.hot_slot = hslot;
    unsigned long long my_leaps = 0;
    for (int round = 0; round < a.rounds; ++round) {
        const int ph = sts[0].phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) break;
        long long *prof = (PROF && a.prof) ? a.prof + (size_t)wg * 32 : nullptr;       // slots 0 / 1: evaluation / everything after it (thread 0)
        const long long t0 = (prof && tid == 0) ? clock64() : 0;
        wide1_eval(P, G, smem, row(V_TH), row(V_G), lpn, er, 1, tid, prof);
        const long long t1 = (prof && tid == 0) ? clock64() : 0;
        wx.prof = prof ? prof + 16 : nullptr;                                // (slots 25, 27..31: the stages of the cooperative tail)
        wide_coop_tail<2, true>(wx, 0, false, my_leaps, tid);
        __syncthreads();
        if (prof && tid == 0) { prof[0] += t1 - t0; prof[1] += clock64() - t1; }
    }
    // the LDS-resident rows go back to the chain's HBM rows
    for (int k = 0; k < nhot; ++k) {
        double *dst = grow(W1_HOT_ORDER[k]);
        for (int j = tid; j < DS; j += SOLO_NT) dst[j] = hot[(size_t)k * DS + j];
    }
    if (tid == 0) {
        chain_state_copy(a.states[unit], sts[0]);
        if (my_leaps) atomicAdd(a.leap_counter, my_leaps);
        const int ph = sts[0].phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) atomicAdd(a.done_counter, 1);
    }
}

// the general one-chain evaluator (bdrt_solo_wide.h) on its own: one point per workgroup (tests)
// ---------------------------------------------------------------------------------------------------------------------------
// Problems beyond the LDS budget (bdrt_big.h): one chain per workgroup, every row of the chain in HBM, the evaluation by the
// streamed evaluator (workspace in HBM), everything after it by the cooperative stage of bdrt_nuts_wide.h.  Slow but working:
// the reference accepts any grid (inversion.py:2127-2209).  State layout: that of nuts_wide1_kernel.
// ---------------------------------------------------------------------------------------------------------------------------
// NJX: elements of a parameter vector per thread of the cooperative stage, D <= 512 NJX (2: D <= 1024; 4: D <= 2048 -- three
// distributions of 301 basis functions are 1821 parameters; 8 and 16: D <= 4096 / 8192, one distribution of 1200 basis functions
// is 2409 -- these two keep most of the stage's rows in scratch memory: the streamed path is the slow path either way).  The
// stage's LDS scratch holds the D momentum normals behind its 512 doubles of reduction scratch.
constexpr int BIG_MAX_D = 8192;
__host__ __device__ inline int big_njx(int D) { return D <= 1024 ? 2 : (D <= 2048 ? 4 : (D <= 4096 ? 8 : 16)); }
__host__ __device__ inline int big_scratch_doubles(int njx) { return njx <= 2 ? W1_SCRATCH : 512 + 512 * njx + 64; }
__host__ __device__ inline size_t nuts_big_lds_bytes(int njx = 2) { return (size_t)(big_scratch_doubles(njx) + 2 + 9 * 8) * sizeof(double) + sizeof(ChainState) + 128; }

template <int NJX>
__global__ __launch_bounds__(SOLO_NT) void nuts_big_kernel(const DevProblem *__restrict__ Pp, NutsParams np, NutsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x;
    const int wg = blockIdx.x, unit = a.unit_map ? a.unit_map[wg] : wg;
    const int D = P.D, DS = a.ds;
    double *scr = smem;
    double *lpn = scr + big_scratch_doubles(NJX);
    double *red = lpn + 2;
    ChainState *sts = reinterpret_cast<ChainState *>(red + 9 * 8);
    signed char *hslot = reinterpret_cast<signed char *>(sts + 1);      // [V_COUNT] (64 bytes): no row is LDS-resident
    double *V = a.vecs + (size_t)wg * V_COUNT * DS;       // this chain's rows [V_COUNT][ds]
    double *ws = a.bigws + (size_t)wg * big_ws_doubles(P);
    auto row = [&](int v) -> double * { return V + (size_t)v * DS; };
    if (tid == 0) chain_state_copy(sts[0], a.states[unit]);
    if (tid < V_COUNT) hslot[tid] = (signed char)-1;
    __syncthreads();
    const int spec = sts[0].spec;
    if (!sts[0].kicked) {
        const int ph = sts[0].phase;
        const double e = ph == PH_EPS ? sts[0].eps : (ph == PH_TREE ? sts[0].dir * sts[0].eps : 0.0);
        double *TH = row(V_TH), *Pm = row(V_P), *Gr = row(V_G), *MI = row(V_MINV);
        if (ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)
            for (int j = tid; j < D; j += SOLO_NT) {
                const double p = Pm[j] + 0.5 * e * Gr[j];
                Pm[j] = p;
                TH[j] += e * MI[j] * p;
            }
        __syncthreads();
        if (tid == 0) sts[0].kicked = 1;
        __syncthreads();
    }
    WideCtx wx;
    wx.P = Pp; wx.np = &np; wx.a = &a; wx.V = V; wx.smem = scr; wx.sts = sts; wx.lpn = lpn; wx.hvy = nullptr; wx.hvk = nullptr;
    wx.prof = nullptr; wx.D = D; wx.DS = DS; wx.TH2OFF = 0; wx.c0 = unit; wx.nvalid = 1; wx.slot_unit = nullptr; wx.ncol = 1;
    wx.hot_base = scr; wx.hot_slot = hslot;
    unsigned long long my_leaps = 0;
    for (int round = 0; round < a.rounds; ++round) {
        const int ph = sts[0].phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) break;
        // the rows of the last cooperative stage (global stores of other threads) are read by this evaluation
        __threadfence_block();
        __syncthreads();
        big_eval(P, ws, row(V_TH), row(V_G), lpn, spec, 1, red, tid);
        __threadfence_block();
        __syncthreads();
        wide_coop_tail<NJX, true>(wx, 0, false, my_leaps, tid);
        __syncthreads();
    }
    if (tid == 0) {
        chain_state_copy(a.states[unit], sts[0]);
        if (my_leaps) atomicAdd(a.leap_counter, my_leaps);
        const int ph = sts[0].phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) atomicAdd(a.done_counter, 1);
    }
}

__global__ __launch_bounds__(SOLO_NT) void wide1_eval_kernel(const DevProblem *__restrict__ Pp, Wide1Geom G, const double *theta,
                                                             const int *spec, int B, int jacobian, double *lp, double *grad)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x;
    wide1_init(P, G, smem, tid);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {          // (a workgroup per point, or a grid-stride loop over the points)
        const Wide1Regs er = wide1_setup(P, G, spec ? spec[b] : 0, tid);
        __syncthreads();
        wide1_eval(P, G, smem, theta + (size_t)b * P.D, grad + (size_t)b * P.D, lp + b, er, jacobian, tid);
    }
}


// The tail of a large run.  The 16-chain kernel advances every live chain by one leapfrog per ~33 us whatever the number of
// live chains; a run lasts as long as its longest chain (BASELINE config 4: 0.33 .. 0.98 M leapfrogs per chain), and in the
// tail most tile columns are empty.  Once the number of live chains is below what the one-chain-per-workgroup kernel
// finishes faster (8.3 us per leapfrog, one chain per CU at a time: below ~4 chains per CU), the live chains move there:
// this kernel copies a chain's rows from the 16-chain layout [wg][V_*][column][ds16] to [slot][SV_* / SG_*][dss].
// Same counter-based random numbers and the same arithmetic up to summation order, so the chains continue as they were.
__global__ void nuts_migrate_kernel(const double *v16, int ds16, const int *unit_loc, const int *unit_map, double *vsolo, int dss, int D)
{
    const int slot = blockIdx.x, u = unit_map[slot];
    const int wg = unit_loc[u] >> 4, col = slot_col(unit_loc[u] & 15);      // where the unit sits in the 16-chain layout
    const double *src = v16 + (size_t)wg * V_COUNT * NC * ds16;
    double *dst = vsolo + (size_t)slot * SG_COUNT * dss;
    for (int r = 0; r < SG_COUNT; ++r) {
        int v;                                            // row of the 16-chain layout that holds row r of the solo layout
        if (r < SV_CKC) { constexpr int head[SV_CKC] = {V_TH, V_P, V_G, V_THM, V_PM, V_GM, V_THP, V_PP, V_GP, V_THS, V_GS, V_THQ, V_GQ, V_RHO, V_MINV}; v = head[r]; }
        else if (r < SV_CKP) v = V_CKC + (r - SV_CKC);
        else if (r < SV_COUNT) v = V_CKP + (r - SV_CKP);
        else v = r == SG_WMEAN ? V_WMEAN : V_WM2;
        const double *sr = src + ((size_t)v * NC + col) * ds16;
        for (int j = threadIdx.x; j < dss; j += blockDim.x) dst[(size_t)r * dss + j] = j < D ? sr[j] : 0.0;
    }
}
static_assert(SOLO_MAXD == MAXD, "the two kernels keep the same number of checkpoint levels");

// the same hand-over for the models of the general one-chain kernel (bdrt_solo_wide.h): rows keep their meaning, the chain's
// column of [wg][V_*][16][ds] becomes [slot][V_*][ds]
__global__ void nuts_migrate_wide1_kernel(const double *v16, int ds, const int *unit_loc, const int *unit_map, double *v1, ChainState *states)
{
    const int slot = blockIdx.x, u = unit_map[slot];
    const int wg = unit_loc[u] >> 4, col = slot_col(unit_loc[u] & 15);
    const double *src = v16 + (size_t)wg * V_COUNT * NC * ds;
    double *dst = v1 + (size_t)slot * V_COUNT * ds;
    const int live = states[u].thsel ? V_TH2 : V_TH;       // the one-chain kernel keeps theta in V_TH
    for (int v = 0; v < V_COUNT; ++v) {
        const int sv = v == V_TH ? live : v;
        for (int j = threadIdx.x; j < ds; j += blockDim.x) dst[(size_t)v * ds + j] = src[((size_t)sv * NC + col) * ds + j];
    }
    __syncthreads();
    if (threadIdx.x == 0) states[u].thsel = 0;
}

// Compaction of a large run.  A finished chain leaves its column of the 16-column MFMA tile empty, and a workgroup costs the
// same ~33 us per round however few of its columns are live.  While there are more workgroups than CUs (more than 16 live
// chains per CU), the live chains are therefore re-packed into fewer, full workgroups from time to time: workgroup `blockIdx.x`
// of the NEW layout gathers the rows of its up to 16 units from wherever they sat in the old one.  Rows are copied verbatim and
// every random number is keyed by (seed, chain id, iteration, ...), never by the slot, so the chains continue bit for bit
// (tests/test_gpu_config4.py).  Empty slots get the finite placeholders of a fresh sampler (inverse metric 1, zeros elsewhere).
__global__ __launch_bounds__(256) void nuts_compact_kernel(const double *vold, const int *old_loc, const int *new_slot_unit,
                                                           double *vnew, int ds)
{
    const int wg = blockIdx.x;
    const size_t rowlen = (size_t)ds;
    for (int k = 0; k < NC; ++k) {
        const int u = new_slot_unit[wg * NC + k];
        const int col = slot_col(k);
        double *dst = vnew + (size_t)wg * V_COUNT * NC * rowlen;
        if (u < 0) {
            for (int v = 0; v < V_COUNT; ++v)
                for (int j = threadIdx.x; j < ds; j += blockDim.x) dst[((size_t)v * NC + col) * rowlen + j] = v == V_MINV ? 1.0 : 0.0;
            continue;
        }
        const int owg = old_loc[u] >> 4, ocol = slot_col(old_loc[u] & 15);
        const double *src = vold + (size_t)owg * V_COUNT * NC * rowlen;
        for (int v = 0; v < V_COUNT; ++v)
            for (int j = threadIdx.x; j < ds; j += blockDim.x)
                dst[((size_t)v * NC + col) * rowlen + j] = src[((size_t)v * NC + ocol) * rowlen + j];
    }
}

// can two workgroups of the one-chain kernel share a CU for this problem (LDS of the trimmed variant)?
static bool solo_duo_fits(const DevProblem &P)
{
    if (const char *e = getenv("BDRT_SOLO_DUO")) { if (atoi(e) == 0) return false; }
    const SoloGeom g = solo_geometry(P.nf, P.blk[0].K, P.D);
    return 2 * (((size_t)g.o_vec + (size_t)SOLO_NHOT * g.DSS) * sizeof(double) + 64) <= 160 * 1024;
}

// Which kernel advances the one-chain layout (state rows [unit][SG_COUNT][ds]) while `live` chains are running: the one-chain-per-
// wave kernel from more than two live chains per CU on (measured at 81 x 161, profiles/r05/kernel_sweep.txt: up to two per CU two
// 512-thread workgroups finish a round in 9.4 us; a third chain on any CU is a second turn for them, 14.6 us, against 13.1 us of
// the wave kernel), up to the eight per CU it keeps resident.  BDRT_WAVE=1 / 0: always / never.
static bool wave_pays(int live, int n_cu) { return live > 2 * n_cu; }
static int wave_max_units(int n_cu, const DevProblem &P) { return wave_chains_per_cu(P) * n_cu; }      // (a ninth chain on any CU is a second turn of the machine: 31 us per round instead of 19)

// liveness of every unit (1: the chain is still running), for the host's re-packing decision
__global__ void nuts_live_kernel(const ChainState *states, int n, int *live)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) { const int ph = states[u].phase; live[u] = (ph == PH_INIT || ph == PH_EPS || ph == PH_TREE) ? 1 : 0; }
}

// device allocation freed on scope exit
struct DevTmpBuf {
    void *p = nullptr;
    ~DevTmpBuf() { if (p) hipFree(p); }
};

struct Sampler {
    Problem *prob = nullptr;
    NutsParams np;
    NutsArgs args;
    int n_units = 0, n_wg = 0, D = 0;
    size_t lds_bytes = 0;
    bool use_s1 = false;     // S1 evaluator with theta rows resident in LDS (MODE 2)
    bool s1_hbm = false;     // S1 evaluator, sampler state in HBM (MODE 3: outlier parameters, K near 192)
    bool hw = false;         // general half-wave evaluator (MODE 4: several distributions, parallel blocks)
    bool solo = false;       // one chain per workgroup, state in LDS (bdrt_solo.h): few chains of the headline family
    SoloGeom geom;
    bool wave = false;       // the one-chain layout may be advanced by the one-chain-per-WAVE kernel (bdrt_wave.h: same state layout)
    int wave_force = -1;     // BDRT_WAVE: 1 always, 0 never (-1: by the number of live chains, wave_pays)
    bool solo_ok = false;    // the 512-thread one-chain kernels take this problem (else the wave kernel advances the layout whatever `live`)
    bool wave_last = false;  // the last launch used the wave kernel (bdrt_sampler_kind)
    int live = 0;            // chains of the one-chain layout still running (after the last launch that read the done counter)
    WaveGeom geomw;
    bool wide1 = false;      // one chain per workgroup, general block model (bdrt_solo_wide.h): few chains of any other Toeplitz family
    Wide1Geom geom1;
    int nhot1 = 0;           // rows of the chain that kernel keeps in LDS
    bool big = false;        // problem beyond the LDS budget (bdrt_big.h): the wide1 layout advanced by nuts_big_kernel
    double *d_bigws = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms_total = 0.0;
    int64_t n_launch = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int *d_done = nullptr;
    unsigned long long *d_leaps = nullptr;
    int rounds_default = 256;
    long long *d_prof = nullptr;
    int prof_wg = 0;                  // workgroups d_prof was allocated for (the layout can change under it: compaction, tail migration)
    // tail migration (nuts_migrate_kernel)
    int *d_active = nullptr;          // live chains after the last launch of the 16-chain kernel
    int n_cu = 256;
    bool may_migrate = false, migrated = false;
    double *vecs16 = nullptr;         // the 16-chain rows, kept until the sampler is destroyed
    int *d_unit_map = nullptr;
    int n_solo = 0;                   // workgroups of the one-chain-per-workgroup kernel (= n_units unless migrated)
    // unit <-> slot of the 16-chain kernel (compaction: nuts_compact_kernel)
    std::vector<int> slot_unit;       // host copy of args.slot_unit: [n_wg][16]
    std::vector<int> unit_loc;        // unit -> wg * 16 + slot (-1: retired: the chain had finished when its workgroup was re-packed)
    int *d_slot_unit = nullptr, *d_unit_loc = nullptr;
    size_t vecs_capacity = 0;         // doubles allocated behind args.vecs
    double *vecs_alt = nullptr;       // second buffer of the same size: re-packing ping-pongs between the two (no allocation,
    int *d_slot_alt = nullptr;        //  hence no implicit device synchronisation, per pass)
    int *d_live = nullptr;
    bool may_compact = false;
    int n_compactions = 0;
};

// as many LDS-resident rows as fit beside the evaluator (160 KiB minus a margin)
static int wide1_hot_rows(const Wide1Geom &G, int ds)
{
    int n = W1_HOT_MAX;
    while (n > 0 && wide1_lds_bytes(G, ds, n) > 158 * 1024) --n;
    return n;
}

// what the 16-chain kernel keeps beside the tile region: lp / hand-over cells, chain states, spectrum ids / offsets / flags, and
// the uniforms of sixteen leaves per chain
constexpr size_t NUTS16_SCALAR_LDS = (size_t)3 * NC * sizeof(double) + NC * sizeof(ChainState) + 3 * NC * sizeof(int) + 16 + (size_t)NC * 16 * sizeof(double);
static_assert(NUTS16_SCALAR_LDS <= SAMPLER_LDS_RESERVE,
              "bdrt_problem_create reserves SAMPLER_LDS_RESERVE bytes for what the sampler keeps beside the tile region");
static size_t nuts_lds_bytes(const DevProblem &P, bool s1)
{
    const int nj = s1_nj(P.D);
    const size_t tile = s1 ? s1_lds_doubles(P) + (size_t)NC * 32 * nj : lds_doubles(P);   // s1: + theta rows
    return tile * sizeof(double) + NUTS16_SCALAR_LDS;
}

// any_b: the caller wants ONE evaluator whatever the batch size (the Newton iteration's trial points: a fit's numbers must not depend on
// how many other fits share its batch) -- the headline family's one-workgroup-per-point evaluator then takes any B (grid-stride), every
// other family answers 1 (the caller's tile evaluator, also for every B)
int launch_logp_grad_few(Problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp, double *d_grad,
                         hipStream_t stream, int any_b)
{
    Problem &P = *p;
    const char *sw = getenv("BDRT_FEW_POINTS");                     // diagnostics / tests: 0 = the tile evaluator whatever B is
    const bool off = sw && atoi(sw) == 0;
    if (off || B < 1 || !d_grad || !d_lp) return 1;
    if (P.few_kind < 0) {
        P.few_kind = solo_capable(P.dev) ? 1 : (wide1_capable(P.dev) ? 2 : 0);
        hipDeviceProp_t prop;
        P.few_ncu = (hipGetDeviceProperties(&prop, P.device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    if (P.few_kind == 0) return 1;
    const bool solo = P.few_kind == 1;
    const int ncu = P.few_ncu;
    BDRT_HIP(hipSetDevice(P.device));
    // a few points per CU: beyond that the 16-column tiles win (26 us up to 16 points per CU).  BDRT_FEW_POINTS=n: n points per CU
    // at most (default 5 for the LDS-light evaluator of the headline family, three workgroups of which share a CU; 1 otherwise)
    const int per_cu = sw ? atoi(sw) : (solo ? 5 : 1);
    if (any_b ? !solo : B > per_cu * ncu) return 1;
    static LdsAttrCache attr_solo, attr_w1;
    if (solo) {
        const SoloGeom g = solo_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D);
        // The LDS request doubles as a placement hint: the dispatcher packs as many workgroups on a CU as their resources allow
        // and leaves other CUs idle, so each workgroup asks for its share of a CU -- all of it while there are at most as many
        // points as CUs (measured: 8.0 us at B = 1 against 9.5 us when three fit), half or a third beyond
        const int wgs = B <= ncu ? B : std::min(B, 3 * ncu);
        const int share = std::min(3, (wgs + ncu - 1) / ncu);
        const size_t lds = std::max(solo_eval_lds_bytes(g), (size_t)(160 * 1024) / share - 2048);
        const size_t lds_max = (size_t)160 * 1024 - 2048;
        BDRT_HIP(attr_solo.ensure(lds_max, [&]() {
            hipError_t e = hipFuncSetAttribute((const void *)solo_eval_one_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)solo_eval_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            return e; }));
        if (B <= ncu)
            hipLaunchKernelGGL(solo_eval_one_kernel, dim3(wgs), dim3(SOLO_NT), (size_t)g.total * sizeof(double) + 64, stream, (const DevProblem *)P.d_dev, g,
                               d_theta, d_spec, jacobian, d_lp, d_grad);
        else
            hipLaunchKernelGGL(solo_eval_kernel<true>, dim3(wgs), dim3(SOLO_NT), lds, stream, (const DevProblem *)P.d_dev, g,
                               d_theta, d_spec, B, jacobian, d_lp, d_grad);
    } else {
        const Wide1Geom G = wide1_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D, P.dev.nblocks);
        const size_t lds = (size_t)G.total * sizeof(double) + 64;
        BDRT_HIP(attr_w1.ensure(lds, [&]() {
            return hipFuncSetAttribute((const void *)wide1_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
        const int wgs = std::min(B, 2 * ncu);
        hipLaunchKernelGGL(wide1_eval_kernel, dim3(wgs), dim3(SOLO_NT), lds, stream, (const DevProblem *)P.d_dev, G, d_theta, d_spec, B,
                           jacobian, d_lp, d_grad);
    }
    BDRT_HIP(hipGetLastError());
    return 0;
}

// The Stan-style L-BFGS of n fits as one launch (bdrt_lbfgs_dev.h).  Returns 1 when the problem takes neither one-chain
// evaluator (caller: host-driven path), 0 on success (x_out / g_out [n][D]: last iterate and the gradient of -lp there).
int lbfgs_device(Problem &P, const double *x0, const int *spec, int n, const bdrt_opt_options &o, double *x_out, double *g_out,
                 int *iters, int *n_evals, int *rc, double *f)
{
    const bool solo = solo_capable(P.dev), w1 = !solo && wide1_capable(P.dev);
    if ((!solo && !w1) || n < 1 || o.history > LBFGS_MAX_HISTORY) return 1;
    BDRT_HIP(hipSetDevice(P.device));
    const int D = P.dev.D;
    SoloGeom g = solo_geometry(P.dev.nf, P.dev.blk[0].K, D);
    Wide1Geom G = wide1_geometry(P.dev.nf, P.dev.blk[0].K, D, P.dev.nblocks);
    const int DS = (D + 7) & ~7;
    size_t lds;
    if (solo) {
        lds = ((size_t)g.o_vec + (size_t)(2 + 2 * LBFGS_MAX_HISTORY) * g.DSS) * sizeof(double) + 64;
        if (lds > 160 * 1024) return 1;
    } else {
        lds = ((size_t)G.total + 8) * sizeof(double) + 64;
    }
    DevTmpBuf dx0, dxo, dgo, drep, dspec, dwork;
    const size_t nb = (size_t)n * D * sizeof(double);
    BDRT_HIP(hipMalloc(&dx0.p, nb)); BDRT_HIP(hipMalloc(&dxo.p, nb)); BDRT_HIP(hipMalloc(&dgo.p, nb));
    BDRT_HIP(hipMalloc(&drep.p, (size_t)n * sizeof(LbfgsDevReport)));
    BDRT_HIP(hipMemcpy(dx0.p, x0, nb, hipMemcpyHostToDevice));
    if (spec) { BDRT_HIP(hipMalloc(&dspec.p, (size_t)n * sizeof(int))); BDRT_HIP(hipMemcpy(dspec.p, spec, (size_t)n * sizeof(int), hipMemcpyHostToDevice)); }
    if (!solo) BDRT_HIP(hipMalloc(&dwork.p, (size_t)n * LBFGS_WIDE_ROWS * DS * sizeof(double)));
    const long long cap = (long long)std::max(o.max_iter, 0) * 4 + 64;
    const int max_evals = (int)std::min<long long>(cap, 1LL << 30);
    static LdsAttrCache attr_s, attr_w;
    if (solo) {
        BDRT_HIP(attr_s.ensure(lds, [&]() { return hipFuncSetAttribute((const void *)lbfgs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
        hipLaunchKernelGGL(lbfgs_kernel<false>, dim3(n), dim3(SOLO_NT), lds, P.stream, (const DevProblem *)P.d_dev, g, G, (const double *)dx0.p,
                           (const int *)dspec.p, o, max_evals, (double *)dxo.p, (double *)dgo.p, (LbfgsDevReport *)drep.p, (double *)nullptr, DS);
    } else {
        BDRT_HIP(attr_w.ensure(lds, [&]() { return hipFuncSetAttribute((const void *)lbfgs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
        hipLaunchKernelGGL(lbfgs_kernel<true>, dim3(n), dim3(SOLO_NT), lds, P.stream, (const DevProblem *)P.d_dev, g, G, (const double *)dx0.p,
                           (const int *)dspec.p, o, max_evals, (double *)dxo.p, (double *)dgo.p, (LbfgsDevReport *)drep.p, (double *)dwork.p, DS);
    }
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipStreamSynchronize(P.stream));
    std::vector<LbfgsDevReport> reps((size_t)n);
    BDRT_HIP(hipMemcpy(reps.data(), drep.p, reps.size() * sizeof(LbfgsDevReport), hipMemcpyDeviceToHost));
    BDRT_HIP(hipMemcpy(x_out, dxo.p, nb, hipMemcpyDeviceToHost));
    BDRT_HIP(hipMemcpy(g_out, dgo.p, nb, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { iters[i] = reps[i].iters; n_evals[i] = reps[i].n_evals; rc[i] = reps[i].rc; f[i] = reps[i].f; }
    return 0;
}

}  // namespace bdrt

using namespace bdrt;

struct bdrt_sampler {
    bdrt::Sampler impl;
};

// test probe (tests/test_gpu_lean_math.py): a leaf's acceptance decision and new log-weight in the device form of nuts_leaf_joins (one
// exponential) and in the textbook form (log_sum_exp2, u < exp(w - lsw_new)) -- both with the device's lean exp / log
__global__ void leaf_joins_probe_kernel(const double *lsw_sub, const double *w, const double *u, int n, double *lsw_dev, int *join_dev,
                                        double *lsw_ref, int *join_ref, double *prob_ref)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double l;
    join_dev[i] = nuts_leaf_joins(lsw_sub[i], w[i], u[i], l) ? 1 : 0;
    lsw_dev[i] = l;
    const double lr = log_sum_exp2(lsw_sub[i], w[i]);
    const double pr = BDRT_NUTS_EXP(w[i] - lr);
    lsw_ref[i] = lr; prob_ref[i] = pr; join_ref[i] = u[i] < pr ? 1 : 0;
}

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
    if (S.d_prof) hipFree(S.d_prof);
    if (S.d_active) hipFree(S.d_active);
    if (S.d_bigws) hipFree(S.d_bigws);
    if (S.vecs16) hipFree(S.vecs16);
    if (S.d_unit_map) hipFree(S.d_unit_map);
    if (S.d_slot_unit) hipFree(S.d_slot_unit);
    if (S.d_unit_loc) hipFree(S.d_unit_loc);
    if (S.vecs_alt) hipFree(S.vecs_alt);
    if (S.d_slot_alt) hipFree(S.d_slot_alt);
    if (S.d_live) hipFree(S.d_live);
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
    // Stan's argument checks (stan::services: adapt delta in (0,1), gamma / kappa / t0 / stepsize > 0, init radius >= 0); written so
    // that NaN fails them.  Nonsense here does not crash a kernel, it silently gives nonsense chains.
    if (!(c.adapt_delta > 0.0 && c.adapt_delta < 1.0)) { set_error("bdrt_sampler_create: adapt_delta must be in (0,1)"); return nullptr; }
    if (!(c.adapt_gamma > 0.0) || !(c.adapt_kappa > 0.0) || !(c.adapt_t0 > 0.0)) { set_error("bdrt_sampler_create: adapt_gamma, adapt_kappa, adapt_t0 must be positive"); return nullptr; }
    if (!(c.stepsize0 > 0.0) || !std::isfinite(c.stepsize0)) { set_error("bdrt_sampler_create: stepsize0 must be positive and finite"); return nullptr; }
    if (!(c.init_radius >= 0.0) || !std::isfinite(c.init_radius)) { set_error("bdrt_sampler_create: init_radius must be >= 0 and finite"); return nullptr; }
    if (!(c.max_deltaH > 0.0)) { set_error("bdrt_sampler_create: max_deltaH must be positive"); return nullptr; }
    if (c.init_buffer < 0 || c.term_buffer < 0 || c.base_window < 0) { set_error("bdrt_sampler_create: adaptation window sizes must be >= 0"); return nullptr; }
    Problem &P = p->impl;
    if (hipSetDevice(P.device) != hipSuccess) { set_error("bdrt_sampler_create: hipSetDevice(%d) failed", P.device); return nullptr; }
    bdrt_sampler *s = new bdrt_sampler();
    Sampler &S = s->impl;
    memset(&S.args, 0, sizeof(S.args));
    S.prob = &P;
    S.n_units = n_units;
    // few chains of the headline family on log-uniform grids: one chain per workgroup (bdrt_solo.h)
    int n_cu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, P.device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    }
    // (measured at 81 x 161, profiles/r03/solo_duo.txt: one workgroup per CU 30.6 M evals/s, two per CU 42-45 M from 512 units on;
    //  the 16-chain kernel passes that at ~1300 units)
    S.solo = solo_capable(P.dev) && n_units <= (solo_duo_fits(P.dev) ? 5 * n_cu : 4 * n_cu);
    S.n_cu = n_cu;
    // a run that starts on the 16-chain kernel may hand its last live chains to the one-chain-per-workgroup kernel
    S.may_migrate = (solo_capable(P.dev) || wide1_capable(P.dev) || wave_capable(P.dev)) && !S.solo;
    if (const char *e = getenv("BDRT_SOLO")) {                                                  // diagnostics: force / forbid
        S.solo = solo_capable(P.dev) && atoi(e) != 0;
        S.may_migrate = false;
    }
    if (const char *e = getenv("BDRT_TAIL_MIGRATION")) S.may_migrate = S.may_migrate && atoi(e) != 0;
    if (getenv("BDRT_WIDE1") && atoi(getenv("BDRT_WIDE1")) == 0 && !solo_capable(P.dev)) S.may_migrate = false;
    // one chain per wave (bdrt_wave.h) for the one-chain layout: BDRT_WAVE=1 whenever the problem allows, 0 never
    S.solo_ok = solo_capable(P.dev);
    // (families without the LDS-resident one-chain kernel -- the outlier error models: up to one chain per CU the general one-chain kernel
    //  is the faster one, 12.9 against 16.9 us per round at 256 units; from there to four per CU the wave kernel, 53 against 29 M evals/s
    //  at 1024 units: profiles/r05/wave_outliers.txt)
    S.wave = wave_capable(P.dev);
    if (const char *e = getenv("BDRT_WAVE")) { S.wave_force = atoi(e) != 0; S.wave = S.wave && S.wave_force; }
    if (getenv("BDRT_CHAINS_PER_WG")) S.wave = false;                                           // (a forced packing means the 16-chain kernel)
    if (getenv("BDRT_SOLO") && S.wave_force != 1) S.wave = false;                                // (BDRT_SOLO=0 / 1: the 16-chain kernel / the 512-thread one-chain kernels, forced)
    if (S.wave && !getenv("BDRT_SOLO") && (S.wave_force == 1 || (n_units <= wave_max_units(n_cu, P.dev) && (wave_pays(n_units, n_cu) || (!S.solo_ok && (n_units > n_cu || !wide1_capable(P.dev))))))) {
        S.solo = true;                                                                          // start in the one-chain layout
        S.may_migrate = false;
    }
    if (S.wave) S.geomw = wave_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D, P.dev.nblocks);
    if (S.solo) S.geom = solo_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D);
    // chains per workgroup: fill every CU with one workgroup before putting a second chain on any wave
    {
        int cpw = (n_units + n_cu - 1) / n_cu;
        if (const char *e = getenv("BDRT_CHAINS_PER_WG")) cpw = atoi(e);     // diagnostics: force a packing
        S.args.cpw = S.solo ? 1 : std::min(NC, std::max(1, cpw));
    }
    S.n_wg = (n_units + S.args.cpw - 1) / S.args.cpw;
    S.D = P.dev.D;
    // few chains of a model the LDS-resident kernel does not cover: still one chain per workgroup, evaluated by 512 threads
    S.wide1 = !S.solo && n_units <= (11 * n_cu) / 4 && wide1_capable(P.dev);    // (measured at D = 818: 12.7 M evals/s from 256 units on; the 16-chain kernel passes that at ~750)
    if (const char *e = getenv("BDRT_WIDE1")) S.wide1 = S.wide1 && atoi(e) != 0;                // diagnostics: forbid
    if (getenv("BDRT_CHAINS_PER_WG")) S.wide1 = false;                                          // (a forced packing means the 16-chain kernel)
    if (S.wide1) {
        S.geom1 = wide1_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D, P.dev.nblocks);
        S.args.cpw = 1; S.n_wg = n_units;
        S.may_migrate = false;
    }
    // a problem beyond the LDS budget of the tile evaluators (bdrt_big.h): one chain per workgroup in the same row layout, the
    // streamed evaluator, whatever the number of units
    S.big = P.dev.big != 0 && !S.wide1;
    if (S.big) {
        S.solo = false; S.wave = false; S.wide1 = true;
        S.args.cpw = 1; S.n_wg = n_units;
        S.may_migrate = false;
    }
    S.np.warmup = warmup; S.np.n_draws = n_draws; S.np.max_depth = c.max_treedepth;
    S.np.delta = c.adapt_delta; S.np.gamma = c.adapt_gamma; S.np.t0 = c.adapt_t0; S.np.kappa = c.adapt_kappa;
    S.np.init_radius = c.init_radius; S.np.max_deltaH = c.max_deltaH; S.np.stepsize0 = c.stepsize0;
    S.np.seed_lo = (uint32_t)seed; S.np.seed_hi = (uint32_t)(seed >> 32);
    S.np.has_init = init_theta != nullptr;
    // the fast S1 kernel (theta rows resident in LDS) when the problem takes that path and the rows fit
    S.use_s1 = P.dev.fast_s1 && P.dev.outlier_mode == 0 && P.dev.D <= 2 * RW && P.dev.D <= 32 * 16 &&
               nuts_lds_bytes(P.dev, true) <= 160 * 1024;
    S.s1_hbm = !S.use_s1 && P.dev.fast_s1 && P.dev.D <= 32 * 16;
    S.hw = P.dev.fast_hw && P.dev.D <= 32 * 27;
    if (S.s1_hbm || S.hw)
        S.lds_bytes = (S.hw ? hw_lds_doubles(P.dev) : s1_lds_doubles(P.dev)) * sizeof(double) + NUTS16_SCALAR_LDS;
    else
        S.lds_bytes = nuts_lds_bytes(P.dev, S.use_s1);
    if (S.wide1 && !S.big) {
        const int ds1 = S.D <= 32 * 11 ? 32 * 11 : (S.D <= 32 * 16 ? 32 * 16 : 32 * 27);
        S.nhot1 = wide1_hot_rows(S.geom1, ds1);
        S.lds_bytes = std::max(S.lds_bytes, wide1_lds_bytes(S.geom1, ds1, S.nhot1));     // (one attribute value for every kernel)
    }
    if (S.big) S.lds_bytes = nuts_big_lds_bytes(big_njx(S.D));
    auto fail = [&](const char *msg) -> bdrt_sampler * { set_error("%s", msg); bdrt_sampler_destroy(s); return nullptr; };
    if (S.lds_bytes > 160 * 1024) return fail("bdrt_sampler_create: problem too large for the 160 KiB LDS budget");
    for (int u = 0; u < n_units; ++u)
        if (spec && (spec[u] < 0 || spec[u] >= P.dev.n_spectra)) return fail("bdrt_sampler_create: spectrum index out of range");

    // row stride of the state vectors = 32*NJ of the kernel instantiation; the solo kernel keeps [unit][row][ds] with one column
    const int DS = S.solo ? S.geom.DSS : (S.use_s1 ? 32 * s1_nj(S.D) : (S.D <= 32 * 11 ? 32 * 11 : (S.D <= 32 * 16 ? 32 * 16 : (S.D <= 32 * 27 ? 32 * 27 : 512 * big_njx(S.D)))));
    if (S.D > (S.big ? BIG_MAX_D : 32 * 27)) {
        set_error("bdrt_sampler_create: D = %d > %d not supported", S.D, S.big ? BIG_MAX_D : 864); bdrt_sampler_destroy(s); return nullptr;
    }
    S.args.ds = DS;
    const int ncol = (S.solo || S.wide1) ? 1 : NC, nrow = S.solo ? (int)SG_COUNT : (int)V_COUNT;
    const int r_minv = S.solo ? (int)SV_MINV : (int)V_MINV, r_th = S.solo ? (int)SV_TH : (int)V_TH;
    if (S.solo && S.solo_ok) S.lds_bytes = (size_t)S.geom.total * sizeof(double) + 64;
    const size_t nvec = (size_t)S.n_wg * nrow * ncol * DS;
    std::vector<double> hv(nvec, 0.0);
    std::vector<ChainState> hs((size_t)n_units);
    for (int u = 0; u < n_units; ++u) {
        ChainState &st = hs[u];
        memset(&st, 0, sizeof(st));
        st.phase = PH_INIT;
        st.z_iter = -1;
        st.spec = spec ? spec[u] : 0;
        st.chain_id = chain_id ? chain_id[u] : u;
        st.eps = c.stepsize0;
        st.dir = 1;
        st.lsw_sub = -INFINITY;
        window_init(st, warmup, c.init_buffer, c.term_buffer, c.base_window);
        const int wg = u / S.args.cpw, cc = (S.solo || S.wide1) ? 0 : slot_col(u % S.args.cpw);
        double *V = hv.data() + (size_t)wg * nrow * ncol * DS;
        const Philox rng = {S.np.seed_lo, S.np.seed_hi, (uint32_t)st.chain_id};
        for (int j = 0; j < S.D; ++j) {
            V[((size_t)r_minv * ncol + cc) * DS + j] = 1.0;
            V[((size_t)r_th * ncol + cc) * DS + j] =
                init_theta ? init_theta[(size_t)u * S.D + j]
                           : c.init_radius * (2.0 * rng_uniform(rng, (uint32_t)j, RNG_INIT, 0, 0, 0) - 1.0);
        }
        if (!init_theta) st.init_attempt = 0;
    }
    // unused columns (cpw < 16, last workgroup): finite placeholders
    for (int wg = 0; wg < S.n_wg && !S.solo && !S.wide1; ++wg)
        for (int k = 0; k < NC; ++k) {
            if (k < S.args.cpw && wg * S.args.cpw + k < n_units) continue;
            double *V = hv.data() + (size_t)wg * V_COUNT * NC * DS;
            for (int j = 0; j < S.D; ++j) V[((size_t)V_MINV * NC + slot_col(k)) * DS + j] = 1.0;
        }
    if (!S.solo && !S.wide1) {
        S.slot_unit.assign((size_t)S.n_wg * NC, -1);
        S.unit_loc.assign((size_t)n_units, -1);
        for (int u = 0; u < n_units; ++u) {
            const int wg = u / S.args.cpw, k = u % S.args.cpw;
            S.slot_unit[(size_t)wg * NC + k] = u;
            S.unit_loc[u] = wg * NC + k;
        }
        if (hipMalloc((void **)&S.d_slot_unit, S.slot_unit.size() * sizeof(int)) != hipSuccess) return fail("hipMalloc(slot map) failed");
        if (hipMalloc((void **)&S.d_unit_loc, S.unit_loc.size() * sizeof(int)) != hipSuccess) return fail("hipMalloc(slot map) failed");
        if (hipMemcpy(S.d_slot_unit, S.slot_unit.data(), S.slot_unit.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(S.d_unit_loc, S.unit_loc.data(), S.unit_loc.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return fail("bdrt_sampler_create: upload of the slot map failed");
        S.args.slot_unit = S.d_slot_unit;
        // more workgroups than CUs: finished chains can be squeezed out of the tiles (BDRT_COMPACTION=0 keeps the layout)
        S.may_compact = S.n_wg > n_cu && !(getenv("BDRT_COMPACTION") && atoi(getenv("BDRT_COMPACTION")) == 0);
    }
    S.vecs_capacity = nvec;
    if (hipMalloc((void **)&S.args.vecs, nvec * sizeof(double)) != hipSuccess) return fail("hipMalloc(vecs) failed");
    if (hipMalloc((void **)&S.args.states, hs.size() * sizeof(ChainState)) != hipSuccess) return fail("hipMalloc(states) failed");
    const size_t nd = (size_t)n_units * std::max(n_draws, 1) * S.D;
    if (hipMalloc((void **)&S.args.draws, nd * sizeof(double)) != hipSuccess) return fail("hipMalloc(draws) failed");
    if (hipMalloc((void **)&S.args.lp_draws, (size_t)n_units * std::max(n_draws, 1) * sizeof(double)) != hipSuccess)
        return fail("hipMalloc(lp) failed");
    if (hipMalloc((void **)&S.d_done, sizeof(int)) != hipSuccess) return fail("hipMalloc failed");
    if (hipMalloc((void **)&S.d_leaps, sizeof(unsigned long long)) != hipSuccess) return fail("hipMalloc failed");
    if (hipMalloc((void **)&S.d_active, sizeof(int)) != hipSuccess) return fail("hipMalloc failed");
    if (S.big && hipMalloc((void **)&S.d_bigws, (size_t)S.n_wg * big_ws_doubles(P.dev) * sizeof(double)) != hipSuccess) return fail("hipMalloc(workspace) failed");
    S.args.bigws = S.d_bigws;
    if (hipMemcpy(S.args.vecs, hv.data(), nvec * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return fail("bdrt_sampler_create: upload of the chain vectors failed");
    if (hipMemcpy(S.args.states, hs.data(), hs.size() * sizeof(ChainState), hipMemcpyHostToDevice) != hipSuccess)
        return fail("bdrt_sampler_create: upload of the chain states failed");
    if (hipMemset(S.args.draws, 0, nd * sizeof(double)) != hipSuccess) return fail("bdrt_sampler_create: clearing the draws failed");
    if (hipMemset(S.d_leaps, 0, sizeof(unsigned long long)) != hipSuccess)
        return fail("bdrt_sampler_create: clearing the leapfrog counter failed");
    // hipMemset returns before the fill has happened, and the sampler's kernels run on a NON-BLOCKING stream that does not order
    // itself behind the null stream: without this wait a launch that follows quickly (several host threads sampling at once)
    // can have its first draws / its leapfrog counter zeroed under it.
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail("bdrt_sampler_create: initial fills failed");
    S.args.leap_counter = S.d_leaps;
    S.args.done_counter = S.d_done;
    S.args.active_counter = S.d_active;
    S.args.unit_map = nullptr;
    S.n_solo = n_units;
    S.live = n_units;
    S.args.n_units = n_units;
    if (hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    static LdsAttrCache attr_cache;
    const hipError_t ae = attr_cache.ensure(S.lds_bytes, [&]() {
        const void *fns[38] = {(const void *)nuts_kernel<11, 2, 1, true>, (const void *)nuts_kernel<11, 2, 2, true>,
                               (const void *)nuts_kernel<7, 2, 2, true>, (const void *)nuts_kernel<27, 4, 0, true>,
                               (const void *)nuts_kernel<4, 2, 2>, (const void *)nuts_kernel<6, 2, 2>, (const void *)nuts_kernel<7, 2, 2>,
                               (const void *)nuts_kernel<4, 2, 0>, (const void *)nuts_kernel<6, 2, 0>, (const void *)nuts_kernel<7, 2, 0>,
                               (const void *)nuts_kernel<6, 2, 1>,
                               (const void *)nuts_kernel<11, 2, 2>, (const void *)nuts_kernel<16, 2, 2>,
                               (const void *)nuts_kernel<11, 2, 1>, (const void *)nuts_kernel<16, 2, 1>,
                               (const void *)nuts_kernel<11, 1>, (const void *)nuts_kernel<11, 0>,
                               (const void *)nuts_kernel<16, 1>, (const void *)nuts_kernel<16, 0>,
                               (const void *)nuts_kernel<27, 1>, (const void *)nuts_kernel<27, 0>,
                               (const void *)nuts_kernel<11, 2>, (const void *)nuts_kernel<16, 2>,
                               (const void *)nuts_kernel<11, 3>, (const void *)nuts_kernel<16, 3>,
                               (const void *)nuts_kernel<11, 4>, (const void *)nuts_kernel<16, 4>, (const void *)nuts_kernel<27, 4>,
                               (const void *)nuts_kernel<11, 3, 3>, (const void *)nuts_kernel<16, 3, 3>, (const void *)nuts_kernel<11, 3, 4>,
                               (const void *)nuts_kernel<16, 3, 4>, (const void *)nuts_kernel<11, 4, 3>, (const void *)nuts_kernel<16, 4, 3>,
                               (const void *)nuts_kernel<27, 4, 3>, (const void *)nuts_kernel<11, 4, 4>, (const void *)nuts_kernel<16, 4, 4>,
                               (const void *)nuts_kernel<27, 4, 4>};
        hipError_t e = hipFuncSetAttribute((const void *)nuts_solo_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)nuts_solo_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)nuts_solo_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)nuts_solo_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)nuts_wide1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)nuts_wide1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        for (int i = 0; i < 38 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes);
        return e;
    });
    if (ae != hipSuccess) {
        set_error("hipFuncSetAttribute(nuts_kernel, %zu B dynamic LDS) failed: %s", S.lds_bytes, hipGetErrorString(ae));
        bdrt_sampler_destroy(s);
        return nullptr;
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
    BDRT_HIP(hipSetDevice(S.prob->device));
    S.args.rounds = rounds;
    BDRT_HIP(hipMemsetAsync(S.d_done, 0, sizeof(int), S.stream));
    BDRT_HIP(hipMemsetAsync(S.d_active, 0, sizeof(int), S.stream));
    hipEvent_t e0, e1;
    BDRT_HIP(hipEventCreate(&e0));
    BDRT_HIP(hipEventCreate(&e1));
    BDRT_HIP(hipEventRecord(e0, S.stream));
    {
        const DevProblem *dp = (const DevProblem *)S.prob->d_dev;
        const bool tp = S.prob->dev.toep_all != 0;
#define BDRT_LAUNCH_NUTS(NJV)                                                                                          \
        do {                                                                                                           \
            if (tp) hipLaunchKernelGGL((nuts_kernel<NJV, 1>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args); \
            else hipLaunchKernelGGL((nuts_kernel<NJV, 0>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args);   \
        } while (0)
        if (S.big && S.D <= 1024)
            hipLaunchKernelGGL(nuts_big_kernel<2>, dim3(S.n_wg), dim3(SOLO_NT), nuts_big_lds_bytes(2), S.stream, dp, S.np, S.args);
        else if (S.big && S.D <= 2048)
            hipLaunchKernelGGL(nuts_big_kernel<4>, dim3(S.n_wg), dim3(SOLO_NT), nuts_big_lds_bytes(4), S.stream, dp, S.np, S.args);
        else if (S.big && S.D <= 4096)
            hipLaunchKernelGGL(nuts_big_kernel<8>, dim3(S.n_wg), dim3(SOLO_NT), nuts_big_lds_bytes(8), S.stream, dp, S.np, S.args);
        else if (S.big)
            hipLaunchKernelGGL(nuts_big_kernel<16>, dim3(S.n_wg), dim3(SOLO_NT), nuts_big_lds_bytes(16), S.stream, dp, S.np, S.args);
        else if (S.wide1 && S.args.prof)
            hipLaunchKernelGGL(nuts_wide1_kernel<true>, dim3(S.n_wg), dim3(SOLO_NT), wide1_lds_bytes(S.geom1, S.args.ds, S.nhot1), S.stream, dp,
                               S.np, S.args, S.geom1, S.nhot1);
        else if (S.wide1)
            hipLaunchKernelGGL(nuts_wide1_kernel<false>, dim3(S.n_wg), dim3(SOLO_NT), wide1_lds_bytes(S.geom1, S.args.ds, S.nhot1), S.stream, dp,
                               S.np, S.args, S.geom1, S.nhot1);
        else if (S.solo && S.wave && (S.wave_force == 1 || !S.solo_ok || wave_pays(S.live, S.n_cu)))
        {
            // LDS share (= chains per CU) by the chains still running: finished ones leave their wave at once
            int nhot = 0;
            const size_t lds = wave_lds_request(S.geomw, std::max(1, std::min(S.live, S.n_solo)), S.n_cu, &nhot, wave_chains_per_cu(S.prob->dev));
            if (launch_wave_nuts(dp, S.np, S.args, S.geomw, nhot, S.n_solo, lds, S.stream, S.prob->dev.outlier_mode != 0)) return -10;
            S.wave_last = true;
        }
        else if (S.solo)
        {
            S.wave_last = false;
            // more chains than CUs: two workgroups per CU (128 VGPRs each, 16 of the chain's rows in LDS) overlap each other's
            // latencies; with at most one chain per CU the full-LDS variant is the faster one.  BDRT_SOLO_DUO=0 / 1: never / always.
            const size_t lds2 = ((size_t)S.geom.o_vec + (size_t)SOLO_NHOT * S.geom.DSS) * sizeof(double) + 64;
            const char *e = getenv("BDRT_SOLO_DUO");
            const bool duo = 2 * lds2 <= 160 * 1024 && (e ? atoi(e) != 0 : S.n_solo > S.n_cu);
            if (duo && S.args.prof)
                hipLaunchKernelGGL((nuts_solo_kernel<4, true>), dim3(S.n_solo), dim3(SOLO_NT), lds2, S.stream, dp, S.np, S.args, S.geom);
            else if (duo)
                hipLaunchKernelGGL(nuts_solo_kernel<4>, dim3(S.n_solo), dim3(SOLO_NT), lds2, S.stream, dp, S.np, S.args, S.geom);
            else if (S.args.prof)
                hipLaunchKernelGGL((nuts_solo_kernel<2, true>), dim3(S.n_solo), dim3(SOLO_NT), S.lds_bytes, S.stream, dp, S.np, S.args, S.geom);
            else
                hipLaunchKernelGGL(nuts_solo_kernel<2>, dim3(S.n_solo), dim3(SOLO_NT), S.lds_bytes, S.stream, dp, S.np, S.args, S.geom);
        }
        else if (S.use_s1) {
            const int nj = S.args.ds / 32, ta = S.prob->dev.toepA;
#define BDRT_S1_NUTS(NJ_, TA_) hipLaunchKernelGGL((nuts_kernel<NJ_, 2, TA_>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args)
#define BDRT_S1_NUTS_PROF(NJ_, TA_) hipLaunchKernelGGL((nuts_kernel<NJ_, 2, TA_, true>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args)
            // (the phase profile is filled by the profiling instantiations: the headline family's and BASELINE config 5's)
            if (S.args.prof && nj == 11 && ta == 1) BDRT_S1_NUTS_PROF(11, 1);
            else if (S.args.prof && nj == 11 && ta == 2) BDRT_S1_NUTS_PROF(11, 2);
            else if (S.args.prof && nj == 7 && ta == 2) BDRT_S1_NUTS_PROF(7, 2);
            else
            if (nj == 4) { if (ta == 2) BDRT_S1_NUTS(4, 2); else BDRT_S1_NUTS(4, 0); }
            else if (nj == 6) { if (ta == 2) BDRT_S1_NUTS(6, 2); else if (ta == 1) BDRT_S1_NUTS(6, 1); else BDRT_S1_NUTS(6, 0); }
            else if (nj == 7) { if (ta == 2) BDRT_S1_NUTS(7, 2); else BDRT_S1_NUTS(7, 0); }
            else if (nj == 11) { if (ta == 2) BDRT_S1_NUTS(11, 2); else if (ta == 1) BDRT_S1_NUTS(11, 1); else BDRT_S1_NUTS(11, 0); }
            else { if (ta == 2) BDRT_S1_NUTS(16, 2); else if (ta == 1) BDRT_S1_NUTS(16, 1); else BDRT_S1_NUTS(16, 0); }
#undef BDRT_S1_NUTS
#undef BDRT_S1_NUTS_PROF
        }
        else if (S.s1_hbm || S.hw) {
            // the evaluator's instantiation by the longest basis (3, 4 or 6 basis functions per lane)
            const int kmax = S.hw ? hw_kmax(S.prob->dev) : S.prob->dev.blk[0].K;
            static const bool ku6 = getenv("BDRT_S1_KU") && atoi(getenv("BDRT_S1_KU")) == 6;     // (measurements: the K <= 192 instantiation)
            const int ku = ku6 ? 0 : (kmax <= 96 ? 3 : (kmax <= 128 ? 4 : 0));
            const int nj = S.D <= 32 * 11 ? 11 : (S.D <= 32 * 16 ? 16 : 27);
#define BDRT_X_NUTS(NJ_, MODE_, KU_) hipLaunchKernelGGL((nuts_kernel<NJ_, MODE_, KU_>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args)
#define BDRT_X_KU(NJ_, MODE_) do { if (ku == 3) BDRT_X_NUTS(NJ_, MODE_, 3); else if (ku == 4) BDRT_X_NUTS(NJ_, MODE_, 4); else BDRT_X_NUTS(NJ_, MODE_, 0); } while (0)
            if (S.s1_hbm) { if (nj == 11) BDRT_X_KU(11, 3); else BDRT_X_KU(16, 3); }
            else if (nj == 11) BDRT_X_KU(11, 4);
            else if (nj == 16) BDRT_X_KU(16, 4);
            else if (S.args.prof && ku == 0) hipLaunchKernelGGL((nuts_kernel<27, 4, 0, true>), dim3(S.n_wg), dim3(NT), S.lds_bytes, S.stream, dp, S.np, S.args);
            else BDRT_X_KU(27, 4);
#undef BDRT_X_KU
#undef BDRT_X_NUTS
        }
        else if (S.D <= 32 * 11) BDRT_LAUNCH_NUTS(11);
        else if (S.D <= 32 * 16) BDRT_LAUNCH_NUTS(16);
        else BDRT_LAUNCH_NUTS(27);
#undef BDRT_LAUNCH_NUTS
    }
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipEventRecord(e1, S.stream));
    S.pending.emplace_back(e0, e1);
    S.n_launch += 1;
    if (all_done) {
        int done = 0;
        BDRT_HIP(hipMemcpyAsync(&done, S.d_done, sizeof(int), hipMemcpyDeviceToHost, S.stream));
        BDRT_HIP(hipStreamSynchronize(S.stream));
        *all_done = done >= S.n_wg;
        if (S.solo) S.live = std::max(0, S.n_solo - done);
        return harvest_events(S, true);
    }
    return harvest_events(S, false);
}

int bdrt_sampler_sync(bdrt_sampler *s)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    BDRT_HIP(hipStreamSynchronize(S.stream));
    // the live-chain count that picks the next launch's kernel (one chain per wave / per workgroup, the wave kernel's LDS share):
    // `advance(..., NULL)` does not read the launch's done counter back, a sync does (the last launch wrote it)
    if (S.solo && S.n_launch > 0) {
        int done = 0;
        BDRT_HIP(hipMemcpy(&done, S.d_done, sizeof(int), hipMemcpyDeviceToHost));
        S.live = std::max(0, S.n_solo - done);
    }
    return harvest_events(S, true);
}

// frees its device pointers unless they were handed over (error paths of the two re-layout passes below)
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) hipFree(p); }
    void *release() { void *q = p; p = nullptr; return q; }
};

// Re-pack the live chains of a 16-chain run into fewer workgroups (nuts_compact_kernel).  Called between launches with the
// stream idle and `active` = live chains after the last launch.  Worth it only while the run has more workgroups than CUs:
// with one workgroup per CU a round costs the same whatever the number of live columns, and fewer workgroups would only idle CUs.
static int maybe_compact(Sampler &S, int active)
{
    if (!S.may_compact || S.migrated || S.solo || S.wide1 || active <= 0) return 0;
    const int target = std::max(S.n_cu, (active + NC - 1) / NC);
    // A launch runs its workgroups in turns of one per CU, and every turn lasts the full `rounds` however many CUs it fills:
    // what a re-packing buys is a whole turn, so it is done when -- and only when -- the live chains fit in one turn less
    // (measured on 1536 spectra x 8 chains: re-packing at every 1/16 of the workgroups, 15 passes, 32.7 s; at the turn
    // boundaries 8192 and 4096 live chains ... see profiles/r03/oversubscribed.txt; frozen layout 36.6 s)
    if ((target + S.n_cu - 1) / S.n_cu >= (S.n_wg + S.n_cu - 1) / S.n_cu) return 0;
    if (!S.vecs_alt) {
        // the second buffer and the liveness flags, once (keep going as is when the memory is not there)
        if (hipMalloc((void **)&S.vecs_alt, S.vecs_capacity * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); S.vecs_alt = nullptr; S.may_compact = false; return 0; }
        if (hipMalloc((void **)&S.d_slot_alt, S.slot_unit.size() * sizeof(int)) != hipSuccess ||
            hipMalloc((void **)&S.d_live, (size_t)S.n_units * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); S.may_compact = false; return 0; }
    }
    std::vector<int> alive((size_t)S.n_units);
    hipLaunchKernelGGL(nuts_live_kernel, dim3((S.n_units + 255) / 256), dim3(256), 0, S.stream, (const ChainState *)S.args.states, S.n_units, S.d_live);
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipMemcpyAsync(alive.data(), S.d_live, alive.size() * sizeof(int), hipMemcpyDeviceToHost, S.stream));
    BDRT_HIP(hipStreamSynchronize(S.stream));
    std::vector<int> live;
    for (int wgk = 0; wgk < (int)S.slot_unit.size(); ++wgk) {            // slot order: keeps neighbours (same spectrum) together
        const int u = S.slot_unit[wgk];
        if (u >= 0 && alive[u]) live.push_back(u);
    }
    if (live.empty()) return 0;
    const int n_wg = std::max(std::min(S.n_cu, (int)live.size()), ((int)live.size() + NC - 1) / NC);
    const int cpw = ((int)live.size() + n_wg - 1) / n_wg;
    std::vector<int> slot_unit((size_t)n_wg * NC, -1), unit_loc((size_t)S.n_units, -1);
    for (size_t i = 0; i < live.size(); ++i) {
        const int wg = (int)(i / cpw), k = (int)(i % cpw);
        slot_unit[(size_t)wg * NC + k] = live[i];
        unit_loc[live[i]] = wg * NC + k;
    }
    BDRT_HIP(hipMemcpyAsync(S.d_slot_alt, slot_unit.data(), slot_unit.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    // d_unit_loc still holds the OLD locations: the kernel reads them, then they are replaced
    hipLaunchKernelGGL(nuts_compact_kernel, dim3(n_wg), dim3(256), 0, S.stream, (const double *)S.args.vecs, (const int *)S.d_unit_loc,
                       (const int *)S.d_slot_alt, S.vecs_alt, S.args.ds);
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipMemcpyAsync(S.d_unit_loc, unit_loc.data(), unit_loc.size() * sizeof(int), hipMemcpyHostToDevice, S.stream));
    BDRT_HIP(hipStreamSynchronize(S.stream));            // (the host vectors above are read by the asynchronous copies)
    std::swap(S.args.vecs, S.vecs_alt);
    std::swap(S.d_slot_unit, S.d_slot_alt);
    S.args.slot_unit = S.d_slot_unit;
    S.slot_unit.swap(slot_unit);
    S.unit_loc.swap(unit_loc);
    S.n_wg = n_wg;
    S.n_compactions += 1;
    if (S.d_prof) { S.args.prof = nullptr; }              // (the phase-profile slots were laid out for the old workgroups)
    return 0;
}

// Hand the live chains of a 16-chain run to the one-chain-per-workgroup kernel when that finishes them sooner (see
// nuts_migrate_kernel).  Called between launches with the stream idle.
static int maybe_migrate_tail(Sampler &S, int active)
{
    // the one-chain kernels run one or two chains per CU at a time, ~4x faster per leapfrog: the LDS-resident one wins below ~4.75
    // live chains per CU when two of its workgroups fit a CU (else ~3.5), the general one below ~2.75
    const bool to_solo = solo_capable(S.prob->dev) || S.wave;
    // (the one-chain-per-wave kernel runs eight chains per CU at 108 M evals/s against 73 M of half-empty tiles: profiles/r04/wave_sweep.txt)
    const int limit = S.wave ? wave_chains_per_cu(S.prob->dev) * S.n_cu : (to_solo ? (solo_duo_fits(S.prob->dev) ? (19 * S.n_cu) / 4 : (7 * S.n_cu) / 2) : (11 * S.n_cu) / 4);
    if (active <= 0 || active > limit) return 0;
    std::vector<ChainState> hs((size_t)S.n_units);
    BDRT_HIP(hipMemcpy(hs.data(), S.args.states, hs.size() * sizeof(ChainState), hipMemcpyDeviceToHost));
    std::vector<int> map;
    for (int u = 0; u < S.n_units; ++u)
        if (hs[u].phase == PH_INIT || hs[u].phase == PH_EPS || hs[u].phase == PH_TREE) map.push_back(u);
    if (map.empty()) return 0;
    DevTmp vnew, dmap;
    if (!to_solo) {
        // general one-chain kernel: the 16-chain rows, one column
        const Wide1Geom G = wide1_geometry(S.prob->dev.nf, S.prob->dev.blk[0].K, S.prob->dev.D, S.prob->dev.nblocks);
        int nhot = W1_HOT_MAX;                                // as many LDS-resident rows as the LDS limit set at creation allows
        while (nhot > 0 && wide1_lds_bytes(G, S.args.ds, nhot) > std::min(S.lds_bytes, (size_t)158 * 1024)) --nhot;
        if (wide1_lds_bytes(G, S.args.ds, nhot) > S.lds_bytes) return 0;
        if (hipMalloc(&vnew.p, map.size() * (size_t)V_COUNT * S.args.ds * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (hipMalloc(&dmap.p, map.size() * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return 0; }
        BDRT_HIP(hipMemcpy(dmap.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(nuts_migrate_wide1_kernel, dim3((unsigned)map.size()), dim3(256), 0, S.stream, (const double *)S.args.vecs, S.args.ds,
                           (const int *)S.d_unit_loc, (const int *)dmap.p, (double *)vnew.p, S.args.states);
        BDRT_HIP(hipGetLastError());
        BDRT_HIP(hipStreamSynchronize(S.stream));
        S.vecs16 = S.args.vecs;
        S.args.prof = nullptr;
        S.args.vecs = (double *)vnew.release();
        S.d_unit_map = (int *)dmap.release();
        S.args.unit_map = S.d_unit_map;
        S.geom1 = G;
        S.nhot1 = nhot;
        S.wide1 = true;
        S.n_solo = (int)map.size();
        S.n_wg = S.n_solo;
        S.migrated = true;
        return 0;
    }
    const SoloGeom g = solo_geometry(S.prob->dev.nf, S.prob->dev.blk[0].K, S.prob->dev.D);
    const size_t lds = (size_t)g.total * sizeof(double) + 64;
    if (S.solo_ok && lds > S.lds_bytes) return 0;         // (bdrt_sampler_create raised every kernel's LDS limit to the 16-chain size)
    if (hipMalloc(&vnew.p, map.size() * (size_t)SG_COUNT * g.DSS * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return 0; }   // (keep going as is)
    if (hipMalloc(&dmap.p, map.size() * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    BDRT_HIP(hipMemcpy(dmap.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(nuts_migrate_kernel, dim3((unsigned)map.size()), dim3(256), 0, S.stream, (const double *)S.args.vecs, S.args.ds,
                       (const int *)S.d_unit_loc, (const int *)dmap.p, (double *)vnew.p, g.DSS, S.D);
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipStreamSynchronize(S.stream));
    S.vecs16 = S.args.vecs;
    S.args.prof = nullptr;                                // (the phase-profile slots are laid out per 16-chain workgroup)
    S.args.vecs = (double *)vnew.release();
    S.args.ds = g.DSS;
    S.d_unit_map = (int *)dmap.release();
    S.args.unit_map = S.d_unit_map;
    S.geom = g;
    S.lds_bytes = lds;
    S.solo = true;
    S.n_solo = (int)map.size();
    S.live = S.n_solo;
    S.n_wg = S.n_solo;                                   // (the all-done test counts finished workgroups)
    S.migrated = true;
    return 0;
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
        if (!done && !S.migrated && (S.may_migrate || S.may_compact)) {
            int active = 0;
            BDRT_HIP(hipMemcpy(&active, S.d_active, sizeof(int), hipMemcpyDeviceToHost));
            if (S.may_migrate && (rc = maybe_migrate_tail(S, active))) return rc;
            if (!S.migrated && (rc = maybe_compact(S, active))) return rc;
        }
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

int bdrt_sampler_tail_units(bdrt_sampler *s) { return s && s->impl.migrated ? s->impl.n_solo : 0; }
int bdrt_sampler_compactions(bdrt_sampler *s) { return s ? s->impl.n_compactions : -1; }
int bdrt_sampler_kind(bdrt_sampler *s)
{
    if (!s) return -1;
    const Sampler &S = s->impl;
    if (S.big) return 4;
    if (S.wide1) return 2;
    if (!S.solo) return 0;
    // before the first launch: what the first launch will use
    const bool w = S.n_launch ? S.wave_last : (S.wave && (S.wave_force == 1 || !S.solo_ok || wave_pays(S.live, S.n_cu)));
    return w ? 3 : 1;
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

int bdrt_sampler_phase_profile(bdrt_sampler *s, int enable, long long *cycles32)
{
    if (!s) return -1;
    Sampler &S = s->impl;
    BDRT_HIP(hipStreamSynchronize(S.stream));
    // d_prof holds prof_wg workgroups' slots: the layout may have changed since (compaction, tail migration switch the profile off
    // and can leave MORE workgroups than it was allocated for) -- every copy / fill below is sized by the allocation
    if (cycles32) {
        for (int k = 0; k < 32; ++k) cycles32[k] = 0;
        if (S.d_prof) {
            std::vector<long long> h((size_t)S.prof_wg * 32);
            BDRT_HIP(hipMemcpy(h.data(), S.d_prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            for (int w = 0; w < S.prof_wg; ++w)
                for (int k = 0; k < 32; ++k) cycles32[k] += h[(size_t)w * 32 + k];
        }
    }
    if (S.d_prof && (!enable || S.prof_wg < S.n_wg)) { hipFree(S.d_prof); S.d_prof = nullptr; S.prof_wg = 0; }
    if (enable && !S.d_prof) {
        BDRT_HIP(hipMalloc((void **)&S.d_prof, (size_t)S.n_wg * 32 * sizeof(long long)));
        S.prof_wg = S.n_wg;
    }
    if (S.d_prof) {
        BDRT_HIP(hipMemset(S.d_prof, 0, (size_t)S.prof_wg * 32 * sizeof(long long)));
        BDRT_HIP(hipStreamSynchronize(nullptr));        // (same ordering rule as in bdrt_sampler_create)
    }
    S.args.prof = S.d_prof;
    return 0;
}

int bdrt_sampler_percentiles(bdrt_sampler *s, int unit_lo, int unit_hi, int col0, int ncols, const double *Phi, int M,
                             const double *bias, const double *q, int nq, double *out)
{
    if (!s || !q || nq < 1 || !out) { set_error("bdrt_sampler_percentiles: null argument"); return -1; }
    Sampler &S = s->impl;
    if (unit_lo < 0 || unit_hi > S.n_units || unit_lo >= unit_hi || col0 < 0 || ncols < 1 || col0 + ncols > S.D ||
        (Phi && M < 1) || S.np.n_draws < 1) {
        set_error("bdrt_sampler_percentiles: bad unit / column range");
        return -1;
    }
    BDRT_HIP(hipStreamSynchronize(S.stream));
    const long rows = (long)(unit_hi - unit_lo) * S.np.n_draws;
    if (rows > (1L << 30)) { set_error("bdrt_sampler_percentiles: too many rows"); return -1; }
    const double *dX = S.args.draws + (size_t)unit_lo * S.np.n_draws * S.D + col0;
    return post_percentiles_device(dX, (int)rows, ncols, (long)S.D, Phi, M, bias, q, nq, out);
}

int bdrt_sampler_summary(bdrt_sampler *s, int unit_lo, int unit_hi, const double *q, int nq, double *mean, double *pct)
{
    if (!s || !q || nq < 1 || !pct) { set_error("bdrt_sampler_summary: null argument"); return -1; }
    Sampler &S = s->impl;
    if (unit_lo < 0 || unit_hi > S.n_units || unit_lo >= unit_hi || S.np.n_draws < 1) {
        set_error("bdrt_sampler_summary: bad unit range");
        return -1;
    }
    BDRT_HIP(hipSetDevice(S.prob->device));
    BDRT_HIP(hipStreamSynchronize(S.stream));
    const long rows = (long)(unit_hi - unit_lo) * S.np.n_draws;
    if (rows > (1L << 30)) { set_error("bdrt_sampler_summary: too many rows"); return -1; }
    const double *dX = S.args.draws + (size_t)unit_lo * S.np.n_draws * S.D;
    return post_percentiles_device(dX, (int)rows, S.D, (long)S.D, nullptr, 0, nullptr, q, nq, pct, S.prob->is_pos.data(), mean);
}

const double *bdrt_sampler_draws_dev(bdrt_sampler *s)
{
    if (!s) return nullptr;
    hipStreamSynchronize(s->impl.stream);
    return s->impl.args.draws;
}

/* parity-test hook (not part of include/bdrt.h): the evaluator of the one-chain-per-workgroup path on B points */
int bdrt_debug_solo_logp_grad(bdrt_problem *p, const double *theta, const int *spec, int B, int jacobian, double *lp, double *grad)
{
    if (!p || !theta || B < 1) { set_error("bdrt_debug_solo_logp_grad: bad arguments"); return -1; }
    Problem &P = p->impl;
    if (!solo_capable(P.dev)) { set_error("problem does not take the solo path"); return -2; }
    BDRT_HIP(hipSetDevice(P.device));
    const SoloGeom g = solo_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D);
    const size_t lds = solo_eval_lds_bytes(g);
    BDRT_HIP(hipFuncSetAttribute((const void *)solo_eval_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double *dth = nullptr, *dlp = nullptr, *dg = nullptr;
    int *dsp = nullptr;
    const size_t nb = (size_t)B * P.dev.D * sizeof(double);
    BDRT_HIP(hipMalloc((void **)&dth, nb)); BDRT_HIP(hipMalloc((void **)&dg, nb)); BDRT_HIP(hipMalloc((void **)&dlp, B * sizeof(double)));
    BDRT_HIP(hipMemcpy(dth, theta, nb, hipMemcpyHostToDevice));
    if (spec) { BDRT_HIP(hipMalloc((void **)&dsp, B * sizeof(int))); BDRT_HIP(hipMemcpy(dsp, spec, B * sizeof(int), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(solo_eval_kernel<false>, dim3(B), dim3(SOLO_NT), lds, 0, (const DevProblem *)P.d_dev, g, dth, dsp, B, jacobian, dlp, dg);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && lp) e = hipMemcpy(lp, dlp, B * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && grad) e = hipMemcpy(grad, dg, nb, hipMemcpyDeviceToHost);
    hipFree(dth); hipFree(dg); hipFree(dlp); hipFree(dsp);
    if (e != hipSuccess) { set_error("bdrt_debug_solo_logp_grad: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

// the one-chain-per-wave evaluator (bdrt_wave.h) on B points: parity tests
int bdrt_debug_wave_logp_grad(bdrt_problem *p, const double *theta, const int *spec, int B, int jacobian, double *lp, double *grad)
{
    if (!p || !theta || B < 1) { set_error("bdrt_debug_wave_logp_grad: bad arguments"); return -1; }
    Problem &P = p->impl;
    if (!wave_capable(P.dev)) { set_error("problem does not take the one-chain-per-wave path"); return -2; }
    BDRT_HIP(hipSetDevice(P.device));
    const WaveGeom g = wave_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D, P.dev.nblocks);
    const size_t lds = wave_lds_bytes(g, 0);
    double *dth = nullptr, *dlp = nullptr, *dg = nullptr;
    int *dsp = nullptr;
    const size_t nb = (size_t)B * P.dev.D * sizeof(double);
    BDRT_HIP(hipMalloc((void **)&dth, nb)); BDRT_HIP(hipMalloc((void **)&dg, nb)); BDRT_HIP(hipMalloc((void **)&dlp, B * sizeof(double)));
    BDRT_HIP(hipMemcpy(dth, theta, nb, hipMemcpyHostToDevice));
    if (spec) { BDRT_HIP(hipMalloc((void **)&dsp, B * sizeof(int))); BDRT_HIP(hipMemcpy(dsp, spec, B * sizeof(int), hipMemcpyHostToDevice)); }
    int rc = launch_wave_eval((const DevProblem *)P.d_dev, g, dth, dsp, B, jacobian, dlp, dg, std::min(B, 2048), lds, 0, P.dev.outlier_mode != 0);
    hipError_t e = rc ? hipErrorUnknown : hipDeviceSynchronize();
    if (e == hipSuccess && lp) e = hipMemcpy(lp, dlp, B * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && grad) e = hipMemcpy(grad, dg, nb, hipMemcpyDeviceToHost);
    hipFree(dth); hipFree(dg); hipFree(dlp); hipFree(dsp);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("bdrt_debug_wave_logp_grad: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int bdrt_debug_wide1_logp_grad(bdrt_problem *p, const double *theta, const int *spec, int B, int jacobian, double *lp, double *grad)
{
    if (!p || !theta || B < 1) { set_error("bdrt_debug_wide1_logp_grad: bad arguments"); return -1; }
    Problem &P = p->impl;
    if (!wide1_capable(P.dev)) { set_error("problem does not take the general one-chain evaluator"); return -2; }
    BDRT_HIP(hipSetDevice(P.device));
    const Wide1Geom G = wide1_geometry(P.dev.nf, P.dev.blk[0].K, P.dev.D, P.dev.nblocks);
    const size_t lds = (size_t)G.total * sizeof(double) + 64;
    BDRT_HIP(hipFuncSetAttribute((const void *)wide1_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double *dth = nullptr, *dlp = nullptr, *dg = nullptr;
    int *dsp = nullptr;
    const size_t nbytes = (size_t)B * P.dev.D * sizeof(double);
    BDRT_HIP(hipMalloc((void **)&dth, nbytes)); BDRT_HIP(hipMalloc((void **)&dg, nbytes)); BDRT_HIP(hipMalloc((void **)&dlp, B * sizeof(double)));
    BDRT_HIP(hipMemcpy(dth, theta, nbytes, hipMemcpyHostToDevice));
    BDRT_HIP(hipMemset(dg, 0, nbytes));
    if (spec) { BDRT_HIP(hipMalloc((void **)&dsp, B * sizeof(int))); BDRT_HIP(hipMemcpy(dsp, spec, B * sizeof(int), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(wide1_eval_kernel, dim3(B), dim3(SOLO_NT), lds, 0, (const DevProblem *)P.d_dev, G, dth, dsp, B, jacobian, dlp, dg);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && lp) e = hipMemcpy(lp, dlp, B * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && grad) e = hipMemcpy(grad, dg, nbytes, hipMemcpyDeviceToHost);
    hipFree(dth); hipFree(dg); hipFree(dlp); hipFree(dsp);
    if (e != hipSuccess) { set_error("bdrt_debug_wide1_logp_grad: %s", hipGetErrorString(e)); return -10; }
    return 0;
}

int bdrt_debug_leaf_joins(const double *lsw_sub, const double *w, const double *u, int n, double *lsw_dev, int *join_dev, double *lsw_ref,
                          int *join_ref, double *prob_ref)
{
    if (!lsw_sub || !w || !u || !lsw_dev || !join_dev || !lsw_ref || !join_ref || !prob_ref || n < 1) {
        set_error("bdrt_debug_leaf_joins: bad arguments"); return -1;
    }
    // every device buffer is owned by this list: an error path frees what was allocated before it (the pattern of newton_polish_device)
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *q : owned) hipFree(q); };
#define LJ_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return -10; } } while (0)
    double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int *di[2] = {nullptr, nullptr};
    for (auto &q : d) { LJ_HIP(hipMalloc((void **)&q, (size_t)n * sizeof(double))); owned.push_back(q); }
    for (auto &q : di) { LJ_HIP(hipMalloc((void **)&q, (size_t)n * sizeof(int))); owned.push_back(q); }
    LJ_HIP(hipMemcpy(d[0], lsw_sub, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    LJ_HIP(hipMemcpy(d[1], w, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    LJ_HIP(hipMemcpy(d[2], u, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(leaf_joins_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], d[2], n, d[3], di[0], d[4], di[1], d[5]);
    LJ_HIP(hipGetLastError());
    LJ_HIP(hipMemcpy(lsw_dev, d[3], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    LJ_HIP(hipMemcpy(join_dev, di[0], (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    LJ_HIP(hipMemcpy(lsw_ref, d[4], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    LJ_HIP(hipMemcpy(join_ref, di[1], (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    LJ_HIP(hipMemcpy(prob_ref, d[5], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
#undef LJ_HIP
    cleanup();
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
