// bdrt_nuts_wide.h -- cooperative post-evaluation stages of the wide-vector sampler kernels (D > 352 with the generic tile,
// D > 512 with the half-wave tile: outlier error model, several distributions; BASELINE config 5 has D = 818).
//
// In these kernels a chain's vectors do not fit in the registers of its half-wave next to the evaluator, so its rows are
// streamed from HBM in chunks of MB elements per lane: one dependent memory trip per chunk and per stage, ~5.7 k cycles each
// even on an idle chip (measured: profiles/r02/config5.txt).  A plain tree leaf needs one such pass (nuts_kernel,
// stage C of the wide path).  Everything else -- a leaf that closes sub-subtrees above level 1, the last leaf of a subtree,
// the end of a transition, the step-size search -- used to cost the chain's wave 40-130 k cycles more while the other seven
// waves of the workgroup waited at the round barrier.
//
// Here those chains are finished by the WHOLE workgroup instead: element j of every vector in thread j (two elements per
// thread up to D = 1024), so each stage is a single memory trip with everything in registers, the U-turn dot products of all
// merge levels are reduced together, and the momentum normals of a new transition come from 512 threads at once.  The
// scalar logic is the register version of nuts_kernel (same statements, Stan 2.19 semantics: see the comments there), run
// redundantly by every thread on a private copy of the chain state that thread 0 writes back.
//
// (Included by bdrt_nuts.hip inside namespace bdrt, after the state-vector enum and NutsArgs.)
#pragma once

constexpr int WIDE_NT = 512, WIDE_NW = WIDE_NT / 64;

struct WideCtx {
    const DevProblem *P;
    const NutsParams *np;
    const NutsArgs *a;
    double *V;              // this workgroup's state vectors [V_COUNT][NC][DS]
    double *smem;           // LDS base: the tile region is idle between two evaluations (reduction scratch, normals)
    ChainState *sts;        // [NC] LDS
    const double *lpn;      // [NC] LDS: lp of the point just evaluated
    const int *hvy;         // [NC] LDS: 0 nothing left, 1 everything after the evaluation, 2 second half kick done and the transition ends
    const double *hvk;      // [NC][2] LDS: kinetic energy / non-finite gradient count of the kind-2 chains
    long long *prof;        // optional cycle counters (slots 10..16: stages of the cooperative phase, thread 0)
    int D, DS, TH2OFF, c0, nvalid;
    const int *slot_unit;   // 16-chain kernel: unit of each of the workgroup's 16 slots (-1: empty); nullptr: unit = c0 + slot
    int ncol;               // columns of the row layout: V[(v * ncol + column) * DS] (16; 1 for the one-chain kernel of bdrt_solo_wide.h)
    // one-chain kernel (HOT = true): rows of the chain that live in LDS for the launch -- hot_slot[v] >= 0: hot_base + slot * DS
    double *hot_base;
    const signed char *hot_slot;
};

// sums over the workgroup of N per-thread values (fixed order: lanes by the wave reduction, then waves 0..7); every thread
// gets all N totals.  scr: 9 N doubles of LDS.  Two barriers; scr may be reused by the next call at once.
template <int N>
__device__ inline void wide_block_sum(double (&v)[N], int n, double *scr, int tid)      // only v[0 .. n) are summed (n uniform)
{
    const int lane = tid & 63, wave = tid >> 6;
    if (N >= 8 && n <= 8) {
        // up to eight values (a leaf that closes at most two sub-subtrees: seven of eight): ONE butterfly puts sum j of the wave in lane j,
        // one barrier, then every wave adds the eight waves' partials itself -- lane l reads partial l (wave l >> 3, value l & 7), three
        // exchanges -- instead of a wave reduction per value, a second barrier and n broadcast reads per thread (bdrt_solo.h)
        const double q[8] = {v[0], v[1], v[2], v[3], v[4 % N], v[5 % N], v[6 % N], v[7 % N]};
        double t = sum32_by_lane<8>(q, lane);
        t += __shfl_xor(t, 32);
        if (lane < 8) scr[wave * 8 + lane] = t;
        __syncthreads();
        double u = scr[lane];
        u += dpp_perm<0x128>(u);
        u += swizzle_xor16(u);
        u += __shfl_xor(u, 32);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < N) v[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(u), i), __builtin_amdgcn_readlane(__double2loint(u), i));
        __syncthreads();                                // (scr may be reused by the next call at once)
        return;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < n) {
            const double t = solo_wave_sum(v[i]);       // DPP + readlane: no ds_bpermute round trips
            if (lane == 0) scr[wave * N + i] = t;
        }
    }
    __syncthreads();
    if (tid < n) {                                      // thread i adds the eight wave partials of value i
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WIDE_NW; ++w) t += scr[w * N + tid];
        scr[WIDE_NW * N + tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < n) v[i] = scr[WIDE_NW * N + i];
}
template <int N>
__device__ inline void wide_block_sum(double (&v)[N], double *scr, int tid) { wide_block_sum<N>(v, N, scr, tid); }

// Finish the round of chain (column) hc.  All threads of the workgroup must call with the same arguments.
template <int NJX, bool HOT = false>   // NJX elements per thread: D <= 512 * NJX; HOT: some rows of the chain are in LDS (WideCtx::hot_slot)
__device__ inline void wide_coop_tail(const WideCtx &x, int hc, bool c_done, unsigned long long &my_leaps, int tid)
{
    const NutsParams &np = *x.np;
    const NutsArgs &a = *x.a;
    const int D = x.D, DS = x.DS;
    long long tw_ = (x.prof && tid == 0) ? clock64() : 0;
#define BDRT_WIDE_PROF(slot) do { if (x.prof && tid == 0) { const long long t_ = clock64(); x.prof[slot] += t_ - tw_; tw_ = t_; } } while (0)
    ChainState s;
    chain_state_copy(s, x.sts[hc]);                    // (member by member: a struct assignment goes through scratch)
    const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};
    auto row = [&](int v) -> double * {
        if constexpr (HOT) { const int h = x.hot_slot[v]; if (h >= 0) return x.hot_base + (size_t)h * DS; }
        return x.V + ((size_t)v * x.ncol + hc) * DS;
    };
    const int kslot = col_slot(hc);
    const int unit = x.slot_unit ? x.slot_unit[kslot] : x.c0 + kslot;
    const bool valid = x.slot_unit ? unit >= 0 : kslot < x.nvalid;
    double *scr = x.smem, *zrow = x.smem + 512;        // (9 * 2 * MAXD = 180 doubles of reduction scratch)
    double *TH = row(V_TH) + (s.thsel ? x.TH2OFF : 0), *Pm = row(V_P), *G = row(V_G), *MI = row(V_MINV);

    const int ph0 = s.phase;
    const double e = ph0 == PH_EPS ? s.eps : (ph0 == PH_TREE ? s.dir * s.eps : 0.0);
    const int dir_now = s.dir, leaf_now = s.leaf;
    int jj[NJX];
    bool in[NJX];
#pragma unroll
    for (int m = 0; m < NJX; ++m) { const int j = tid + WIDE_NT * m; in[m] = j < D; jj[m] = in[m] ? j : 0; }

    // ---- one trip for everything the leaf may need: the chain's state rows and -- known from the leaf index before the
    //      verdict on the leaf itself -- the waiting siblings of all levels it closes and, for the last leaf of a subtree, the
    //      trajectory sum and the momentum of the other end.  Second half kick, kinetic energy, every merge's U-turn dot
    //      products and those of the subtree close are reduced together (speculatively: a divergent leaf discards them).
    double p_[NJX], g_[NJX], mi_[NJX], th_[NJX];
#pragma unroll
    for (int m = 0; m < NJX; ++m) { p_[m] = Pm[jj[m]]; g_[m] = G[jj[m]]; mi_[m] = MI[jj[m]]; th_[m] = TH[jj[m]]; }
    int nm_pre = 0;
    bool last_pre = false;
    if (ph0 == PH_TREE && !c_done) {
        while ((leaf_now >> nm_pre) & 1) ++nm_pre;
        last_pre = leaf_now == s.nleaves - 1;
    }
    double rc_[NJX], cpl_[NJX], rt_[NJX];              // rho / first momentum of the sub-subtree that ends here; trajectory rho
    constexpr int NRED = 4 + 2 * MAXD;                  // kin, nonfin, close t0, t1, then two per merge level
    double red[NRED];
#pragma unroll
    for (int i = 0; i < NRED; ++i) red[i] = 0.0;
    double kin = 0.0, nonfin = 0.0;
    if (!c_done) {
        double lr_[MAXD][NJX], lp_[MAXD][NJX], ro_[NJX], po_[NJX];
#pragma unroll
        for (int l = 0; l < MAXD; ++l) {
            if (l < nm_pre) {
                const double *RL = row(l == 0 ? V_CKP : V_CKC + l), *PL = row(V_CKP + l);
#pragma unroll
                for (int m = 0; m < NJX; ++m) { lr_[l][m] = RL[jj[m]]; lp_[l][m] = PL[jj[m]]; }
            } else {
#pragma unroll
                for (int m = 0; m < NJX; ++m) { lr_[l][m] = 0.0; lp_[l][m] = 0.0; }
            }
        }
#pragma unroll
        for (int m = 0; m < NJX; ++m) { ro_[m] = 0.0; po_[m] = 0.0; }
        if (last_pre) {
            const double *RHO = row(V_RHO), *PO = row(dir_now > 0 ? V_PM : V_PP);
#pragma unroll
            for (int m = 0; m < NJX; ++m) { ro_[m] = RHO[jj[m]]; po_[m] = PO[jj[m]]; }
        }
#pragma unroll
        for (int m = 0; m < NJX; ++m) {
            if (in[m]) {
                const double p = p_[m] + 0.5 * e * g_[m];
                p_[m] = p;
                red[0] += mi_[m] * p * p;
                red[1] += isfinite(g_[m]) ? 0.0 : 1.0;
            }
            rc_[m] = p_[m]; cpl_[m] = p_[m];
        }
#pragma unroll
        for (int l = 0; l < MAXD; ++l) {
            if (l < nm_pre) {
#pragma unroll
                for (int m = 0; m < NJX; ++m)
                    if (in[m]) {
                        const double rho = lr_[l][m] + rc_[m];
                        red[4 + 2 * l] += mi_[m] * lp_[l][m] * rho;
                        red[5 + 2 * l] += mi_[m] * p_[m] * rho;
                        rc_[m] = rho;
                        cpl_[m] = lp_[l][m];
                    }
            }
        }
#pragma unroll
        for (int m = 0; m < NJX; ++m) {
            rt_[m] = ro_[m] + rc_[m];
            if (in[m] && last_pre) {
                red[2] += mi_[m] * po_[m] * rt_[m];
                red[3] += mi_[m] * p_[m] * rt_[m];
            }
        }
        wide_block_sum<NRED>(red, 4 + 2 * nm_pre, scr, tid);
        kin = 0.5 * red[0]; nonfin = red[1];
    } else {
#pragma unroll
        for (int m = 0; m < NJX; ++m) { rc_[m] = p_[m]; cpl_[m] = p_[m]; rt_[m] = 0.0; }
        kin = x.hvk[2 * hc]; nonfin = x.hvk[2 * hc + 1];
    }

    BDRT_WIDE_PROF(9);
    // ---- S1: scalar logic after the evaluation
    bool copyq = false, cur2s = false, tree = false, last = false;
    bool upds = false, welf = false, wend = false;
    int nm = 0, endt = 0, next = 0, draw = -1;
    double wn = 0.0;
    {
        const double lp = x.lpn[hc];
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
            if (tid == 0) my_leaps += 1;
            next = nuts_stepsize_trial(s, np, lp, kin);
        } else {   // PH_TREE: one new leaf
            if (tid == 0) my_leaps += 1;
            nuts_tree_leaf(s, np, rng, lp, kin, leaf_now, copyq, tree, nm, last, endt);
        }
    }

    BDRT_WIDE_PROF(11);
    // ---- D: proposal copy, merges of the waiting sub-subtrees (all levels in one trip), subtree close
    if (copyq || cur2s) {
        double *THQ = row(V_THQ), *GQ = row(V_GQ), *THS = row(V_THS), *GS = row(V_GS);
#pragma unroll
        for (int m = 0; m < NJX; ++m)
            if (in[m]) {
                const int j = jj[m];
                if (copyq) { THQ[j] = th_[m]; GQ[j] = g_[m]; }
                if (cur2s) { THS[j] = th_[m]; GS[j] = g_[m]; }
            }
    }
    if (tree) {
        bool ok = !c_done;                              // (kind 2 and not divergent: the chain's own pass found the U-turn)
#pragma unroll
        for (int l = 0; l < MAXD; ++l)
            if (l < nm) ok = ok && (red[4 + 2 * l] > 0.0) && (red[5 + 2 * l] > 0.0);
        if (ok && !last) {
            double *PLn = row(V_CKP + nm), *RLn = row(V_CKC + nm);
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) { PLn[jj[m]] = cpl_[m]; if (nm > 0) RLn[jj[m]] = rc_[m]; }
        }
        if (!ok) {
            endt = 1;                                   // U-turn inside the new subtree: discard it, stop
        } else if (last) {
            // subtree complete and valid (Stan base_nuts::transition after build_tree): extend the trajectory
            double *RHO = row(V_RHO);
            double *THE = row(dir_now > 0 ? V_THP : V_THM), *PE = row(dir_now > 0 ? V_PP : V_PM);
            double *GE = row(dir_now > 0 ? V_GP : V_GM);
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) {
                    const int j = jj[m];
                    RHO[j] = rt_[m];
                    THE[j] = th_[m]; PE[j] = p_[m]; GE[j] = g_[m];
                }
            const int depth = s.depth + 1;
            s.depth = depth;
            const double lsw = s.lsw, lsw_sub = s.lsw_sub;
            bool take;
            if (lsw_sub > lsw) take = true;
            else take = rng_uniform(rng, 0, RNG_TOP, (uint32_t)depth, 0, (uint32_t)s.iter) < BDRT_NUTS_EXP(lsw_sub - lsw);
            if (take) { upds = true; s.lps = s.lpq; }
            s.lsw = log_sum_exp2(lsw, lsw_sub);
            const bool keep_going = (red[2] > 0.0) && (red[3] > 0.0);
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
    BDRT_WIDE_PROF(12);
    if (endt) {
        next = nuts_transition_end(s, np, endt, draw, welf, wend, wn);
        if (draw >= 0 && a.lp_draws && valid && tid == 0) a.lp_draws[(size_t)unit * np.n_draws + draw] = s.lps;
    }

    // ---- A': the trajectory continues from the point just evaluated
    if (next == 0 && s.phase == PH_TREE) {
        const double e1 = s.dir * s.eps;
#pragma unroll
        for (int m = 0; m < NJX; ++m)
            if (in[m]) {
                const double p = p_[m] + 0.5 * e1 * g_[m];
                Pm[jj[m]] = p;
                TH[jj[m]] = th_[m] + e1 * mi_[m] * p;
            }
    }

    BDRT_WIDE_PROF(13);
    // ---- E: sample update, metric adaptation, draw output, start of the next leapfrog when the trajectory does not continue
    if (upds || welf || wend || draw >= 0 || next) {
        const uint32_t iter = (uint32_t)s.iter, trial = (uint32_t)s.eps_trials, att = (uint32_t)s.init_attempt;
        double *THS = row(V_THS), *GS = row(V_GS);
        double ths_[NJX], gs_[NJX];
#pragma unroll
        for (int m = 0; m < NJX; ++m) { ths_[m] = 0.0; gs_[m] = 0.0; }
        if (upds || welf || wend || draw >= 0 || next == 1 || next == 3) {
            // current sample: the proposal of the tree if it was just accepted, else the stored sample.  (After a cur2s copy
            // of this very round the stored sample is what this thread wrote above: same thread, same elements.)
            const double *ST = upds ? row(V_THQ) : THS, *SG = upds ? row(V_GQ) : GS;
#pragma unroll
            for (int m = 0; m < NJX; ++m) { ths_[m] = ST[jj[m]]; gs_[m] = SG[jj[m]]; }
        }
        if (upds) {
#pragma unroll
            for (int m = 0; m < NJX; ++m) if (in[m]) { THS[jj[m]] = ths_[m]; GS[jj[m]] = gs_[m]; }
        }
        if (welf || wend) {
            double *WM = row(V_WMEAN), *W2 = row(V_WM2);
            double wm_[NJX], w2_[NJX];
#pragma unroll
            for (int m = 0; m < NJX; ++m) { wm_[m] = WM[jj[m]]; w2_[m] = W2[jj[m]]; }
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) {
                    double mean = wm_[m], m2 = w2_[m];
                    if (welf) {            // Welford (stan::math::welford_var_estimator)
                        const double delta = ths_[m] - mean;
                        mean += delta / wn;
                        m2 += (ths_[m] - mean) * delta;
                    }
                    if (wend) {            // var_adaptation::learn_variance
                        const double var = wn > 1.0 ? m2 / (wn - 1.0) : 0.0;
                        mi_[m] = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
                        MI[jj[m]] = mi_[m];
                        mean = 0.0; m2 = 0.0;
                    }
                    WM[jj[m]] = mean; W2[jj[m]] = m2;
                }
        }
        if (draw >= 0 && valid) {
            double *dr = a.draws + ((size_t)unit * np.n_draws + draw) * D;
#pragma unroll
            for (int m = 0; m < NJX; ++m) if (in[m]) dr[jj[m]] = ths_[m];
        }
        if (next == 1 || next == 3) {
            // fresh momentum p ~ N(0, M), M = diag(1/Minv): normals 2i and 2i+1 share one Philox block and one Box-Muller
            // transform; thread i produces the pair, the LDS row hands element j to thread j
            // (a thread produces the pairs tid, tid + 512, ...: one pair per thread covers D <= 1024, the streamed path goes beyond)
#pragma unroll
            for (int mp = 0; mp < (NJX + 1) / 2; ++mp) {
                const int i = tid + WIDE_NT * mp;
                if (2 * i < D) {
                    double z0, z1;
                    rng_normal_pair(rng, (uint32_t)i, next == 1 ? RNG_MOMENTUM : RNG_EPS_MOMENTUM, next == 1 ? 0u : trial, iter, z0, z1);
                    zrow[2 * i] = z0; zrow[2 * i + 1] = z1;
                }
            }
            __syncthreads();
            double pn_[NJX];
            double k0[1] = {0.0};
#pragma unroll
            for (int m = 0; m < NJX; ++m) {
                const double p = in[m] ? zrow[jj[m]] / sqrt(mi_[m]) : 0.0;
                pn_[m] = p;
                k0[0] += in[m] ? mi_[m] * p * p : 0.0;
            }
            wide_block_sum<1>(k0, scr, tid);     // (its barriers also protect zrow until the next use)
            s.H0 = -s.lps + 0.5 * k0[0];
            if (next == 1) {
                s.lsw = 0.0; s.lsw_sub = -INFINITY; s.depth = 0; s.leaf = 0; s.nleaves = 1;
                s.n_leap_iter = 0; s.sum_metro = 0.0;
                s.dir = rng_uniform(rng, 0, RNG_DIRECTION, 0, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
            }
            const double e1 = next == 1 ? s.dir * s.eps : s.eps;
            double *rTHM = row(V_THM), *rTHP = row(V_THP), *rPM = row(V_PM), *rPP = row(V_PP), *rGM = row(V_GM),
                   *rGP = row(V_GP), *rRHO = row(V_RHO);
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) {
                    const int j = jj[m];
                    const double p = pn_[m];
                    if (next == 1) {
                        rTHM[j] = ths_[m]; rTHP[j] = ths_[m];
                        rPM[j] = p; rPP[j] = p;
                        rGM[j] = gs_[m]; rGP[j] = gs_[m];
                        rRHO[j] = p;
                    }
                    const double pk = p + 0.5 * e1 * gs_[m];
                    Pm[j] = pk;
                    TH[j] = ths_[m] + e1 * mi_[m] * pk;
                }
        } else if (next == 2) {
            // continue from the trajectory end in the new direction
            const int dir = s.dir;
            const double e1 = dir * s.eps;
            double et_[NJX], ep_[NJX], eg_[NJX];
            if (dir == dir_now) {
#pragma unroll
                for (int m = 0; m < NJX; ++m) { et_[m] = th_[m]; ep_[m] = p_[m]; eg_[m] = g_[m]; }
            } else {
                const double *ET = row(dir > 0 ? V_THP : V_THM), *EP = row(dir > 0 ? V_PP : V_PM), *EG = row(dir > 0 ? V_GP : V_GM);
#pragma unroll
                for (int m = 0; m < NJX; ++m) { et_[m] = ET[jj[m]]; ep_[m] = EP[jj[m]]; eg_[m] = EG[jj[m]]; }
            }
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) {
                    const double pk = ep_[m] + 0.5 * e1 * eg_[m];
                    Pm[jj[m]] = pk;
                    TH[jj[m]] = et_[m] + e1 * mi_[m] * pk;
                }
        } else if (next == 4) {
#pragma unroll
            for (int m = 0; m < NJX; ++m)
                if (in[m]) {
                    TH[jj[m]] = np.init_radius * (2.0 * rng_uniform(rng, (uint32_t)jj[m], RNG_INIT, 0, att, 0) - 1.0);
                    Pm[jj[m]] = 0.0;
                }
        }
    }
    // every thread holds the same new state; the rows written above are read next by the evaluator / the chain's own lanes
    // after the round barrier
    if (tid == 0) chain_state_copy(x.sts[hc], s);
    BDRT_WIDE_PROF(14);
    if (x.prof && tid == 0) x.prof[15] += 1;
#undef BDRT_WIDE_PROF
}
