// bdrt_wave_nuts.h -- the NUTS transition loop of the one-chain-per-wave sampler (bdrt_wave.h); included by bdrt_nuts.hip after
// NutsArgs / SoloHot.  Same statements as nuts_solo_kernel (Stan 2.19 multinomial NUTS, reference call site
// bayes_drt/inversion.py:1218-1221; behaviour: SURVEY.md Appendix A) on register arrays: slot u of lane l holds element
// wave_slot_index(u, l) of every D-vector.  Global state layout: that of the one-chain-per-workgroup kernel,
// vecs [n_units][SG_COUNT][ds], states [n_units].
#pragma once

namespace bdrt {

typedef __attribute__((address_space(3))) double wv_lds_d;
typedef __attribute__((address_space(1))) double wv_glb_d;

// PROF: the instantiation that fills the phase profile (a kernel of its own: see nuts_kernel, bdrt_nuts16.h)
// OM: the outlier error model's parameters (two per frequency) are slots of the lanes too; one wave per SIMD then (512 registers)
// NB: distributions of the model (wave_eval_nb from two on)
// OCC: waves per SIMD the instantiation is scheduled for (wave_eval, bdrt_wave.h): 1 is launched when a CU gets at most four chains
#ifndef BDRT_WAVE_OM_OCC1
#define BDRT_WAVE_OM_OCC1 1          // the outlier-model instantiations (one wave per SIMD by construction) take the OCC = 1 schedule of the evaluator
#endif
template <int KS, int NS, bool PROF = false, bool OM = false, int NB = 1, int OCC = 2>
__global__ __launch_bounds__(WV_NT, (OM || NB > 1) ? 1 : OCC) void nuts_wave_kernel(const DevProblem *__restrict__ Pp, NutsParams np, NutsArgs a, WaveGeom g, int nhot)
{
    constexpr int NJ = wave_slots_nb<KS, NS, OM, NB>();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int lane = threadIdx.x;
    const int unit = a.unit_map ? a.unit_map[blockIdx.x] : blockIdx.x;
    const int D = g.D, DS = a.ds;
    double *Vg = a.vecs + (size_t)blockIdx.x * SG_COUNT * DS;    // global rows
    double *hot = smem + g.o_hot;

    int jx[NJ];                                            // D-index of each slot, clamped; ok[u]: the slot holds an element
    bool ok[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) { const int j = (NB == 1 ? wave_slot_index<KS, NS, OM>(P, g.K, u, lane) : wave_slot_index_nb<KS, NS, OM, NB>(P, g.K, u, lane)); ok[u] = j >= 0; jx[u] = j >= 0 ? j : 0; }

    // row access: a row is LDS-resident when its rank is below nhot (uniform), else it is read / written where it is in HBM
    auto load_row = [&](int v, double (&o)[NJ]) {
        const int r = wave_hot_rank(v);
        if (r < nhot) {
            const wv_lds_d *b = (const wv_lds_d *)(hot + (size_t)r * g.DSS);
#pragma unroll
            for (int u = 0; u < NJ; ++u) { const double t = b[jx[u]]; o[u] = ok[u] ? t : 0.0; }
        } else {
            const wv_glb_d *b = (const wv_glb_d *)(Vg + (size_t)v * DS);
#pragma unroll
            for (int u = 0; u < NJ; ++u) { const double t = b[jx[u]]; o[u] = ok[u] ? t : 0.0; }
        }
    };
    auto store_row = [&](int v, const double (&x)[NJ]) {
        const int r = wave_hot_rank(v);
        if (r < nhot) {
            wv_lds_d *b = (wv_lds_d *)(hot + (size_t)r * g.DSS);
#pragma unroll
            for (int u = 0; u < NJ; ++u) if (ok[u]) b[jx[u]] = x[u];
        } else {
            wv_glb_d *b = (wv_glb_d *)(Vg + (size_t)v * DS);
#pragma unroll
            for (int u = 0; u < NJ; ++u) if (ok[u]) b[jx[u]] = x[u];
        }
    };
    auto load_grow = [&](int v, double (&o)[NJ]) {         // rows that are never LDS-resident
        const wv_glb_d *b = (const wv_glb_d *)(Vg + (size_t)v * DS);
#pragma unroll
        for (int u = 0; u < NJ; ++u) { const double t = b[jx[u]]; o[u] = ok[u] ? t : 0.0; }
    };
    auto store_grow = [&](int v, const double (&x)[NJ]) {
        wv_glb_d *b = (wv_glb_d *)(Vg + (size_t)v * DS);
#pragma unroll
        for (int u = 0; u < NJ; ++u) if (ok[u]) b[jx[u]] = x[u];
    };

    // the chain's scalar state lives in LDS; the lanes keep its hot part in registers (identical updates)
    ChainState *cold = reinterpret_cast<ChainState *>(smem + g.o_state);
    if (lane == 0) chain_state_copy(*cold, a.states[unit]);         // (member by member: a struct assignment goes through scratch)
    wave_eval_init(P, g, smem, lane);
    for (int v = 0; v < SV_COUNT; ++v) {                   // LDS-resident rows: copied in once
        const int r = wave_hot_rank(v);
        if (r >= nhot) continue;
        for (int j = lane; j < g.DSS; j += WV_NT) hot[(size_t)r * g.DSS + j] = j < D ? Vg[(size_t)v * DS + j] : 0.0;
    }
    wv_sync();
    SoloHot s;
    s.from(*cold);
    const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)cold->chain_id};
    const WaveEvalRegs<NS> er = wave_eval_setup<NS>(P, g, cold->spec, lane);
    double th[NJ], p[NJ], gq[NJ], mi[NJ];
    load_grow(SV_TH, th); load_grow(SV_P, p); load_grow(SV_G, gq); load_grow(SV_MINV, mi);
#pragma unroll
    for (int u = 0; u < NJ; ++u) mi[u] = ok[u] ? mi[u] : 1.0;

    auto with_full_state = [&](auto fn) {
        ChainState full;
        chain_state_copy(full, *cold);
        s.to(full);
        const int r = fn(full);
        s.from(full);
        wv_sync();
        if (lane == 0) chain_state_copy(*cold, full);
        wv_sync();
        return r;
    };

    if (!cold->kicked) {
        const int ph = s.phase;
        const double e = ph == PH_EPS ? s.eps : (ph == PH_TREE ? s.dir * s.eps : 0.0);
        if (ph == PH_INIT || ph == PH_EPS || ph == PH_TREE) {
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                p[u] = p[u] + 0.5 * e * gq[u];
                th[u] += e * mi[u] * p[u];
            }
        }
        wv_sync();
        if (lane == 0) cold->kicked = 1;
        wv_sync();
    }
    unsigned long long my_leaps = 0;
    // uniforms of 64 consecutive leaves of a subtree at once (lane l: leaf 64 b + l), refreshed when the key changes
    double uvec = 0.0;
    int u_iter = -1, u_depth = -1, u_blk = -1;
    long long *prof = (PROF && a.prof) ? a.prof + (size_t)blockIdx.x * 32 : nullptr;

    for (int round = 0; round < a.rounds; ++round) {
        const int ph0 = s.phase;
        const bool act = ph0 == PH_INIT || ph0 == PH_EPS || ph0 == PH_TREE;
        if (!act) break;
        const double e = ph0 == PH_EPS ? s.eps : (ph0 == PH_TREE ? s.dir * s.eps : 0.0);
        // ---- B: log-posterior + gradient at the new point ------------------------------------------------------------------
        if (ph0 == PH_TREE && (s.iter != u_iter || s.depth != u_depth || (s.leaf >> 6) != u_blk)) {
            u_iter = s.iter; u_depth = s.depth; u_blk = s.leaf >> 6;
            uvec = rng_uniform(rng, (uint32_t)(64 * u_blk + lane), RNG_LEAF, (uint32_t)s.depth, 0, (uint32_t)s.iter);
        }
        double lp;
        if constexpr (NB == 1) lp = wave_eval<KS, NS, OM, (OM && BDRT_WAVE_OM_OCC1) ? 1 : OCC>(P, g, smem, th, gq, er, 1.0, lane, prof);
        else lp = wave_eval_nb<KS, NS, OM, NB>(P, g, smem, th, gq, er, 1.0, lane);
        {
            // the slot indices from a lane number the optimiser cannot see through: otherwise every row's per-slot address
            // (37 rows x NJ) is computed once in front of the loop and spilled.  After the evaluation: not live across it.
            int ln = lane;
            __asm__ volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < NJ; ++u) { const int j = (NB == 1 ? wave_slot_index<KS, NS, OM>(P, g.K, u, ln) : wave_slot_index_nb<KS, NS, OM, NB>(P, g.K, u, ln)); ok[u] = j >= 0; jx[u] = j >= 0 ? j : 0; }
        }
        long long tsp = prof ? clock64() : 0;
#define BDRT_WV_NPROF(slot) do { if (prof) { const long long t_ = clock64(); if (lane == 0) prof[slot] += t_ - tsp; tsp = t_; } } while (0)

        // ---- C: second half kick, kinetic energy, finiteness of the gradient ----------------------------------------------------
        double kin = 0.0;
        bool bad = false;
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            p[u] = p[u] + 0.5 * e * gq[u];
            kin = fma(mi[u] * p[u], p[u], kin);
            bad = bad || !isfinite(gq[u]);
        }
        kin = 0.5 * wv_sum(kin);
        const bool nonfin = __ballot(bad) != 0ull;
        BDRT_WV_NPROF(8);

        // ---- S1: scalar logic after the evaluation (identical in every lane) ---------------------------------------------------
        bool copyq = false, cur2s = false, tree = false, last = false;
        bool upds = false, welf = false, wend = false;
        int nm = 0, endt = 0, next = 0, draw = -1;
        double wn = 0.0;
        const int dir_now = s.dir;
        const int leaf_now = s.leaf;
        {
            const bool finite_pt = isfinite(lp) && !nonfin;
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
                    const double u = wv_bcast(uvec, __builtin_amdgcn_readfirstlane(leaf_now & 63));
                    double lsw_new;
                    const bool joins = nuts_leaf_joins(s.lsw_sub, w, u, lsw_new);           // (bdrt_nuts_device.h)
                    if (leaf_now == 0 || joins) { copyq = true; s.lpq = lp; }
                    s.lsw_sub = lsw_new;
                    tree = true;
                    while ((leaf_now >> nm) & 1) ++nm;
                    last = leaf_now == s.nleaves - 1;
                }
            }
        }
        BDRT_WV_NPROF(9);

        // ---- D: proposal copy, checkpoints, U-turn tests, subtree close ----------------------------------------------------------
        if (copyq) { store_row(SV_THQ, th); store_row(SV_GQ, gq); }
        if (cur2s) { store_row(SV_THS, th); store_row(SV_GS, gq); }
        if (tree) {
            // binary-counter bookkeeping of the new subtree: see nuts_kernel (level l: rho / first momentum of the completed
            // left sub-subtree of 2^l leaves that waits for its sibling; level 0 keeps only the momentum)
            double rc[NJ], cpl[NJ];
#pragma unroll
            for (int u = 0; u < NJ; ++u) { rc[u] = p[u]; cpl[u] = p[u]; }
            bool okt = true;
            for (int l = 0; l < nm; ++l) {
                double lpv[NJ], lr[NJ];
                load_row(SV_CKP + l, lpv);
                if (l > 0) load_row(SV_CKC + l, lr);
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    const double rho = (l == 0 ? lpv[u] : lr[u]) + rc[u];
                    a0 = fma(mi[u] * lpv[u], rho, a0);
                    a1 = fma(mi[u] * p[u], rho, a1);
                    rc[u] = rho;
                    cpl[u] = lpv[u];
                }
                wv_sum2(a0, a1, lane);
                okt = okt && (a0 > 0.0) && (a1 > 0.0);
            }
            if (okt && !last) {
                store_row(SV_CKP + nm, cpl);
                if (nm > 0) store_row(SV_CKC + nm, rc);
            }
            if (!okt) {
                endt = 1;
            } else if (last) {
                double po[NJ], rt[NJ];
                load_grow(dir_now > 0 ? SV_PM : SV_PP, po);                 // momentum at the other end
                load_row(SV_RHO, rt);
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    rt[u] += rc[u];
                    t0 = fma(mi[u] * po[u], rt[u], t0);
                    t1 = fma(mi[u] * p[u], rt[u], t1);
                }
                store_row(SV_RHO, rt);
                store_grow(dir_now > 0 ? SV_THP : SV_THM, th);
                store_grow(dir_now > 0 ? SV_PP : SV_PM, p);
                store_grow(dir_now > 0 ? SV_GP : SV_GM, gq);
                wv_sum2(t0, t1, lane);
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
            if (draw >= 0 && a.lp_draws && lane == 0) a.lp_draws[(size_t)unit * np.n_draws + draw] = s.lps;
        }
        BDRT_WV_NPROF(10);

        // ---- A': the trajectory continues from the point just evaluated: half kick + drift of the next leapfrog ---------------------
        if (next == 0 && s.phase == PH_TREE) {
            const double e1 = s.dir * s.eps;
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                p[u] = p[u] + 0.5 * e1 * gq[u];
                th[u] = th[u] + e1 * mi[u] * p[u];
            }
        }
        // ---- E: sample update, metric adaptation, draw output, start of the next leapfrog when the trajectory does not simply
        //      continue (new transition, next doubling, step-size search, re-initialisation) ------------------------------------------
        if (upds || welf || wend || draw >= 0 || next) {
            const uint32_t iter = (uint32_t)s.iter, trial = (uint32_t)s.eps_trials, att = (uint32_t)s.init_attempt;
            double ths[NJ], gs[NJ];
#pragma unroll
            for (int u = 0; u < NJ; ++u) { ths[u] = 0.0; gs[u] = 0.0; }
            if (upds || welf || wend || draw >= 0 || next == 1 || next == 3) {
                if (upds) { load_row(SV_THQ, ths); load_row(SV_GQ, gs); }
                else { load_row(SV_THS, ths); load_row(SV_GS, gs); }
            }
            if (upds) { store_row(SV_THS, ths); store_row(SV_GS, gs); }
            if (welf || wend) {
                double mean[NJ], m2[NJ];
                load_grow(SG_WMEAN, mean); load_grow(SG_WM2, m2);
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    if (welf) {            // Welford (stan::math::welford_var_estimator)
                        const double delta = ths[u] - mean[u];
                        mean[u] += delta / wn;
                        m2[u] += (ths[u] - mean[u]) * delta;
                    }
                    if (wend) {            // var_adaptation::learn_variance
                        const double var = wn > 1.0 ? m2[u] / (wn - 1.0) : 0.0;
                        mi[u] = ok[u] ? (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0)) : 1.0;
                        mean[u] = 0.0; m2[u] = 0.0;
                    }
                }
                store_grow(SG_WMEAN, mean); store_grow(SG_WM2, m2);
            }
            if (draw >= 0) {
                wv_glb_d *dr = (wv_glb_d *)(a.draws + ((size_t)unit * np.n_draws + draw) * D);
#pragma unroll
                for (int u = 0; u < NJ; ++u) if (ok[u]) dr[jx[u]] = ths[u];
            }
            if (next == 1 || next == 3) {
                // fresh momentum p ~ N(0, M): element j is normal j of the stream = branch j & 1 of Philox block j >> 1 (same streams as
                // nuts_kernel / nuts_solo_kernel)
                double kin0 = 0.0;
                double pn[NJ];
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    double z0, z1;
                    rng_normal_pair(rng, (uint32_t)(jx[u] >> 1), next == 1 ? RNG_MOMENTUM : RNG_EPS_MOMENTUM, next == 1 ? 0u : trial, iter, z0, z1);
                    const double z = (jx[u] & 1) ? z1 : z0;
                    pn[u] = ok[u] ? z / sqrt(mi[u]) : 0.0;
                    kin0 = fma(mi[u] * pn[u], pn[u], kin0);
                }
                kin0 = wv_sum(kin0);
                s.H0 = -s.lps + 0.5 * kin0;
                if (next == 1) {
                    s.lsw = 0.0; s.lsw_sub = -INFINITY; s.depth = 0; s.leaf = 0; s.nleaves = 1;
                    s.n_leap_iter = 0; s.sum_metro = 0.0;
                    s.dir = rng_uniform(rng, 0, RNG_DIRECTION, 0, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                }
                const double e1 = next == 1 ? s.dir * s.eps : s.eps;
                if (next == 1) {
                    store_grow(SV_THM, ths); store_grow(SV_THP, ths);
                    store_grow(SV_PM, pn); store_grow(SV_PP, pn);
                    store_grow(SV_GM, gs); store_grow(SV_GP, gs);
                    store_row(SV_RHO, pn);
                }
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    p[u] = pn[u] + 0.5 * e1 * gs[u];
                    th[u] = ths[u] + e1 * mi[u] * p[u];
                }
            } else if (next == 2) {
                // continue from the trajectory end in the new direction
                const int dir = s.dir;
                const double e1 = dir * s.eps;
                if (dir != dir_now) {
                    load_grow(dir > 0 ? SV_THP : SV_THM, th);
                    load_grow(dir > 0 ? SV_PP : SV_PM, p);
                    load_grow(dir > 0 ? SV_GP : SV_GM, gq);
                }
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    p[u] = p[u] + 0.5 * e1 * gq[u];
                    th[u] = th[u] + e1 * mi[u] * p[u];
                }
            } else if (next == 4) {
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    const double r = np.init_radius * (2.0 * rng_uniform(rng, (uint32_t)jx[u], RNG_INIT, 0, att, 0) - 1.0);
                    th[u] = ok[u] ? r : 0.0;
                    p[u] = 0.0;
                }
            }
        }
        BDRT_WV_NPROF(11);
#undef BDRT_WV_NPROF
    }

    // ---- write the chain back ---------------------------------------------------------------------------------------------------
    store_grow(SV_TH, th); store_grow(SV_P, p); store_grow(SV_G, gq); store_grow(SV_MINV, mi);
    wv_sync();
    for (int v = 0; v < SV_COUNT; ++v) {
        const int r = wave_hot_rank(v);
        if (r >= nhot) continue;
        for (int j = lane; j < D; j += WV_NT) Vg[(size_t)v * DS + j] = hot[(size_t)r * g.DSS + j];
    }
    if (lane == 0) {
        ChainState full;
        chain_state_copy(full, *cold);
        s.to(full);
        chain_state_copy(a.states[unit], full);
        if (my_leaps) atomicAdd(a.leap_counter, my_leaps);
        const int ph = s.phase;
        if (!(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE)) atomicAdd(a.done_counter, 1);
    }
}

// evaluator of the one-chain-per-wave path on its own (parity tests; few-point batches): a wave per point, grid-stride
template <int KS, int NS, bool OM = false, int NB = 1, int OCC = 2>
__global__ __launch_bounds__(WV_NT, (OM || NB > 1) ? 1 : OCC) void wave_eval_kernel(const DevProblem *__restrict__ Pp, WaveGeom g, const double *theta,
                                                             const int *spec, int B, int jacobian, double *lp, double *grad)
{
    constexpr int NJ = wave_slots_nb<KS, NS, OM, NB>();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int lane = threadIdx.x;
    wave_eval_init(P, g, smem, lane);
    wv_sync();
    int jx[NJ];
    bool ok[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) { const int j = (NB == 1 ? wave_slot_index<KS, NS, OM>(P, g.K, u, lane) : wave_slot_index_nb<KS, NS, OM, NB>(P, g.K, u, lane)); ok[u] = j >= 0; jx[u] = j >= 0 ? j : 0; }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        double th[NJ], gr[NJ];
#pragma unroll
        for (int u = 0; u < NJ; ++u) { const double t = theta[(size_t)b * g.D + jx[u]]; th[u] = ok[u] ? t : 0.0; }
        const WaveEvalRegs<NS> er = wave_eval_setup<NS>(P, g, spec ? spec[b] : 0, lane);
        double v;
        if constexpr (NB == 1) v = wave_eval<KS, NS, OM, OCC>(P, g, smem, th, gr, er, jacobian ? 1.0 : 0.0, lane);
        else v = wave_eval_nb<KS, NS, OM, NB>(P, g, smem, th, gr, er, jacobian ? 1.0 : 0.0, lane);
        if (grad) {
#pragma unroll
            for (int u = 0; u < NJ; ++u) if (ok[u]) grad[(size_t)b * g.D + jx[u]] = gr[u];
        }
        if (lane == 0 && lp) lp[b] = v;
        wv_sync();
    }
}

}  // namespace bdrt
