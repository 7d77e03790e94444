// bdrt_nuts_args.h -- launch arguments and register-resident chain state shared by the sampler kernels' translation units
// (bdrt_nuts.hip: 16-chain and one-chain-per-workgroup kernels, host side; bdrt_wave.hip: one-chain-per-wave kernel).
#pragma once
#include "bdrt_nuts_device.h"

namespace bdrt {

struct NutsArgs {
    double *vecs;          // [n_wg][V_COUNT][16 chains][ds]  (one contiguous row per chain and vector)
    ChainState *states;    // [n_units]
    double *draws;         // [n_units][n_draws][D]
    double *lp_draws;      // [n_units][n_draws]
    unsigned long long *leap_counter;   // total leapfrogs (all chains)
    int *done_counter;     // workgroups whose chains are all finished
    int n_units;
    const int *slot_unit;  // 16-chain kernel: [n_wg][16] unit held by slot k of workgroup wg (-1: empty).  Unit <-> slot is an
                           // indirection so that live chains can be re-packed into fewer workgroups during a run (compaction)
    int cpw;               // chains per workgroup at creation (1..16): few chains are spread over many workgroups / waves
    int rounds;
    int ds;                // row stride of the state vectors (D rounded up to 32)
    long long *prof;       // optional [n_wg][32] cycle counters (phase profile)
    const int *unit_map;   // one-chain-per-workgroup kernel: unit of workgroup b (nullptr: b) -- the tail of a large run (below)
    int *active_counter;   // 16-chain kernel: chains still running at the end of the launch (all workgroups)
    double *bigws;         // problems beyond the LDS budget (bdrt_big.h): [n_wg][big_ws_doubles] workspace of the evaluator
};

// The part of a chain's scalar state that a leapfrog touches: kept in registers by every thread (identical updates).  The
// rest (adaptation windows, dual averaging, counters) stays in LDS and is visited when a transition ends or the step size is
// searched; same member names as ChainState, so the statements of the three samplers read alike.
struct SoloHot {
    int phase, iter, depth, leaf, nleaves, dir, n_leap_iter, init_attempt, eps_dir, eps_trials;
    double eps, H0, lsw, lsw_sub, lps, lpq, sum_metro;
    __device__ __forceinline__ void from(const ChainState &c)
    {
        phase = c.phase; iter = c.iter; depth = c.depth; leaf = c.leaf; nleaves = c.nleaves; dir = c.dir; n_leap_iter = c.n_leap_iter;
        init_attempt = c.init_attempt; eps_dir = c.eps_dir; eps_trials = c.eps_trials;
        eps = c.eps; H0 = c.H0; lsw = c.lsw; lsw_sub = c.lsw_sub; lps = c.lps; lpq = c.lpq; sum_metro = c.sum_metro;
    }
    __device__ __forceinline__ void to(ChainState &c) const
    {
        c.phase = phase; c.iter = iter; c.depth = depth; c.leaf = leaf; c.nleaves = nleaves; c.dir = dir; c.n_leap_iter = n_leap_iter;
        c.init_attempt = init_attempt; c.eps_dir = eps_dir; c.eps_trials = eps_trials;
        c.eps = eps; c.H0 = H0; c.lsw = lsw; c.lsw_sub = lsw_sub; c.lps = lps; c.lpq = lpq; c.sum_metro = sum_metro;
    }
};

}  // namespace bdrt
