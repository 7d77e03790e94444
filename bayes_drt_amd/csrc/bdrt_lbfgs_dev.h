// bdrt_lbfgs_dev.h -- the Stan-style L-BFGS of bdrt_lbfgs.h as ONE kernel launch: a workgroup per fit, the whole iteration on
// the device (replaces the host round trip per evaluation of `StanModel.optimizing`, reference bayes_drt/inversion.py:1216).
//
// The decisions are `LbfgsCore` (bdrt_lbfgs.h) itself, instantiated on a device vector back end: element j of every vector
// lives in thread j (two elements per thread for the wide parameter vectors), the history pairs in rows of LDS (headline
// family) or HBM (general block model) that only the owning thread touches, dot products are workgroup reductions in a fixed
// order, and the scalar state machine runs redundantly -- and identically -- in every thread.  Evaluations: the one-chain
// evaluators of bdrt_solo.h (S1 family on log-uniform grids) and bdrt_solo_wide.h (every other Toeplitz-capable model).
// ~8 us per iteration instead of ~70 us host-driven.
//
// (Included by bdrt_nuts.hip inside namespace bdrt, after bdrt_solo.h and bdrt_solo_wide.h.)
#pragma once

struct LbfgsDevReport {
    int iters, n_evals, rc, resets;
    double f;
};

template <int NE>
struct DevVecs {
    double x[NE], g[NE], xt[NE], gt[NE], p[NE], q[NE];
    bool own[NE];
    double *hist;               // [2 * LBFGS_MAX_HISTORY][stride]: S rows, then Y rows (each thread reads and writes its own elements)
    int stride;
    double *red;                // two rotating slots of 8-wave partials (solo_block_sum2)
    int rslot, wave, lane, tid;

    __device__ __forceinline__ double *S(int slot, int k) const { return hist + (size_t)slot * stride + tid + SOLO_NT * k; }
    __device__ __forceinline__ double *Y(int slot, int k) const { return hist + (size_t)(LBFGS_MAX_HISTORY + slot) * stride + tid + SOLO_NT * k; }
    __device__ __forceinline__ void bsum2(double &a, double &b) { solo_block_sum2(a, b, red, rslot, wave, lane); }
    __device__ __forceinline__ double bsum(double a) { double b = 0.0; bsum2(a, b); return a; }

    __device__ __forceinline__ double dot_g_p() { double a = 0.0; for (int k = 0; k < NE; ++k) a += own[k] ? g[k] * p[k] : 0.0; return bsum(a); }
    __device__ __forceinline__ double dot_g_g() { double a = 0.0; for (int k = 0; k < NE; ++k) a += own[k] ? g[k] * g[k] : 0.0; return bsum(a); }
    __device__ __forceinline__ double dot_gt_p(bool &finite)
    {
        double a = 0.0, bad = 0.0;
        for (int k = 0; k < NE; ++k) if (own[k]) { a += gt[k] * p[k]; bad += isfinite(gt[k]) ? 0.0 : 1.0; }
        if (bad != 0.0) a = 0.0;                       // (keeps the other threads' sum finite; the value is not used then)
        bsum2(a, bad);
        finite = bad == 0.0;
        return finite ? a : 0.0;
    }
    __device__ __forceinline__ void p_minus_g() { for (int k = 0; k < NE; ++k) p[k] = -g[k]; }
    __device__ __forceinline__ void set_trial(double a) { for (int k = 0; k < NE; ++k) xt[k] = x[k] + a * p[k]; }
    __device__ __forceinline__ void accept(int slot, double &sy, double &yy, double &ss)
    {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < NE; ++k) if (own[k]) {
            const double s = xt[k] - x[k], y = gt[k] - g[k];
            *S(slot, k) = s; *Y(slot, k) = y;
            a += s * y; b += y * y; c += s * s;
            x[k] = xt[k]; g[k] = gt[k];
        }
        bsum2(a, b);
        c = bsum(c);
        sy = a; yy = b; ss = c;
    }
    __device__ __forceinline__ void q_from_g() { for (int k = 0; k < NE; ++k) q[k] = g[k]; }
    __device__ __forceinline__ double dot_S_q(int slot) { double a = 0.0; for (int k = 0; k < NE; ++k) a += own[k] ? *S(slot, k) * q[k] : 0.0; return bsum(a); }
    __device__ __forceinline__ double dot_Y_q(int slot) { double a = 0.0; for (int k = 0; k < NE; ++k) a += own[k] ? *Y(slot, k) * q[k] : 0.0; return bsum(a); }
    __device__ __forceinline__ void q_axpy_Y(int slot, double c) { for (int k = 0; k < NE; ++k) if (own[k]) q[k] += c * *Y(slot, k); }
    __device__ __forceinline__ void q_axpy_S(int slot, double c) { for (int k = 0; k < NE; ++k) if (own[k]) q[k] += c * *S(slot, k); }
    __device__ __forceinline__ void q_scale(double c) { for (int k = 0; k < NE; ++k) q[k] *= c; }
    __device__ __forceinline__ void p_minus_q() { for (int k = 0; k < NE; ++k) p[k] = -q[k]; }
};

// rows of global work space per fit of the general-model variant: theta, gradient, history
constexpr int LBFGS_WIDE_ROWS = 2 + 2 * LBFGS_MAX_HISTORY;

template <bool WIDE>
__global__ __launch_bounds__(SOLO_NT) void lbfgs_kernel(const DevProblem *__restrict__ Pp, SoloGeom g, Wide1Geom G, const double *x0,
                                                        const int *spec, bdrt_opt_options o, int max_evals, double *x_out,
                                                        double *g_out, LbfgsDevReport *rep, double *work, int DS)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NE = WIDE ? 2 : 1;
    typedef LbfgsCore<DevVecs<NE>> Core;
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x, fit = blockIdx.x, D = P.D;
    const int sp = spec ? spec[fit] : 0;
    Core c;
    c.opt = o;
    DevVecs<NE> &v = c.v;
    v.tid = tid; v.lane = tid & 63; v.wave = tid >> 6; v.rslot = 0;
    double *TH, *GR, *lpv;
    SoloEvalRegs ers;
    Wide1Regs erw;
    if constexpr (WIDE) {
        wide1_init(P, G, smem, tid);
        erw = wide1_setup(P, G, sp, tid);
        double *w = work + (size_t)fit * LBFGS_WIDE_ROWS * DS;
        TH = w; GR = w + DS; v.hist = w + 2 * (size_t)DS; v.stride = DS;
        lpv = smem + G.total; v.red = smem + G.o_red;
    } else {
        solo_eval_init(P, g, smem, tid);
        ers = solo_eval_setup(P, g, sp, tid);
        TH = smem + g.o_vec; GR = TH + g.DSS; v.hist = GR + g.DSS; v.stride = g.DSS;
        lpv = smem + g.o_scv + 12; v.red = smem + g.o_red;
    }
    for (int k = 0; k < NE; ++k) {
        const int e = tid + SOLO_NT * k;
        v.own[k] = e < D;
        v.x[k] = v.own[k] ? x0[(size_t)fit * D + e] : 0.0;
        v.g[k] = 0.0; v.xt[k] = v.x[k]; v.gt[k] = 0.0; v.p[k] = 0.0; v.q[k] = 0.0;
    }
    __syncthreads();
    // log-posterior (no Jacobian: Stan's `optimizing`) and gradient at `pt`; returns f = -lp, grad = -grad lp
    auto evaluate = [&](const double (&pt)[NE], double (&grad)[NE]) -> double {
        for (int k = 0; k < NE; ++k) if (v.own[k]) TH[tid + SOLO_NT * k] = pt[k];
        __syncthreads();
        if constexpr (WIDE) wide1_eval(P, G, smem, TH, GR, lpv, erw, 0, tid);
        else solo_eval(P, g, smem, TH, GR, lpv, ers, 0, tid);
        const double lp = *lpv;
        for (int k = 0; k < NE; ++k) grad[k] = v.own[k] ? -GR[tid + SOLO_NT * k] : 0.0;
        __syncthreads();
        return -lp;
    };
    c.start(evaluate(v.x, v.g));
    while (c.phase != Core::DONE && c.n_evals < max_evals) c.feed_trial(evaluate(v.xt, v.gt));
    if (c.phase != Core::DONE) c.rc = 1;
    for (int k = 0; k < NE; ++k)
        if (v.own[k]) {
            x_out[(size_t)fit * D + tid + SOLO_NT * k] = v.x[k];
            g_out[(size_t)fit * D + tid + SOLO_NT * k] = v.g[k];
        }
    if (tid == 0) {
        LbfgsDevReport r;
        r.iters = c.iters; r.n_evals = c.n_evals; r.rc = c.rc; r.resets = c.ls_fail_resets; r.f = c.f;
        rep[fit] = r;
    }
}
