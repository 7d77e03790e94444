// bdrt_host.h -- host-side objects shared by the translation units of libbdrt.so
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bdrt.h"
#include "bdrt_device.h"
#include "bdrt_tile_s1.h"
#include "bdrt_tile_hw.h"

namespace bdrt {

void set_error(const char *fmt, ...);

#define BDRT_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            bdrt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -10;                                                                             \
        }                                                                                           \
    } while (0)

// The HIP current device is per host thread.  bdrt_set_device records the process-wide device; entry points that take no
// handle (bdrt_build_*, bdrt_gram, bdrt_qp_box_batch, bdrt_percentiles) bind the calling thread to it, entry points that
// take a problem / sampler bind to the device the problem was created on.
extern std::atomic<int> g_process_device;
inline void bind_process_device()
{
    const int d = g_process_device.load(std::memory_order_relaxed);
    if (d >= 0) hipSetDevice(d);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: high-water mark per device id, guarded by a mutex.
struct LdsAttrCache {
    std::mutex mu;
    size_t hw[64] = {};
    template <class F>
    hipError_t ensure(size_t bytes, F set_attrs)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> g(mu);
        if (dev >= 0 && dev < 64 && bytes <= hw[dev]) return hipSuccess;
        e = set_attrs();
        if (e == hipSuccess && dev >= 0 && dev < 64) hw[dev] = bytes;
        return e;
    }
};

struct Problem {
    DevProblem dev;                 // device view (pointers are device pointers), host copy
    DevProblem *d_dev = nullptr;    // the same struct in HBM: kernels take it by pointer (uniform scalar loads)
    int sync_dev();                 // upload `dev` to d_dev
    std::vector<void *> allocs;     // owned device allocations
    std::vector<unsigned char> is_pos;
    hipStream_t stream = nullptr;
    size_t lds_bytes = 0;
    int device = 0;
    double *d_Z = nullptr;
    size_t z_capacity = 0;          // doubles
    // host copies of the layout
    int o_x[MAXB], o_ups[MAXB], o_d[MAXB];
    // scratch buffers for host-pointer entry points (grown on demand)
    double *d_theta = nullptr, *d_grad = nullptr, *d_lp = nullptr;
    int *d_spec = nullptr;
    size_t scratch_rows = 0;
    int ensure_scratch(size_t rows);
    // launch_logp_grad_few: which one-workgroup-per-point evaluator the problem takes (-1 not looked at yet, 0 none, 1 the
    // headline family's, 2 the general one), decided once -- the call is launch-latency-bound, host microseconds count
    int few_kind = -1, few_ncu = 0;
    // problems beyond the LDS budget (bdrt_big.h): per-point workspace of the evaluator, grown on demand
    // (ONE buffer per problem, indexed by workgroup: launches on different streams take turns -- each waits for the event the
    // one before it recorded --, and the bookkeeping is under a mutex)
    double *d_bigws = nullptr;
    size_t bigws_doubles = 0;
    hipEvent_t bigws_done = nullptr;
    std::mutex bigws_mu;
    int ensure_bigws(size_t doubles);
};

// launch the batched evaluator on device buffers (theta/grad [B x D] row-major)
int launch_logp_grad(Problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                     double *d_grad, double *d_params, double *d_Zhat, double *d_sig, hipStream_t stream);

// Few points (B <= #CUs): one WORKGROUP per point with the Toeplitz / banded structure of the problem instead of one
// 16-column MFMA tile with B live columns (bdrt_solo.h / bdrt_solo_wide.h evaluators; defined in bdrt_nuts.hip).  Returns 1
// when the problem / batch does not take that path (nothing launched), 0 when launched, < 0 on error.  Same formulas, other
// summation order: results agree with the tile evaluator to ~1e-13 relative, not bit for bit.
int launch_logp_grad_few(Problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp, double *d_grad,
                         hipStream_t stream, int any_b = 0);

// the Stan-style L-BFGS of n fits as one launch (bdrt_lbfgs_dev.h; defined in bdrt_nuts.hip); 1: not applicable to this problem
int lbfgs_device(Problem &P, const double *x0, const int *spec, int n, const bdrt_opt_options &o, double *x_out, double *g_out,
                 int *iters, int *n_evals, int *rc, double *f);

// percentiles of X Phi^T + bias (Phi == nullptr: of X itself) over the rows of a DEVICE matrix X; Phi, bias, q, out: host
// expcol[K] (host, Phi == nullptr only): columns whose samples are exp(X); mean[ncols] (host): sample means; both optional
int post_percentiles_device(const double *dX, int rows, int K, long ldx, const double *Phi, int M, const double *bias,
                            const double *q, int nq, double *out, const unsigned char *expcol = nullptr,
                            double *mean = nullptr);

// Levenberg-Marquardt Newton polish of n_fits points on the device (bdrt_newton.hip); x0 / x_out [n_fits][D] on the host
int newton_polish_device(Problem &P, const double *x0, const int *spec, int n_fits, int max_iter, double tol, double *x_out,
                         double *lp_out, double *ginf_out, int *iters_out, int *rc_out, int *n_evals_out);

// closed-form Hessian at one point (bdrt_newton_hess.h; tests): 1 when the problem has none
int hessian_at_point(Problem &P, const double *theta, int spec, double *H_out);

}  // namespace bdrt

struct bdrt_problem {
    bdrt::Problem impl;
};
