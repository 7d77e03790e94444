// bdrt_wave.hip -- the one-chain-per-wave sampler and evaluator kernels (bdrt_wave.h, bdrt_wave_nuts.h) and their launchers.
// A translation unit of its own so that it builds beside bdrt_nuts.hip (the 16-chain and one-chain-per-workgroup kernels).
// Replaces StanModel.sampling for the headline family (reference bayes_drt/inversion.py:1218-1221).
#include "bdrt_host.h"
#include "bdrt_nuts_device.h"
#include "bdrt_nuts_args.h"
#include "bdrt_wave.h"
#include "bdrt_wave_nuts.h"

namespace bdrt {

#define BDRT_WAVE_DISPATCH(KS_, NS_, CALL)                                  \
    do {                                                                   \
        if (KS_ == 1 && NS_ == 1) { CALL(1, 1); }                          \
        else if (KS_ == 1) { CALL(1, 2); }                                 \
        else if (KS_ == 2 && NS_ == 1) { CALL(2, 1); }                     \
        else if (KS_ == 2) { CALL(2, 2); }                                 \
        else if (NS_ == 1) { CALL(3, 1); }                                 \
        else { CALL(3, 2); }                                               \
    } while (0)

int launch_wave_nuts_nb(const DevProblem *dp, const NutsParams &np, const NutsArgs &args, const WaveGeom &g, int nhot, int n_wg, size_t lds,
                        hipStream_t stream, int outlier_model);
int launch_wave_eval_nb(const DevProblem *dp, const WaveGeom &g, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                        double *d_grad, int n_wg, size_t lds, hipStream_t stream, int outlier_model);

static hipError_t wave_set_lds_limit(size_t bytes)
{
    static LdsAttrCache cache;
    return cache.ensure(bytes, [&]() {
        hipError_t e = hipSuccess;
#define BDRT_WV_ATTR(KS_, NS_)                                                                                                        \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wave_eval_kernel<KS_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wave_eval_kernel<KS_, NS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_, false, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_, true, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wave_eval_kernel<KS_, NS_, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        BDRT_WV_ATTR(1, 1) BDRT_WV_ATTR(1, 2) BDRT_WV_ATTR(2, 1) BDRT_WV_ATTR(2, 2) BDRT_WV_ATTR(3, 1) BDRT_WV_ATTR(3, 2)
#undef BDRT_WV_ATTR
        return e;
    });
}

// LDS request of a launch with `n_wg` chains on `n_cu` CUs: the chain's share of its CU (the request doubles as a placement
// hint: the dispatcher fills a CU as far as the resources allow before it goes to the next), at most eight chains per CU;
// *nhot: the rows that fit beside the evaluator's scratch
size_t wave_lds_request(const WaveGeom &g, int n_wg, int n_cu, int *nhot, int max_per_cu)
{
    int c = (n_wg + n_cu - 1) / n_cu;
    if (const char *e = getenv("BDRT_WAVE_PER_CU")) c = atoi(e);          // diagnostics: force the packing
    c = c < 1 ? 1 : (c > max_per_cu ? max_per_cu : c);
    const size_t need0 = wave_lds_bytes(g, 0);
    // (LDS is allocated in granules -- 1280 bytes assumed: a share that rounds up past its c-th of the CU costs a whole turn of the
    //  machine, measured at 3 chains per CU: 28.8 us per round instead of 14.4)
    const size_t granule = 1280;
    size_t share = ((size_t)(160 * 1024) / c / granule) * granule - 128;
    if (share > (size_t)100 * 1024) share = (size_t)100 * 1024;            // (one chain per CU: more than half of the LDS keeps the next chain off this CU)
    if (share < need0) share = need0;
    int n = (int)((share - need0) / ((size_t)g.DSS * sizeof(double)));
    if (n > WV_HOT_MAX) n = WV_HOT_MAX;
    if (const char *e = getenv("BDRT_WAVE_HOT")) { const int f = atoi(e); if (f >= 0 && f < n) n = f; }   // diagnostics: fewer LDS-resident rows
    *nhot = n;
    return share;
}

// Which schedule of the headline family's kernels a launch takes: the one for ONE wave per SIMD (OCC = 1 instantiations, bdrt_wave.h) when
// the LDS share of a chain leaves room for at most four chains on a CU -- wave_lds_request hands out such shares exactly when the launch
// has at most four chains per CU.  BDRT_WAVE_OCC=1 / 2 forces one (tests, A/B runs).
static bool wave_one_per_simd(size_t lds)
{
    if (const char *e = getenv("BDRT_WAVE_OCC")) { if (e[0] == '1') return true; if (e[0] == '2') return false; }
    return lds * 5 > (size_t)160 * 1024;
}

int launch_wave_nuts(const DevProblem *dp, const NutsParams &np, const NutsArgs &args, const WaveGeom &g, int nhot, int n_wg, size_t lds,
                     hipStream_t stream, int outlier_model)
{
    if (g.nb > 1) return launch_wave_nuts_nb(dp, np, args, g, nhot, n_wg, lds, stream, outlier_model);     // (bdrt_wave_nb.hip)
    BDRT_HIP(wave_set_lds_limit(lds));
    const bool one = wave_one_per_simd(lds);
#define BDRT_WV_CALL1(KS_, NS_) hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_, false, false, 1, 1>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot)
#define BDRT_WV_CALL1_PROF(KS_, NS_) hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_, true, false, 1, 1>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot)
#define BDRT_WV_CALL(KS_, NS_) hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot)
#define BDRT_WV_CALL_PROF(KS_, NS_) hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_, true>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot)
#define BDRT_WV_CALL_OM(KS_, NS_) hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_, false, true>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot)
    if (outlier_model) BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL_OM);     // (no profiling instantiation of the outlier variant)
    else if (args.prof && one) BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL1_PROF);
    else if (args.prof) BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL_PROF);
    else if (one) BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL1);
    else BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL);
#undef BDRT_WV_CALL
#undef BDRT_WV_CALL_PROF
#undef BDRT_WV_CALL_OM
#undef BDRT_WV_CALL1
#undef BDRT_WV_CALL1_PROF
    BDRT_HIP(hipGetLastError());
    return 0;
}

int launch_wave_eval(const DevProblem *dp, const WaveGeom &g, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                     double *d_grad, int n_wg, size_t lds, hipStream_t stream, int outlier_model)
{
    if (g.nb > 1) return launch_wave_eval_nb(dp, g, d_theta, d_spec, B, jacobian, d_lp, d_grad, n_wg, lds, stream, outlier_model);
    BDRT_HIP(wave_set_lds_limit(lds));
#define BDRT_WV_CALL(KS_, NS_) hipLaunchKernelGGL((wave_eval_kernel<KS_, NS_>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, g, d_theta, d_spec, B, jacobian, d_lp, d_grad)
#define BDRT_WV_CALL_OM(KS_, NS_) hipLaunchKernelGGL((wave_eval_kernel<KS_, NS_, true>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, g, d_theta, d_spec, B, jacobian, d_lp, d_grad)
#define BDRT_WV_CALL1(KS_, NS_) hipLaunchKernelGGL((wave_eval_kernel<KS_, NS_, false, 1, 1>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, g, d_theta, d_spec, B, jacobian, d_lp, d_grad)
    const char *occ_env = getenv("BDRT_WAVE_OCC");                 // (the evaluator on its own runs a grid-stride loop on a full machine: the default schedule unless forced)
    if (outlier_model) BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL_OM);
    else if (occ_env && occ_env[0] == '1') BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL1);
    else BDRT_WAVE_DISPATCH(g.KS, g.NS, BDRT_WV_CALL);
#undef BDRT_WV_CALL
#undef BDRT_WV_CALL_OM
#undef BDRT_WV_CALL1
    BDRT_HIP(hipGetLastError());
    return 0;
}

}  // namespace bdrt
