// bdrt_wave_nb.hip -- the one-chain-per-wave kernels (bdrt_wave.h, bdrt_wave_nuts.h) for the models of several distributions
// (Series-Parallel, Series-2Parallel, with or without the outlier error models: wave_eval_nb).  A translation unit of its own: sixteen
// instantiations of the sampler kernel build beside the others.  Replaces StanModel.sampling for those model files at mid occupancy
// (reference bayes_drt/inversion.py:1218-1221, stan_model_files/Series-Parallel*_modelcode.txt).
#include "bdrt_host.h"
#include "bdrt_nuts_device.h"
#include "bdrt_nuts_args.h"
#include "bdrt_wave.h"
#include "bdrt_wave_nuts.h"

namespace bdrt {

// basis length 65 .. 192 (KS 2 or 3), any frequency count up to 128 (NS 1 or 2), two or three blocks, with / without outlier parameters
#define BDRT_WNB_SHAPES(X, OM_, NB_) X(2, 1, OM_, NB_) X(2, 2, OM_, NB_) X(3, 1, OM_, NB_) X(3, 2, OM_, NB_)
#define BDRT_WNB_ALL(X) BDRT_WNB_SHAPES(X, false, 2) BDRT_WNB_SHAPES(X, true, 2) BDRT_WNB_SHAPES(X, false, 3) BDRT_WNB_SHAPES(X, true, 3)

static hipError_t wave_nb_set_lds_limit(size_t bytes)
{
    static LdsAttrCache cache;
    return cache.ensure(bytes, [&]() {
        hipError_t e = hipSuccess;
#define BDRT_WNB_ATTR(KS_, NS_, OM_, NB_)                                                                                                       \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nuts_wave_kernel<KS_, NS_, false, OM_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)wave_eval_kernel<KS_, NS_, OM_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        BDRT_WNB_ALL(BDRT_WNB_ATTR)
#undef BDRT_WNB_ATTR
        return e;
    });
}

int launch_wave_nuts_nb(const DevProblem *dp, const NutsParams &np, const NutsArgs &args, const WaveGeom &g, int nhot, int n_wg, size_t lds,
                        hipStream_t stream, int outlier_model)
{
    BDRT_HIP(wave_nb_set_lds_limit(lds));
    bool done = false;
#define BDRT_WNB_CALL(KS_, NS_, OM_, NB_)                                                                                                       \
    if (!done && g.KS == KS_ && g.NS == NS_ && (outlier_model != 0) == OM_ && g.nb == NB_) {                                                      \
        hipLaunchKernelGGL((nuts_wave_kernel<KS_, NS_, false, OM_, NB_>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, np, args, g, nhot); done = true; }
    BDRT_WNB_ALL(BDRT_WNB_CALL)
#undef BDRT_WNB_CALL
    if (!done) { set_error("one-chain-per-wave kernel: no instantiation for KS %d, NS %d, %d blocks", g.KS, g.NS, g.nb); return -2; }
    BDRT_HIP(hipGetLastError());
    return 0;
}

int launch_wave_eval_nb(const DevProblem *dp, const WaveGeom &g, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                        double *d_grad, int n_wg, size_t lds, hipStream_t stream, int outlier_model)
{
    BDRT_HIP(wave_nb_set_lds_limit(lds));
    bool done = false;
#define BDRT_WNB_CALL(KS_, NS_, OM_, NB_)                                                                                                       \
    if (!done && g.KS == KS_ && g.NS == NS_ && (outlier_model != 0) == OM_ && g.nb == NB_) {                                                      \
        hipLaunchKernelGGL((wave_eval_kernel<KS_, NS_, OM_, NB_>), dim3(n_wg), dim3(WV_NT), lds, stream, dp, g, d_theta, d_spec, B, jacobian, d_lp, d_grad); done = true; }
    BDRT_WNB_ALL(BDRT_WNB_CALL)
#undef BDRT_WNB_CALL
    if (!done) { set_error("one-chain-per-wave evaluator: no instantiation for KS %d, NS %d, %d blocks", g.KS, g.NS, g.nb); return -2; }
    BDRT_HIP(hipGetLastError());
    return 0;
}

}  // namespace bdrt
