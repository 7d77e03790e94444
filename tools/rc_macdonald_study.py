"""RC_Macdonald_0.25: why does the posterior mean of this one spectrum of the published HMC study sit 17-18 % from
the stored curve, in every seed, while its three siblings (RC_Orazem / uniform / noiseless) land within 0.2 %?
 (1) HIP = oracle?  Inverter.fit 4 x (1000 + 1000) against the CPU oracle's long run (tests/golden/rc_macdonald_oracle.npz,
     tools/rc_macdonald_oracle.py): coefficient posterior means.
 (2) The stored curves against each other: stored HMC mean vs the stored MAP curve of the same spectrum (the siblings: 3-5 %).
 (3) The reference's run had ONE of its two chains saturated at tree depth 10 for all 200 draws (notebook diagnostics): a chain that
     does not move.  If the stored result is the 50/50 mixture of a converged chain and a chain frozen at some point g_s, then
     g_s = 2 * stored_mean - (our mean) must be a plausible distribution (non-negative, unit area) and the stored 2.5 % / 97.5 % curves
     must be reproduced by the percentiles of {200 x g_s} u {200 draws of ours}: checked here.
GPU box.  Usage: python tools/rc_macdonald_study.py"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

STEM = 'RC_Macdonald_0.25'
S = load('hmc_suite')
stems = [str(s) for s in S['stems']]
tau_plot = np.logspace(-7, 2, 200)


def mixture_check(stem, seed=1234, verbose=True):
    i = stems.index(stem)
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    ref, gmap = S['Gout_bayes'][i], S['Gout_map'][i][:, 1]
    inv = Inverter(basis_freq=f)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=200, samples=200, chains=2, random_seed=seed)
    fit = inv._sample_result
    ours = inv.predict_distribution('DRT', eval_tau=tau_plot)
    # coefficient draws of ONE converged chain; gamma(tau) = Phi(tau) x, and the reference's percentile curves are Phi applied to the
    # per-coefficient percentiles (coef_percentile, reference :2517-2519 / :3298-3311)
    from bayes_drt_amd.inversion import _gaussian
    info = inv.distributions['DRT']
    Phi = _gaussian(np.log(tau_plot[:, None] / info['tau'][None, :]), info['epsilon'])
    X = inv._rescale_coef(fit.chain_draws(inv._get_stan_coef_name('DRT'))[0], 'series')       # [200, K]
    x_stored = np.linalg.lstsq(Phi, ref[:, 1], rcond=None)[0]                                    # coefficients of the stored mean curve
    x_s = 2.0 * x_stored - X.mean(axis=0)                                                        # the frozen chain, if the hypothesis holds
    g_s = Phi @ x_s
    mix = np.vstack([X, np.tile(x_s, (X.shape[0], 1))])
    lo, hi = Phi @ np.percentile(mix, 2.5, axis=0), Phi @ np.percentile(mix, 97.5, axis=0)
    G = X                                                                                         # (for the single-chain comparison below)
    area = np.trapezoid(g_s, np.log(tau_plot)) if hasattr(np, 'trapezoid') else np.trapz(g_s, np.log(tau_plot))
    out = dict(stem=stem, sat=fit.n_max_treedepth, ours_vs_stored=rel_l2(ours, ref[:, 1]), stored_vs_map=rel_l2(ref[:, 1], gmap),
               ours_vs_map=rel_l2(ours, gmap), gs_min=float(g_s.min() / g_s.max()), gs_area=float(area), gs_peak=float(g_s.max()),
               lo_err=rel_l2(lo, ref[:, 2]), hi_err=rel_l2(hi, ref[:, 3]),
               lo_err_plain=rel_l2(Phi @ np.percentile(G, 2.5, axis=0), ref[:, 2]), hi_err_plain=rel_l2(Phi @ np.percentile(G, 97.5, axis=0), ref[:, 3]))
    if verbose:
        print('%-18s seed %d: saturated %3d | mean: ours vs stored %.4f, stored vs stored MAP %.4f, ours vs stored MAP %.4f | frozen-chain curve '
              '2*stored - ours: min/max %.3f, area %.3f, peak %.3f | stored 2.5 %% / 97.5 %% curves vs the 50/50 mixture %.4f / %.4f (vs our chain alone %.4f / %.4f)'
              % (stem, seed, out['sat'], out['ours_vs_stored'], out['stored_vs_map'], out['ours_vs_map'], out['gs_min'], out['gs_area'], out['gs_peak'],
                 out['lo_err'], out['hi_err'], out['lo_err_plain'], out['hi_err_plain']))
    return out


if __name__ == '__main__':
    print('== (2), (3): the stored curves against each other and the frozen-chain mixture')
    for stem in ('RC_Macdonald_0.25', 'RC_Orazem_0.25', 'RC_uniform_0.25', 'RC_noiseless'):
        for seed in (1234, 3234):
            mixture_check(stem, seed)
    print('== (1): HIP 4 x (1000 + 1000) against the CPU oracle\'s long run')
    O = load('rc_macdonald_oracle')
    i = stems.index(STEM)
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    inv = Inverter(basis_freq=f)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=1000, samples=1000, chains=4, random_seed=1234)
    fit = inv._sample_result
    x = inv._rescale_coef(fit[inv._get_stan_coef_name('DRT')], 'series')
    print('HIP: leapfrogs %d, saturated %d, divergent %d, step sizes %s' % (fit.n_leapfrog, fit.n_max_treedepth, fit.n_divergent, ' '.join('%.4f' % s for s in fit.stepsize)))
    print('oracle: leapfrogs %s, saturated %s, divergent %s, step sizes %s' % (O['n_leapfrog'], O['n_max_treedepth'], O['n_divergent'], O['stepsize']))
    print('coefficient posterior mean, HIP vs oracle: rel-L2 %.4f; 2.5 %% %.4f; 97.5 %% %.4f; R_inf %.6f vs %.6f' % (
        rel_l2(x.mean(axis=0), O['x_mean']), rel_l2(np.percentile(x, 2.5, axis=0), O['x_lo']), rel_l2(np.percentile(x, 97.5, axis=0), O['x_hi']),
        np.mean(inv._rescale_coef(fit['Rinf'], 'series')), float(O['Rinf_mean'])))
    g = inv.predict_distribution('DRT', eval_tau=tau_plot)
    print('gamma mean of the long HIP run vs the stored curve: %.4f (2 x (200 + 200): 0.175-0.179)' % rel_l2(g, S['Gout_bayes'][i][:, 1]))
