"""BASELINE config 5 on the GPU box, end to end through the Inverter mirror: DRT + transmissive planar DDT (2 x 161 basis
functions), outlier error model, the reference's simulated spectrum Z_DRT-2-TpDDT_uniform_0.25 with three injected outliers
(x1.5 modulus at indices 10, 40, 70): MAP, then HMC with 4 chains; wall times, diagnostics, outlier detection."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter

d = load('kat_DRT-2-TpDDT_uniform_0.25')
f, Z = d['data_freq'], d['data_Z'].copy()
bad = (10, 40, 70)
for i in bad:
    Z[i] *= 1.5
bf = np.logspace(10, -6, 161)
dists = {'DRT': {'kernel': 'DRT'},
         'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
inv = Inverter(basis_freq=bf, distributions=dists)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, outliers=True, mode='optimize', max_iter=100)          # first-use costs out of the timings
    t0 = time.time(); inv.fit(f, Z, nonneg=True, outliers=True, mode='optimize'); t1 = time.time()
rep = inv._opt_report
res = np.abs(inv.predict_Z(f) - Z) / np.abs(Z)
so = inv.error_fit['sigma_out']; so_n = np.hypot(so[:81], so[81:])
print('MAP (%s, D=%d): %.1f s, return code %d, |grad|_inf %.2e; relative residual at the injected outliers %s vs median %.4f; '
      'largest sigma_out at indices %s' % (inv.stan_model_name, len(inv._opt_result['theta_unconstrained']), t1 - t0,
                                           rep['return_code'], rep['grad_inf'], np.round(res[list(bad)], 3), np.median(res),
                                           sorted(np.argsort(so_n)[-3:].tolist())))
chains, warm, draws = 4, 400, 300
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    t0 = time.time(); inv.fit(f, Z, nonneg=True, outliers=True, mode='sample', warmup=warm, samples=draws, chains=chains); t1 = time.time()
fit = inv._sample_result
rh = {}
for name in ('xs', 'xp'):
    x = fit.chain_draws(name)
    big = x.mean(axis=(0, 1)) > 0.05 * x.mean(axis=(0, 1)).max()
    h = draws // 2
    hv = np.concatenate([x[:, :h], x[:, h:2 * h]], axis=0)[:, :, big]
    W = hv.var(axis=1, ddof=1).mean(axis=0); Bv = hv.mean(axis=1).var(axis=0, ddof=1) * h
    rh[name] = np.sqrt(((h - 1) / h * W + Bv / h) / W)
print('HMC %d chains x (%d + %d): %.1f s wall, %d leapfrogs (%.0f evals/s on one workgroup), divergent %d, tree-depth hits %d, '
      'split R-hat median xs %.3f xp %.3f (max %.3f)' % (chains, warm, draws, t1 - t0, fit.n_leapfrog, fit.n_leapfrog / (t1 - t0),
                                                        fit.n_divergent, fit.n_max_treedepth, np.median(rh['xs']),
                                                        np.median(rh['xp']), max(rh['xs'].max(), rh['xp'].max())))
