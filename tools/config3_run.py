"""BASELINE config 3 on the GPU box: Inverter.fit(mode='sample'), 4 chains x (1000 warm-up + 1000 draws), 81 frequencies x
161 basis functions (basis_freq = logspace(10, -6, 161)), the reference's simulated 2-ZARC spectrum (committed fixture).
Prints wall time, leapfrogs, divergences, split R-hat and the recovered gamma(ln tau) against the reference's
committed HMC result (code_EchemActa/bayes_results/Gout_2ZARC_uniform_0.25.csv, K = 81) and the true distribution."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
tau_plot = np.logspace(-7, 2, 200)
chains, warm, draws = 4, 1000, 1000
for K, bf in ((161, np.logspace(10, -6, 161)), (81, f)):
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=10, samples=10, chains=chains)       # first-use costs out of the timing
        t0 = time.time()
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=warm, samples=draws, chains=chains)
        t1 = time.time()
    fit = inv._sample_result
    g = inv.predict_distribution('DRT', eval_tau=tau_plot)
    lo = inv.predict_distribution('DRT', eval_tau=tau_plot, percentile=2.5)
    hi = inv.predict_distribution('DRT', eval_tau=tau_plot, percentile=97.5)
    x = fit.chain_draws('x')
    big = x.mean(axis=(0, 1)) > 0.01 * x.mean(axis=(0, 1)).max()
    h = draws // 2
    halves = np.concatenate([x[:, :h], x[:, h:]], axis=0)[:, :, big]
    W = halves.var(axis=1, ddof=1).mean(axis=0); Bv = halves.mean(axis=1).var(axis=0, ddof=1) * h
    rhat = np.sqrt(((h - 1) / h * W + Bv / h) / W)
    ref = c['Gout_bayes']
    print('K=%d: %d chains x (%d + %d): %.2f s wall, %d leapfrogs (%.0f evals/s), divergent %d, treedepth hits %d, '
          'median split R-hat %.3f (max %.3f)' % (K, chains, warm, draws, t1 - t0, fit.n_leapfrog,
                                                  fit.n_leapfrog / (t1 - t0), fit.n_divergent, fit.n_max_treedepth,
                                                  np.median(rhat), rhat.max()))
    print('     gamma mean vs reference HMC result (K=81): rel-L2 %.4f; lo %.4f; hi %.4f; vs true gamma %.4f (reference: %.4f)'
          % (rel_l2(g, ref[:, 1]), rel_l2(lo, ref[:, 2]), rel_l2(hi, ref[:, 3]), rel_l2(g, c['gamma_true'][:, 1]),
             rel_l2(ref[:, 1], c['gamma_true'][:, 1])))
