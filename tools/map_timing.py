"""MAP timing on the GPU box: Inverter.fit(mode='optimize') on the reference's 2-ZARC spectrum at K = 81 / 101 / 161
(device-resident Newton iteration, two starts in one batch), and the Stan-style iterate (algorithm='LBFGS', no polish) beside it, both
against the reference's committed MAP curve (map_results/Gout_2ZARC_uniform_0.25.csv)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
tau_plot = np.logspace(-7, 2, 200)
ref, true = c['Gout_map'][:, 1], c['gamma_true'][:, 1]
for K, bf in ((81, f), (101, None), (161, np.logspace(10, -6, 161))):
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True, mode='optimize')                       # first-use costs out of the timing
        t0 = time.time(); inv.fit(f, Z, nonneg=True, mode='optimize'); t1 = time.time()
    r = inv._opt_report
    g = inv.predict_distribution('DRT', eval_tau=tau_plot)
    print('K=%d: fit(mode=optimize) %.3f s  (L-BFGS %d it + Newton %d it, %d evals), lp %.4f, |grad|_inf %.1e, rc %d; '
          'gamma vs reference MAP rel-L2 %.4f, vs true %.4f' % (K, t1 - t0, r['iterations'], r['newton_iterations'], r['n_evals'],
                                                                 r['lp'], r['grad_inf'], r['return_code'], rel_l2(g, ref), rel_l2(g, true)))
    print('      starts (random, ridge) in one batch: kept start %d; lp of the starts %s' % (
        r['start'], ', '.join('%.4f (rc %d, %d it)' % (x['lp'], x['return_code'], x['newton_iterations']) for x in r['starts'])))
    os.environ['BDRT_MAP_SINGLE_START'] = '1'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.time(); inv.fit(f, Z, nonneg=True, mode='optimize'); t1 = time.time()
    del os.environ['BDRT_MAP_SINGLE_START']
    print('      the random start alone (BDRT_MAP_SINGLE_START=1): %.3f s' % (t1 - t0))
    # the Stan-style iterate beside it: L-BFGS(5) with Stan's termination tests, iter = 50000, no second-order polish
    from bayes_drt_amd.engine import StanModel
    m = StanModel(inv.stan_model_name)
    t0 = time.time(); res = m.optimizing(inv._stan_input, iter=50000, seed=1234, algorithm='LBFGS'); t1 = time.time()
    tau = inv.distributions['DRT']['tau']; eps = inv.distributions['DRT']['epsilon']
    gs = np.exp(-(eps * np.log(tau_plot[:, None] / tau[None, :])) ** 2) @ (res['x'] * inv._Z_scale)
    r = m.last_report
    print('      Stan-style L-BFGS only: %.3f s, %d iterations, lp %.4f, |grad|_inf %.1e; gamma vs reference MAP rel-L2 %.4f, '
          'vs the polished MAP %.4f' % (t1 - t0, r['iterations'], r['lp'], r['grad_inf'], rel_l2(gs, ref), rel_l2(gs, g)))
