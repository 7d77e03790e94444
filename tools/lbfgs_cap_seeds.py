"""The five spectra of the published MAP study on which the Stan-style L-BFGS (algorithm='LBFGS', n_starts=1: Stan 2.19's iteration and
termination tests restated, bdrt_lbfgs.h) ran into the notebook's 50 000-iteration cap with seed 1234 while Stan itself -- going by the
notebook's wall times, ~0.6 ms per iteration -- stopped after 2 000 - 5 500 iterations: eight seeds each.  pystan's start point for its
seed is not reproducible here (another generator), so the question a seed sweep answers is whether the cap is a property of the
restatement or of the start point: iterations, return code (0: a tolerance test fired, 1: cap), lp, and the curve against the polished
optimum of the default fit."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

S = load('hmc_suite')
stems = [str(s) for s in S['stems']]
tau_plot = np.logspace(-7, 2, 200)
CAP = ['ZARC_Macdonald_0.25', '2ZARC_uniform_0.25', 'Gerischer_Orazem_0.25', 'ZARC_uniform_0.25', 'ZARC_Orazem_1.0']
seeds = (1234, 1, 2, 3, 4, 5, 6, 7)
for stem in CAP:
    i = stems.index(stem)
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    kw = dict(nonneg=True, mode='optimize', sigma_min=0.002)
    inv = Inverter(basis_freq=f)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, **kw)
        g_opt, lp_opt = inv.predict_distribution('DRT', eval_tau=tau_plot), inv._opt_report['lp']
        rows = []
        for sd in seeds:
            t0 = time.time(); inv.fit(f, Z, algorithm='LBFGS', n_starts=1, random_seed=sd, **kw); dt = time.time() - t0
            r = inv._opt_report
            rows.append((sd, r['iterations'], r['return_code'], r['lp'], rel_l2(inv.predict_distribution('DRT', eval_tau=tau_plot), g_opt), dt))
    its = [r[1] for r in rows]
    print('%-24s stationary point lp %.4f; reference wall time %.2f s (~%d Stan iterations at 0.6 ms); ours over %d seeds: %d at the cap, the others %s iterations'
          % (stem, lp_opt, S['diag'][i][4], int(S['diag'][i][4] / 0.6e-3), len(seeds), sum(1 for r in rows if r[2] == 1), sorted(r[1] for r in rows if r[2] != 1)))
    for r in rows:
        print('    seed %4d: %6d iterations, rc %d, lp %.4f (optimum %+.4f), curve vs the stationary point %.4f, %.2f s' % (r[0], r[1], r[2], r[3], r[3] - lp_opt, r[4], r[5]))
