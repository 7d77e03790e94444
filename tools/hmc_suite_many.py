"""The reference's published HMC study (tests/golden/hmc_suite.npz: the 60 simulated DRT spectra of code_EchemActa/Run fits.ipynb
cell 5) through Inverter.fit_many: the spectra that share their options -- sign constraint, sigma_min -- are ONE batch (one shared
problem, 2 chains per spectrum as units of one device-resident sampler), i.e. the notebook's loop of 60 `fit` calls in three calls.
Same settings as tools/hmc_suite_run.py (which runs the 60 calls one after the other); per spectrum the same columns.
Usage: hmc_suite_many.py [--seed s]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

seed = int(sys.argv[sys.argv.index('--seed') + 1]) if '--seed' in sys.argv else 1234
S = load('hmc_suite')
tau_plot = np.logspace(-7, 2, 200)
stems = [str(s) for s in S['stems']]
f = S['Z'][0][:, 0]
assert all(np.array_equal(S['Z'][i][:, 0], f) for i in range(len(stems)))
groups = {}
for i, stem in enumerate(stems):
    groups.setdefault((not stem.startswith('ZARC-RL'), 0.005 if 'noiseless' in stem else 0.002), []).append(i)
rows = {}
t_all = time.time()
for (nonneg, smin), idx in groups.items():
    Zs = [S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2] for i in idx]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.time()
        views = Inverter(basis_freq=f).fit_many(f, Zs, nonneg=nonneg, mode='sample', warmup=200, samples=200, chains=2, sigma_min=smin,
                                                random_seed=seed)
        dt = time.time() - t0
    print('# batch nonneg=%s sigma_min=%g: %d spectra x 2 chains x (200 + 200) in %.2f s' % (nonneg, smin, len(idx), dt), flush=True)
    for i, v in zip(idx, views):
        rows[i] = v
wall = time.time() - t_all
tot = np.zeros(4)
print('%-26s | %-22s | %-15s | %s' % ('spectrum', 'saturated ours | ref', 'divergent o | r', 'gamma mean / lo / hi rel-L2 vs stored; leapfrogs; stepsizes'))
for i, stem in enumerate(stems):
    v = rows[i]; fit = v._sample_result; ref, d = S['Gout_bayes'][i], S['diag'][i]
    g = [v.predict_distribution('DRT', eval_tau=tau_plot)] + [v.predict_distribution('DRT', eval_tau=tau_plot, percentile=p) for p in (2.5, 97.5)]
    print('%-26s | %4d of 400 | %4d     | %3d | %3d       | %.4f %.4f %.4f ; %8d ; %s' % (
        stem, fit.n_max_treedepth, d[0], fit.n_divergent, d[1], rel_l2(g[0], ref[:, 1]), rel_l2(g[1], ref[:, 2]), rel_l2(g[2], ref[:, 3]),
        fit.n_leapfrog, ' '.join('%.4f' % s for s in fit.stepsize)))
    tot += [fit.n_max_treedepth, d[0], fit.n_divergent, d[1]]
print('totals: saturated ours %d | reference %d of %d; divergent ours %d | reference %d' % (tot[0], tot[1], 400 * len(stems), tot[2], tot[3]))
print('wall time of the %d fit_many calls (60 spectra): %.2f s (the notebook: 60 fit calls, 32-180 s each)' % (len(groups), wall))
