"""The reference's published HMC study re-run through the drop-in API (tests/golden/hmc_suite.npz: the 60 simulated DRT spectra
of code_EchemActa/Run fits.ipynb cell 5 with the curves and pystan diagnostics the notebook stored / printed).

Settings = the notebook's: basis = measurement frequencies (K = 81), nonneg except ZARC-RL, sigma_min 0.002 (0.005 noiseless),
2 chains x (200 + 200), seed 1234, random init; and cell 6: 2RC_uniform_0.25 with 4 chains x (500 + 500).
Per spectrum: iterations that saturated tree depth 10 / divergent iterations (ours | reference), gamma mean / 2.5 % / 97.5 %
rel-L2 against the stored curves.  Usage: hmc_suite_run.py [stem-substring ...] [--seeds n]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

args = [a for a in sys.argv[1:] if not a.startswith('--')]
nseeds = int(sys.argv[sys.argv.index('--seeds') + 1]) if '--seeds' in sys.argv else 1
S = load('hmc_suite')
tau_plot = np.logspace(-7, 2, 200)


def run(f, Z, stem, chains, warm, draws, seed):
    inv = Inverter(basis_freq=f)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.time()
        inv.fit(f, Z, nonneg=not stem.startswith('ZARC-RL'), mode='sample', warmup=warm, samples=draws, chains=chains,
                sigma_min=0.005 if 'noiseless' in stem else 0.002, random_seed=seed)
        dt = time.time() - t0
    fit = inv._sample_result
    g = [inv.predict_distribution('DRT', eval_tau=tau_plot)] + \
        [inv.predict_distribution('DRT', eval_tau=tau_plot, percentile=p) for p in (2.5, 97.5)]
    return fit, g, dt


tot = np.zeros(4)
print('%-26s %5s | %-22s | %-15s | %s' % ('spectrum', 'secs', 'saturated ours | ref', 'divergent o | r', 'gamma mean / lo / hi rel-L2 vs stored; leapfrogs; stepsizes'))
for i, stem in enumerate(S['stems']):
    stem = str(stem)
    if args and not any(a in stem for a in args):
        continue
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    ref, d = S['Gout_bayes'][i], S['diag'][i]
    for sd in range(nseeds):
        fit, g, dt = run(f, Z, stem, 2, 200, 200, 1234 + 1000 * sd)
        print('%-26s %5.1f | %4d of 400 | %4d     | %3d | %3d       | %.4f %.4f %.4f ; %8d ; %s' % (
            stem, dt, fit.n_max_treedepth, d[0], fit.n_divergent, d[1], rel_l2(g[0], ref[:, 1]), rel_l2(g[1], ref[:, 2]),
            rel_l2(g[2], ref[:, 3]), fit.n_leapfrog, ' '.join('%.4f' % s for s in fit.stepsize)), flush=True)
    tot += [fit.n_max_treedepth, d[0], fit.n_divergent, d[1]]
print('totals: saturated ours %d | reference %d of %d; divergent ours %d | reference %d' % (tot[0], tot[1], 400 * len(S['stems']), tot[2], tot[3]))

if not args or any('4x1000' in a for a in args):
    c = load('csv_2RC_uniform_0.25'); c4 = load('csv_2RC_uniform_0.25_4x1000'); r4 = S['run4x1000']
    f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
    for sd in range(nseeds):
        fit, g, dt = run(f, Z, '2RC_uniform_0.25', 4, 500, 500, 1234 + 1000 * sd)
        ref = c4['Gout_bayes']
        print('2RC_uniform_0.25 4x(500+500): %.1f s (reference %.0f s); saturated %d of 2000 (reference %d); divergent %d (reference %d); '
              'gamma mean %.4f lo %.4f hi %.4f; leapfrogs %d; stepsizes %s' % (
                  dt, r4[3], fit.n_max_treedepth, r4[0], fit.n_divergent, r4[1], rel_l2(g[0], ref[:, 1]), rel_l2(g[1], ref[:, 2]),
                  rel_l2(g[2], ref[:, 3]), fit.n_leapfrog, ' '.join('%.5f' % s for s in fit.stepsize)))
