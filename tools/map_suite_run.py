"""The reference's published MAP study re-run through the drop-in API: the 60 simulated DRT spectra of code_EchemActa/Run fits.ipynb
cell 4 (tests/golden/hmc_suite.npz holds the spectra, the stored curves map_results/Gout_*.csv and the notebook's wall times).
Settings = the notebook's: basis = measurement frequencies (K = 81), nonneg except ZARC-RL, sigma_min 0.002 (0.005 noiseless),
random init seed 1234.  Per spectrum: gamma of the default fit (stationary point, two starts) and of algorithm='LBFGS', n_starts=1
(the reference's kind of iterate) against the stored curve; wall times ours | reference."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

S = load('hmc_suite')
tau_plot = np.logspace(-7, 2, 200)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
print('%-26s | %-28s | %-38s | %s' % ('spectrum', 'default fit: dL2 vs stored, s', "algorithm='LBFGS', n_starts=1: dL2, its, s", 'reference s'))
acc = []
for i, stem in enumerate(S['stems']):
    stem = str(stem)
    if args and not any(a in stem for a in args):
        continue
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    ref = S['Gout_map'][i][:, 1]
    kw = dict(nonneg=not stem.startswith('ZARC-RL'), mode='optimize', sigma_min=0.005 if 'noiseless' in stem else 0.002)
    inv = Inverter(basis_freq=f)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if not acc:
            inv.fit(f, Z, **kw)                      # first-use costs out of the timing
        t0 = time.time(); inv.fit(f, Z, **kw); t1 = time.time()
        g = inv.predict_distribution('DRT', eval_tau=tau_plot); r = inv._opt_report
        t2 = time.time(); inv.fit(f, Z, algorithm='LBFGS', n_starts=1, **kw); t3 = time.time()
        gs = inv.predict_distribution('DRT', eval_tau=tau_plot); rs = inv._opt_report
    d, ds = rel_l2(g, ref), rel_l2(gs, ref)
    acc.append((d, ds, t1 - t0, t3 - t2, S['diag'][i][4]))
    print('%-26s | %.4f  %.3f s (start %d, rc %d) | %.4f  %6d its  rc %2d  %.3f s         | %.2f' % (
        stem, d, t1 - t0, r['start'], r['return_code'], ds, rs['iterations'], rs['return_code'], t3 - t2, S['diag'][i][4]), flush=True)
a = np.array(acc)
print('median rel-L2 vs the stored MAP curves: default fit %.4f (max %.4f), Stan-style iterate %.4f (max %.4f); '
      'total wall time %.1f s | %.1f s | reference %.1f s' % (np.median(a[:, 0]), a[:, 0].max(), np.median(a[:, 1]), a[:, 1].max(),
                                                               a[:, 2].sum(), a[:, 3].sum(), a[:, 4].sum()))
