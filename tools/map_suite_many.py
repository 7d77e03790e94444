"""The reference's published MAP study (tests/golden/hmc_suite.npz: the 60 simulated DRT spectra of code_EchemActa/Run fits.ipynb cell 4)
through Inverter.fit_many(mode='optimize'): the spectra that share their options -- sign constraint, sigma_min -- are ONE lock-step batch
of (spectrum, start) fits, i.e. the notebook's loop of 60 `fit` calls in a few calls.  Same settings as tools/map_suite_run.py (which runs
the 60 calls one after the other); per spectrum gamma against the stored curve and against the one-call-per-spectrum result."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

S = load('hmc_suite')
tau_plot = np.logspace(-7, 2, 200)
stems = [str(s) for s in S['stems']]
f = S['Z'][0][:, 0]
groups = {}
for i, stem in enumerate(stems):
    groups.setdefault((not stem.startswith('ZARC-RL'), 0.005 if 'noiseless' in stem else 0.002), []).append(i)
check_single = '--no-single' not in sys.argv
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    Inverter(basis_freq=f).fit_many(f, [S['Z'][0][:, 1] + 1j * S['Z'][0][:, 2]], nonneg=True, mode='optimize', sigma_min=0.002)   # first-use costs
rows, t_all = {}, time.time()
for (nonneg, smin), idx in groups.items():
    Zs = [S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2] for i in idx]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.time()
        views = Inverter(basis_freq=f).fit_many(f, Zs, nonneg=nonneg, mode='optimize', sigma_min=smin)
        dt = time.time() - t0
    print('# batch nonneg=%s sigma_min=%g: %d spectra in %.3f s' % (nonneg, smin, len(idx), dt), flush=True)
    for i, v in zip(idx, views):
        rows[i] = v
wall = time.time() - t_all
acc, t_single = [], 0.0
for i, stem in enumerate(stems):
    v = rows[i]
    g = v.predict_distribution('DRT', eval_tau=tau_plot)
    d = rel_l2(g, S['Gout_map'][i][:, 1])
    ds = float('nan')
    if check_single:
        inv = Inverter(basis_freq=f)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            t0 = time.time()
            inv.fit(f, S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2], nonneg=not stem.startswith('ZARC-RL'), mode='optimize',
                    sigma_min=0.005 if 'noiseless' in stem else 0.002)
            t_single += time.time() - t0
        ds = rel_l2(g, inv.predict_distribution('DRT', eval_tau=tau_plot))
    acc.append((d, ds, S['diag'][i][4]))
    print('%-26s | vs stored %.4f | vs its own fit() call %.2e | start %d rc %d' % (stem, d, ds, v._opt_report['start'], v._opt_report['return_code']))
a = np.array(acc)
print('median rel-L2 vs the stored MAP curves %.4f (max %.4f); largest difference to the one-call-per-spectrum result %.2e' % (
    np.median(a[:, 0]), a[:, 0].max(), np.nanmax(a[:, 1]) if check_single else float('nan')))
print('wall time: %d fit_many calls (60 spectra) %.2f s | 60 fit calls %.2f s | the notebook %.1f s' % (len(groups), wall, t_single, a[:, 2].sum()))
