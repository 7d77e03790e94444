import sys, warnings, numpy as np
sys.path.insert(0, '/root/repo')
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter
TAU_PLOT = np.logspace(-7, 2, 200)
c, c4 = load('csv_2RC_uniform_0.25'), load('csv_2RC_uniform_0.25_4x1000')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
ref = c4['Gout_bayes']
gs = []
for seed in [None, 1, 2, 3, 4, 5, 6, 7]:
    inv = Inverter(basis_freq=f)
    kw = {} if seed is None else dict(random_seed=seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=500, samples=500, chains=4, sigma_min=0.002, **kw)
    g = inv.predict_distribution('DRT', eval_tau=TAU_PLOT)
    lo = inv.predict_distribution('DRT', eval_tau=TAU_PLOT, percentile=2.5)
    hi = inv.predict_distribution('DRT', eval_tau=TAU_PLOT, percentile=97.5)
    r = inv._sample_result
    gs.append(g)
    print('seed %s: mean %.4f lo %.4f hi %.4f; saturated %d divergent %d' % (seed, rel_l2(g, ref[:, 1]), rel_l2(lo, ref[:, 2]), rel_l2(hi, ref[:, 3]), r.n_max_treedepth, r.n_divergent), flush=True)
gm = np.mean(gs, axis=0)
print('mean of runs vs reference: %.4f; run-to-run (each vs mean of ours): %s' % (rel_l2(gm, ref[:, 1]), ' '.join('%.4f' % rel_l2(g, gm) for g in gs)))
