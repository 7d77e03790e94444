import os, sys, time, warnings, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
inv = Inverter(basis_freq=np.logspace(10, -6, 161))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, mode='sample', warmup=10, samples=10, chains=4)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.time()
    inv.fit(f, Z, nonneg=True, mode='sample', warmup=1000, samples=1000, chains=4)
    dt = time.time() - t0
    pr.disable()
fit = inv._sample_result
print('wall %.2f s, %d leapfrogs, longest chain x 5.74 us = ?' % (dt, fit.n_leapfrog))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
