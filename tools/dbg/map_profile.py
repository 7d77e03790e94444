"""cProfile of Inverter.fit(mode='optimize') at K = 161 (second call: first-use costs out): where the host time of a MAP fit goes."""
import os, sys, cProfile, pstats, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
inv = Inverter(basis_freq=np.logspace(10, -6, 161))
warnings.simplefilter('ignore')
inv.fit(f, Z, nonneg=True, mode='optimize')
ts = []
for _ in range(5):
    t0 = time.time(); inv.fit(f, Z, nonneg=True, mode='optimize'); ts.append(time.time() - t0)
print('fit times', ['%.1f ms' % (1e3 * t) for t in ts])
pr = cProfile.Profile(); pr.enable(); inv.fit(f, Z, nonneg=True, mode='optimize'); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
