import os, sys, time, warnings, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
inv = Inverter(basis_freq=f)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, mode='optimize')
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): inv.fit(f, Z, nonneg=True, mode='optimize')
    pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
