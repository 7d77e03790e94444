"""GPU box: wall time of Inverter.ridge_fit (hyper-lambda ridge, one bdrt_ridge launch) and Inverter.ridge_ReImCV (61 lambda_0 x
{real, imag} hyper-ridge fits in one launch) on the reference's 2-ZARC spectrum, K = 81 and K = 161; first-use costs excluded."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter

c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
for K, bf in ((81, f), (161, np.logspace(10, -6, 161))):
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.ridge_fit(f, Z)
        t0 = time.time(); inv.ridge_fit(f, Z); t1 = time.time()
        n_it = len(getattr(inv, '_iter_history', []) or [])
        print('K=%d: ridge_fit (hyper_lambda) %.4f s' % (K, t1 - t0) + (', %d outer iterations' % n_it if n_it else ''))
        lam = np.logspace(-9, -3, 61)
        inv.ridge_ReImCV(f, Z, lambdas=lam)
        t0 = time.time(); best = inv.ridge_ReImCV(f, Z, lambdas=lam); t1 = time.time()
        print('K=%d: ridge_ReImCV over %d lambda_0 (122 hyper-ridge fits, one launch) %.4f s, optimum %s' % (K, lam.size, t1 - t0, str(best)[:60]))
