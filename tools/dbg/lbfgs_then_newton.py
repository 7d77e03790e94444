"""MAP wall time against the number of Stan-style L-BFGS iterations in front of the Newton iteration (lbfgs_before_newton)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import load
from bayes_drt_amd import stan_models
from bayes_drt_amd.inversion import Inverter
c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
for K, bf in ((81, f), (101, None), (161, np.logspace(10, -6, 161))):
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, mode='optimize', n_starts=1)
    model = stan_models.load_pickle(inv.stan_model_name)
    ref = None
    for nb in (0, 100, 300, 1000, 3000):
        model.optimizing(inv._stan_input, seed=1234, lbfgs_before_newton=nb)
        t0 = time.perf_counter(); res = model.optimizing(inv._stan_input, seed=1234, lbfgs_before_newton=nb); t1 = time.perf_counter()
        r = model.last_report
        x = res['x']
        if ref is None: ref = x
        print('K=%d lbfgs_before_newton=%5d: %.3f s, L-BFGS it %d, Newton it %d, evals %d, lp %.6f, rc %d, |g| %.1e, x vs Newton-only rel %.1e' % (
            K, nb, t1 - t0, r['iterations'], r['newton_iterations'], r['n_evals'], r['lp'], r['return_code'], r['grad_inf'],
            np.linalg.norm(x - ref) / np.linalg.norm(ref)), flush=True)
