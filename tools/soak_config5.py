"""GPU box: complete HMC runs of BASELINE config 5's model (DRT + transmissive planar DDT, outlier error model, D = 818)
through the wide-vector sampler path at two packings (64 units: one chain per workgroup; 1024 units: four per workgroup):
every chain finishes, draws are finite; prints wall time, leapfrogs, divergences, tree-depth hits."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.getcwd())
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
from bayes_drt_amd import stan_models
from bayes_drt_amd.engine import sample_units
from bayes_drt_amd._lib import NutsControl
import ctypes as C
d = load('kat_DRT-2-TpDDT_uniform_0.25')
f, Z = d['data_freq'], d['data_Z'].copy()
bf = np.logspace(10, -6, 161)
dists = {'DRT': {'kernel': 'DRT'}, 'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
inv = Inverter(basis_freq=bf, distributions=dists)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, outliers=True, mode='sample', warmup=5, samples=5, chains=1)
prob = stan_models.load_pickle(inv.stan_model_name)._prepare(inv._stan_input)
ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.adapt_delta = 0.9
for n in (64, 1024):
    t0 = time.time()
    draws, lp, diag = sample_units(prob, n, 60, 40, 5, ctrl)
    t1 = time.time()
    nl = sum(x['n_leapfrog'] for x in diag)
    print('%d units x (60+40), D=%d: %.1f s, %d leapfrogs (%.2f M evals/s), finite %s, divergent %d, depth hits %d' % (
        n, prob.D, t1 - t0, nl, nl / (t1 - t0) / 1e6, bool(np.isfinite(draws).all()), sum(x['n_divergent'] for x in diag), sum(x['n_max_treedepth'] for x in diag)))
