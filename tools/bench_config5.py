"""GPU box: throughput of the GENERIC block evaluator inside the sampler on BASELINE config 5 (DRT + transmissive planar DDT,
2 x 161 basis functions, outlier error model, D = 818) and, for comparison, the headline S1 family through the generic path
(BDRT_GENERIC_TILE=1).  Real NUTS rounds, n units resident, evaluations per second."""
import ctypes as C, os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
from bayes_drt_amd import stan_models
from bayes_drt_amd._lib import NutsControl, check, ptr

d = load('kat_DRT-2-TpDDT_uniform_0.25')
f, Z = d['data_freq'], d['data_Z'].copy()
for i in (10, 40, 70):
    Z[i] *= 1.5
bf = np.logspace(10, -6, 161)
if '--default-basis' in sys.argv:            # the package default: the basis on the measurement frequencies (K = 81 per distribution)
    bf = None
dists = {'DRT': {'kernel': 'DRT'},
         'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
if '--series-outliers' in sys.argv:          # single DRT with the outlier error model: D = 2*161 + 9 + 2*81 = 493
    dists = {'DRT': {'kernel': 'DRT'}}
inv = Inverter(basis_freq=bf, distributions=dists)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, outliers=True, mode='sample', warmup=5, samples=5, chains=1)
model = stan_models.load_pickle(inv.stan_model_name)
prob = model._prepare(inv._stan_input)
lib = prob._lib
n_units = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
ctrl = NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.adapt_delta, ctrl.adapt_t0 = 0.9, 10.0
h = lib.bdrt_sampler_create(prob.handle, n_units, None, None, 1000000, 1, C.c_uint64(7), None, C.byref(ctrl))
assert h, lib.bdrt_last_error()
def adv(r):
    while r > 0:
        k = min(50, r); check(lib.bdrt_sampler_advance(h, k, None), 'advance'); r -= k
adv(300); check(lib.bdrt_sampler_sync(h), 'sync')
prof = '--phase-profile' in sys.argv
if prof: check(lib.bdrt_sampler_phase_profile(h, 1, None), 'prof')
n0 = lib.bdrt_sampler_total_leapfrogs(h); t0 = time.perf_counter()
adv(600); check(lib.bdrt_sampler_sync(h), 'sync')
t1 = time.perf_counter(); n1 = lib.bdrt_sampler_total_leapfrogs(h)
print('config 5 (%s, D=%d, %d units): %.2f M evals/s, %.1f us per round' % (inv.stan_model_name, prob.D, n_units,
      (n1 - n0) / (t1 - t0) / 1e6, (t1 - t0) / 600 * 1e6))
if prof:
    cyc = (C.c_longlong * 32)(); check(lib.bdrt_sampler_phase_profile(h, 0, cyc), 'prof')
    if lib.bdrt_sampler_kind(h) == 2:                             # one-chain-per-workgroup kernel (bdrt_solo_wide.h)
        print('ONE-CHAIN KERNEL: evaluation %.0f cycles/round, stages after it %.0f' % (cyc[0] / n_units / 600, cyc[1] / n_units / 600))
        print('   evaluation phases (all blocks): constrain %.0f, products + prior %.0f, sums + L^T w %.0f, likelihood %.0f, operand %.0f, '
              'transposed products %.0f, chain rule %.0f, scalars + lp %.0f' % tuple(cyc[k] / n_units / 600 for k in range(2, 10)))
        print('   stages after it: first trip + reduction %.0f, scalar logic %.0f, checkpoint / subtree close %.0f, transition end + continue %.0f, '
              'sample / new start %.0f' % tuple(cyc[k] / n_units / 600 for k in (25, 27, 28, 29, 30)))
        lib.bdrt_sampler_destroy(h); sys.exit(0)
    nwg = (n_units + 15) // 16
    wr = nwg * 600 * 8                                            # wave-rounds
    # wide-vector kernels (D > 512): per-wave averages of the round's stages (bdrt_nuts.hip, bdrt_nuts_wide.h)
    for nm, k in (('evaluation (MFMA tile)', 17), ("chain's own pass (phase P)", 18), ('cooperative phase (phase H)', 20), ('round barrier wait', 23)):
        print('WAVE-AVG %-28s %8.0f cycles/round' % (nm, cyc[k] / wr))
    if cyc[4]:                                                   # sub-phases of the half-wave evaluator (bdrt_tile_hw.h, PROFT), both blocks summed
        print('EVALUATOR per wave-round: ' + ', '.join('%s %.0f' % (nm, cyc[k] / wr) for nm, k in (
            ('P1', 4), ('B1 wait', 5), ('A x', 6), ('prior chain', 7), ('B2 wait', 8), ('accumulate', 25), ('likelihood', 26),
            ('operand', 27), ('B3 wait', 28), ('A^T g', 29), ('B4 wait', 30), ('epilogue', 31))))
    for k, nm in enumerate(['finished by its own pass', 'with a chain finished cooperatively']):
        if cyc[2 * k + 1]: print('WAVE-CLASS %-36s %5.1f %% of wave-rounds, %8.0f cycles after the evaluation' % (
            nm, 100.0 * cyc[2 * k + 1] / wr, cyc[2 * k] / cyc[2 * k + 1]))
    if cyc[15]:
        print('COOP %.2f chains per workgroup-round; cycles per chain (thread 0): first trip + reduction %.0f, scalar logic %.0f, '
              'checkpoint / subtree close %.0f, transition end + continue %.0f, sample / new start %.0f' % (
              cyc[15] / nwg / 600, *[cyc[k] / cyc[15] for k in (9, 11, 12, 13, 14)]))
lib.bdrt_sampler_destroy(h)
