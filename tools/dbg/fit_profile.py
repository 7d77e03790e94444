"""Where the wall time of one Inverter.fit(mode='optimize') goes on the host (cProfile, after a warm-up fit)."""
import cProfile, pstats, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
d = load('kat_2ZARC_uniform_0.25') if False else None
import bench
kw = bench.build_problem_kwargs(1)
f = kw['freq']; Z = kw['Z'][0]; Z = Z[:len(f)] + 1j * Z[len(f):]
for K in (81, 161):
    bf = np.logspace(7, -3, K) if K == 101 else (np.logspace(np.log10(f[0]) + 0.0, np.log10(f[-1]), K) if K == 81 else np.logspace(np.log10(f[0]) + 4, np.log10(f[-1]) - 4, K))
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, mode='optimize')
        t0 = time.perf_counter(); inv.fit(f, Z, mode='optimize'); t1 = time.perf_counter()
        print('K=%d fit: %.3f s' % (K, t1 - t0))
        pr = cProfile.Profile(); pr.enable(); inv.fit(f, Z, mode='optimize'); pr.disable()
    st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(18)
