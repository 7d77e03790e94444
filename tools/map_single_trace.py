"""One spectrum, Inverter.fit(mode='optimize') repeated: wall time per fit against the sum of kernel time (run under rocprofv3 --kernel-trace)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load
from bayes_drt_amd.inversion import Inverter
c = load('csv_2ZARC_uniform_0.25')
f, Z = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 81
bf = f if K == 81 else np.logspace(10, -6, K)
inv = Inverter(basis_freq=bf)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    inv.fit(f, Z, nonneg=True, mode='optimize')
    n = 10
    t0 = time.time()
    for _ in range(n):
        inv.fit(f, Z, nonneg=True, mode='optimize')
    dt = (time.time() - t0) / n
print('K = %d: %.1f ms per fit (two starts in one batch); report %s' % (K, dt * 1e3, {k: inv._opt_report[k] for k in ('iterations', 'newton_iterations', 'return_code')}))
