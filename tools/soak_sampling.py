"""GPU box: a complete multi-spectrum HMC run (BASELINE config 4 shard, scaled): 256 spectra x 8 chains, 150 warm-up + 150
draws each, through sample_units; checks that every chain finishes, draws are finite, and reports wall time, leapfrogs,
divergences, tree-depth hits, step sizes and a cross-chain R-hat of the largest coefficients per spectrum."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_spectra, K
from bayes_drt_amd import matrices as gm
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import sample_units

ns, nc, warm, nd = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 8, 150, 150)
if len(sys.argv) >= 6:
    K = int(sys.argv[5])                                   # another basis length on the same ten-per-decade grid (101: the package default)
f, Z = synth_spectra(ns)
ext = (K - len(f)) // 2 / 10.0
bf = np.logspace(10, -6, K) if K == 161 else np.logspace(np.log10(f[0]) + ext, np.log10(f[0]) + ext - (K - 1) / 10.0, K)
tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
prob = Problem([dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], Z, f, ups_alpha=1.0, ups_beta=0.1)
spec = np.repeat(np.arange(ns, dtype=np.int32), nc)
cid = np.tile(np.arange(nc, dtype=np.int32), ns)
from bayes_drt_amd.engine import Sampler
t0 = time.time()
with Sampler(prob, ns * nc, warm, nd, 2026, None, spec=spec, chain_ids=cid) as smp:
    smp.run()
    draws, lp, diag = smp.results()
    print('layout: %d re-packings of the 16-chain workgroups (compaction), %d chains handed to the one-chain kernel for the tail'
          % (smp.compactions(), smp.tail_units()))
t1 = time.time()
nl = sum(d['n_leapfrog'] for d in diag)
per = np.sort(np.array([d['n_leapfrog'] for d in diag], dtype=float))
print('%d spectra x %d chains x (%d + %d): %.1f s wall, %.2f M leapfrogs, %.1f M evals/s end to end' %
      (ns, nc, warm, nd, t1 - t0, nl / 1e6, nl / (t1 - t0) / 1e6))
print('leapfrogs per chain: min %.3g, median %.3g, 95th percentile %.3g, max %.3g' % (per[0], per[len(per) // 2], per[int(0.95 * len(per))], per[-1]))
print('finite draws: %s; divergent transitions %d of %d; tree-depth hits %d; step size median %.3g (min %.3g, max %.3g)' %
      (bool(np.all(np.isfinite(draws)) and np.all(np.isfinite(lp))), sum(d['n_divergent'] for d in diag), ns * nc * nd,
       sum(d['n_max_treedepth'] for d in diag), np.median([d['stepsize'] for d in diag]),
       min(d['stepsize'] for d in diag), max(d['stepsize'] for d in diag)))
x = np.exp(draws[:, :, 2:2 + K]).reshape(ns, nc, nd, K)
rh = []
for s in range(ns):
    m = x[s].mean(axis=(0, 1)); big = m > 0.05 * m.max()
    xs = x[s][:, :, big]
    W = xs.var(axis=1, ddof=1).mean(axis=0); B = xs.mean(axis=1).var(axis=0, ddof=1) * nd
    rh.append(np.sqrt(((nd - 1) / nd * W + B / nd) / W).max())
rh = np.array(rh)
print('R-hat (8 chains, largest coefficients) per spectrum: median %.3f, 95th percentile %.3f, max %.3f' %
      (np.median(rh), np.percentile(rh, 95), rh.max()))
