"""GPU box: MAP of BASELINE config 4's 512 spectra as ONE batch (bdrt_optimize: every Newton round serves all fits --
2 D probe gradients per fit in one evaluator launch, one Cholesky workgroup per fit), K = 81 and K = 161."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import optimize_batch

n = 512
kw = bench.build_problem_kwargs(n)
blocks, Z, freq = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
prob = Problem(blocks, Z, freq, **kw)
th0 = np.random.RandomState(1).uniform(-2, 2, (n, prob.D))
optimize_batch(prob, th0[:8], spec=np.arange(8))
t0 = time.time(); x, rep = optimize_batch(prob, th0, spec=np.arange(n)); t1 = time.time()
its = [r['newton_iterations'] for r in rep]
print('%d spectra, D = %d: %.2f s (%.1f ms per fit), Newton steps %d..%d (median %d), converged %d of %d, max |grad|_inf %.1e' % (
    n, prob.D, t1 - t0, (t1 - t0) / n * 1e3, min(its), max(its), int(np.median(its)), sum(r['return_code'] == 0 for r in rep), n,
    max(r['grad_inf'] for r in rep)))
