"""Debug / parity sweep: the one-chain-per-wave evaluator against the 16-column tile evaluator over problem shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayes_drt_amd import matrices as gm
from bayes_drt_amd.model import Problem
from tests.test_gpu_wave import _wave_logp_grad

shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or [(81, 161), (81, 81), (81, 101), (41, 51), (53, 81), (96, 192), (81, 97), (81, 65), (33, 33), (64, 128), (65, 129), (20, 15)]
for nf, K in shapes:
    f = np.logspace(6, -2, nf)
    ppd = (nf - 1) / 8.0
    # basis on the same logarithmic spacing (or half of it) so that A is exactly Toeplitz
    ratio = max(1, int(round((K - 1) / (nf - 1))))
    lo = 6 + ((K - 1) / ratio - (nf - 1)) / 2 / ppd
    bf = np.logspace(lo, lo - (K - 1) / (ppd * ratio), K)
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    rng = np.random.default_rng(nf * 1000 + K)
    Z = rng.standard_normal(2 * nf)
    try:
        prob = Problem([dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], Z, f, ups_alpha=1.0, ups_beta=0.1)
    except Exception as e:
        print(nf, K, 'problem refused:', e); continue
    th = rng.uniform(-2, 2, (40, prob.D))
    try:
        lp, g = _wave_logp_grad(prob, th, True)
    except AssertionError as e:
        print(nf, K, 'not wave-capable:', e); prob.close(); continue
    lp16, g16 = prob.logp_grad(th, jacobian=True)
    el = np.max(np.abs(lp - lp16) / np.maximum(1, np.abs(lp16)))
    eg = np.max(np.abs(g - g16) / np.maximum(1.0, np.max(np.abs(g16), axis=1, keepdims=True)))
    j = np.unravel_index(np.argmax(np.abs(g - g16) / np.maximum(1.0, np.max(np.abs(g16), axis=1, keepdims=True))), g.shape)
    print('%3d x %3d  evaluator %d  lp err %.2e  grad err %.2e (point %d, element %d of %d)' % (nf, K, prob.evaluator() if hasattr(prob, 'evaluator') else -1, el, eg, j[0], j[1], prob.D), flush=True)
    prob.close()
