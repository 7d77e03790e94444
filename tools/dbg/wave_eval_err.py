"""Debug: where does the one-chain-per-wave evaluator differ from the tile evaluator?  usage: wave_eval_err.py K81 [sample|optimize] [pos]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_wave import _problem, _wave_logp_grad
tag = sys.argv[1] if len(sys.argv) > 1 else 'K81'
mode = sys.argv[2] if len(sys.argv) > 2 else 'sample'
pos = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
prob, om = _problem(tag, mode, pos)
K = (prob.D - 9) // 2
rng = np.random.default_rng(3)
theta = rng.uniform(-2, 2, (70, prob.D))
lp, g = _wave_logp_grad(prob, theta, mode == 'sample')
lp16, g16 = prob.logp_grad(theta, jacobian=(mode == 'sample'))
print(tag, mode, pos, 'D', prob.D, 'K', K, 'max lp err', np.max(np.abs(lp - lp16) / np.maximum(1, np.abs(lp16))))
names = ['Rinf', 'induc'] + ['x%d' % i for i in range(K)] + ['sres', 'ap', 'ar', 'ai'] + ['u%d' % i for i in range(K)] + ['d0', 'd1', 'd2']
for i in range(3):
    e = np.abs(g[i] - g16[i]) / np.maximum(1.0, np.max(np.abs(g16[i])))
    top = np.argsort(-e)[:8]
    print(i, ' '.join('%s:%.2e(%.3g vs %.3g)' % (names[j], e[j], g[i][j], g16[i][j]) for j in top))
