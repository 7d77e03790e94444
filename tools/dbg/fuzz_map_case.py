"""One case of tests/fuzz_parity.py: the device MAP iterate against the oracle at the answer (x_sum_raw, lp, gradient)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import fuzz_parity as fp
from oracle import oracle as orc
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import optimize_batch

n = int(sys.argv[1])
case, text = fp.make_case(n)
print(text, case['kw'])
blocks, kw = case['blocks'], case['kw']
prob = Problem(blocks, case['Z'], case['freq'], **kw)
om = orc.OracleModel(blocks, case['Z'][0], case['freq'], **kw)
th0 = np.random.default_rng(n + 7).uniform(-2, 2, (1, prob.D))
out, rep = optimize_batch(prob, th0, max_iter=2000)
print(rep[0])
lp_d, g_d = prob.logp_grad(out, jacobian=False)
lr, gr = om.logp_grad(out[0], False)
print('device lp %.12g |g| %.3g ; oracle lp %.12g' % (lp_d[0], np.max(np.abs(g_d[0])), lr))
j = int(np.argmax(np.abs(g_d[0])))
print('largest gradient entry: index %d, value %.3g, theta %.6g' % (j, g_d[0][j], out[0][j]))
print('theta range', out[0].min(), out[0].max(), 'non-finite', np.sum(~np.isfinite(out[0])))
lay = prob.layout() if hasattr(prob, 'layout') else None
print('layout', lay)
params = prob.transformed(out)[0][0]
print('use_x_sum' , kw.get('use_x_sum'), {k: v for k, v in kw.items() if 'x_sum' in k})
for b, blk in enumerate(blocks):
    print('block', b, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in blk.items() if k not in ('A', 'L0', 'L1', 'L2')})

