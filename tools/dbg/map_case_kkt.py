"""One fuzz case under both coordinate choices of the MAP iteration: where the two stationary points differ, and the KKT state of the
coefficients at the floor (g_x = g_y / x must be <= 0 there).  Usage: map_case_kkt.py <case>"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.fuzz_inverter import make_case
from bayes_drt_amd.inversion import Inverter
n = int(sys.argv[1])
case, text = make_case(n)
print(text)
res = {}
for tag, env in (('linear', None), ('log', '0')):
    if env is None: os.environ.pop('BDRT_NEWTON_LINEAR', None)
    else: os.environ['BDRT_NEWTON_LINEAR'] = env
    inv = Inverter(basis_freq=case['bf'])
    kw = dict(case['kw']); kw['outliers'] = False
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(case['f'], case['Z'], mode='optimize', n_starts=1, **kw)
    model = inv._get_stan_model(kw['nonneg'], False, False, None, False, False)[0]
    prob = model._prepare(inv._stan_input)
    o = inv._opt_result
    raw = np.concatenate([[o['Rinf_raw'], o['induc_raw']], o['x'], [o['sigma_res_raw'], o['alpha_prop_raw'], o['alpha_re_raw'], o['alpha_im_raw']], o['ups_raw'],
                          [o['d0_strength'], o['d1_strength'], o['d2_strength']]])
    th = prob.unconstrain(raw)
    lp_, g_ = prob.logp_grad(th[None], jacobian=False)
    K = len(o['x'])
    gx = g_[0][2:2 + K] / o['x'] if kw['nonneg'] else g_[0][2:2 + K]
    small = o['x'] <= 1e-10 * o['x'].max()
    print(tag, 'lp recomputed %.6f; coefficients below 1e-10 of the largest: %d; of them with g_x > 0 (want to grow): %d, largest g_x * xmax there %.3g' % (
        lp_[0], int(small.sum()), int((gx[small] > 0).sum()), float((gx[small] * o['x'].max()).max()) if small.any() else 0.0))
    res[tag] = (inv, prob)
    r = inv._opt_report
    x = inv._opt_result['x']
    print(tag, 'lp %.6f rounds %d |g| %.1e; x: max %.4g, at/below 1e-12 max: %d of %d' % (r['lp'], r['newton_iterations'], r['grad_inf'], x.max(), int((x <= 1e-12 * x.max()).sum()), len(x)))
xa, xb = res['linear'][0]._opt_result['x'], res['log'][0]._opt_result['x']
d = np.abs(xa - xb)
print('largest coefficient differences at', np.argsort(-d)[:8], d[np.argsort(-d)[:8]], 'linear:', xa[np.argsort(-d)[:8]], 'log:', xb[np.argsort(-d)[:8]])
