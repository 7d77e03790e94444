"""Every stored Stan `optimizing` result of the reference (code_EchemActa/map_results/obj_*.pkl -> tests/golden/kat_*.npz)
against this build's MAP: starting from the stored point, the optimiser must reach a log-posterior at least as high (same
density: the stored transformed parameters pin it to 1e-12, tests/test_oracle_model.py); the distance between the stored
coefficients and the stationary point is reported per model family (the stored point is an un-converged L-BFGS iterate,
SURVEY fact 4).  Also from a random start (the reference's own kind of start)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import kat_names, kat_to_model, rel_l2
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import optimize_batch

print('%-34s %-30s %5s %12s %12s %9s %9s %4s | %12s %9s %4s' % ('stored fit', 'model', 'D', 'lp stored', 'lp ours', '|g|inf st.', 'coef dL2',
                                                              'rc', 'lp random', 'coef dL2', 'rc'))
for name in kat_names():
    k = kat_to_model(name)
    if k is None or not k['has_Z']:
        print('%-34s not usable (no data / pre-_raw snapshot)' % name); continue
    prob = Problem(**k['kw'])
    lay = prob.layout()
    th_ref = prob.unconstrain(k['params'])
    lp_ref, g_ref = prob.logp_grad(th_ref[None], jacobian=False)
    th0 = np.vstack([th_ref, np.random.RandomState(1234).uniform(-2, 2, prob.D)])
    out, rep = optimize_batch(prob, th0)
    con = prob.constrain(out)

    def coef(p):
        return np.concatenate([p[lay['x'][b]:lay['x'][b] + K] for b, K in enumerate(prob.Ks)])
    print('%-34s %-30s %5d %12.4f %12.4f %9.2e %9.3e %4d | %12.4f %9.3e %4d' % (
        name, k['family'], prob.D, lp_ref[0], rep[0]['lp'], np.max(np.abs(g_ref)), rel_l2(coef(con[0]), coef(k['params'])),
        rep[0]['return_code'], rep[1]['lp'], rel_l2(coef(con[1]), coef(k['params'])), rep[1]['return_code']), flush=True)
    prob.close()
