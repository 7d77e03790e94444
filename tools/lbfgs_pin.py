"""Stan's L-BFGS termination against the stored Stan iterates: for every usable stored MAP (tests/golden/kat_*.npz) the Stan-style
L-BFGS (algorithm='LBFGS': newton_max_iter = 0, one start) is started AT the stored point -- where Stan's own run stopped by one
of its tolerance tests (reference call site bayes_drt/inversion.py:1216; 50 000-iteration cap never reached: notebook wall times) --
and the record says whether it stops there too: iterations, return code, gain in lp, how far the coefficients / the predicted
spectrum move."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import kat_names, kat_to_model, rel_l2
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import optimize_batch

rows = []
print('%-34s %-28s %7s %3s %12s %10s %10s %10s %9s' % ('stored fit', 'model', 'its', 'rc', 'lp gain', 'coef dL2', 'ident dL2', 'Z_hat dL2', '|g|inf at start'))
for name in kat_names():
    k = kat_to_model(name)
    if k is None or not k['has_Z']:
        continue
    prob = Problem(**k['kw'])
    lay = prob.layout()
    th = prob.unconstrain(k['params'])
    lp0, g0 = prob.logp_grad(th[None], jacobian=False)
    out, rep = optimize_batch(prob, th[None], newton_max_iter=0)
    con = prob.constrain(out)
    K0 = prob.Ks[0]
    def coef(p):
        return np.concatenate([p[lay['x'][b]:lay['x'][b] + K] for b, K in enumerate(prob.Ks)])
    d = rel_l2(coef(con[0]), coef(k['params']))
    A0 = np.asarray(k['kw']['blocks'][0]['A'], dtype=float)
    U, sv, Vt = np.linalg.svd(A0, full_matrices=False)
    V = Vt[sv >= 1e-2 * sv[0]]
    xo, xr = con[0][lay['x'][0]:lay['x'][0] + K0], k['params'][lay['x'][0]:lay['x'][0] + K0]
    dp = float(np.linalg.norm(V @ (xo - xr)) / np.linalg.norm(V @ xr))
    _, Zh, _ = prob.transformed(out)
    dz = rel_l2(Zh[0], k['opt']['Z_hat'])
    rows.append((rep[0]['iterations'], rep[0]['return_code'], rep[0]['lp'] - lp0[0], d, dp, dz))
    print('%-34s %-28s %7d %3d %12.4e %10.3e %10.3e %10.3e %9.2e' % (name, k['family'], rep[0]['iterations'], rep[0]['return_code'],
          rep[0]['lp'] - lp0[0], d, dp, dz, np.max(np.abs(g0))), flush=True)
    prob.close()
r = np.array(rows)
print('iterations: median %d, 90th percentile %d, max %d; stopped by a tolerance test (rc 0): %d of %d' % (
    np.median(r[:, 0]), np.percentile(r[:, 0], 90), r[:, 0].max(), int(np.sum(r[:, 1] == 0)), len(r)))
print('movement of the coefficients on the well-determined directions: median %.2e, 90th percentile %.2e, max %.2e; of Z_hat: median %.2e, max %.2e' % (
    np.median(r[:, 4]), np.percentile(r[:, 4], 90), r[:, 4].max(), np.median(r[:, 5]), r[:, 5].max()))
