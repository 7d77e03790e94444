"""CPU test of the closed-form Hessian's numpy statement (tests/hessian_numpy.py) against central differences of the ORACLE's
gradient: the formulas the HIP kernels of csrc/bdrt_newton_hess.h restate (the GPU test holds those to this file's function)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load
from tests.hessian_numpy import series_hessian


@pytest.mark.parametrize('name,pos', [('dat_optimize_2ZARC_uniform_0.25_K81', True), ('dat_optimize_2ZARC_uniform_0.25_K81', False),
                                      ('dat_sample_2ZARC_uniform_0.25_K81', True)])
def test_numpy_hessian_equals_central_differences_of_the_oracle_gradient(name, pos):
    d = load(name)
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']))
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    D = om.D
    y = np.random.RandomState(4).uniform(-1, 1, D)
    lp0, g0 = om.logp_grad(y, False)
    lp, g, H = series_hessian(y, d['A'], (d['L0'], d['L1'], d['L2']), d['Z'], 2 * np.pi * d['freq'], pos=pos, **kw)
    assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0))
    assert np.max(np.abs(g - g0)) <= 1e-12 * np.max(np.abs(g0))
    Hfd = np.zeros((D, D))
    for j in range(D):
        h = 1e-6 * max(1.0, abs(y[j]))
        yp, ym = y.copy(), y.copy()
        yp[j] += h; ym[j] -= h
        Hfd[j] = (om.logp_grad(yp, False)[1] - om.logp_grad(ym, False)[1]) / (2 * h)
    Hfd = 0.5 * (Hfd + Hfd.T)
    assert np.max(np.abs(H - H.T)) <= 1e-12 * np.max(np.abs(H))
    assert np.max(np.abs(H - Hfd)) <= 1e-8 * np.max(np.abs(Hfd))
