"""The tuned CPU evaluator of the headline family (oracle/bdrt_tuned.c: bench.py's cpu_baseline.tuned leg) against the oracle:
same log-posterior and gradient, dense and banded form, both sign conventions, with and without the Jacobian."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load


@pytest.mark.parametrize('tag', ['K81', 'K161'])
@pytest.mark.parametrize('pos', [True, False])
def test_tuned_evaluator_equals_the_oracle(tag, pos):
    d = load('dat_sample_2ZARC_uniform_0.25_' + tag)
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']))
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    rng = np.random.default_rng(3)
    for banded in (True, False):
        tm = orc.TunedS1(blk, d['Z'], d['freq'], banded=banded, **kw)
        assert tm.D == om.D and tm.banded == banded
        for _ in range(4):
            th = rng.uniform(-2, 2, om.D)
            for jac in (True, False):
                lp, g = tm.logp_grad(th, jac)
                lp_ref, g_ref = om.logp_grad(th, jac)
                assert abs(lp - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (banded, jac, lp, lp_ref)
                assert np.max(np.abs(g - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), (banded, jac)
