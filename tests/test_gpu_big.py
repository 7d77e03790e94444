"""GPU tests (-m gpu) of the path for problems BEYOND the LDS budget of the tile / one-chain evaluators (bdrt_big.h, nuts_big_kernel):
the reference builds whatever grids it is given (bayes_drt/inversion.py:2127-2209) and Stan has no size limit.  One workgroup per
point / chain, vectors in an HBM workspace, plain copies of the matrices: slow but working, against the oracle as everything else."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import kat_names, kat_to_model

pytestmark = pytest.mark.gpu


def _big_problem(nf=200, K=301, pos=True, seed=1):
    from bayes_drt_amd import matrices as gm
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    f = np.logspace(6, -3, nf)
    bf = np.logspace(8, -5, K)
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    w = 2 * np.pi * f
    z = 1.0 + 1.2 / (1 + (1j * w * 3e-3) ** 0.8) + 0.7 / (1 + (1j * w * 2e-1) ** 0.9)
    rs = np.random.RandomState(seed)
    z = z + 0.003 * (rs.normal(size=nf) + 1j * rs.normal(size=nf))
    z_raw = z.copy()
    z = z / (np.std(np.abs(z)) / np.sqrt(nf / 81))
    Z = np.concatenate([z.real, z.imag])
    blk = dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=pos)
    kw = dict(sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1)
    return Problem([blk], Z, f, **kw), orc.OracleModel([blk], Z, f, **kw), (f, z_raw, bf)


@pytest.mark.parametrize('pos', [True, False])
def test_200_frequencies_301_basis_functions_evaluate_like_the_oracle(pos):
    prob, om, _ = _big_problem(pos=pos)
    assert prob.evaluator() == 5 and prob.D == 2 * 301 + 9
    rng = np.random.default_rng(4)
    th = rng.uniform(-2, 2, (20, prob.D))
    for jac in (True, False):
        lp, g = prob.logp_grad(th, jacobian=jac)
        for i in (0, 7, 19):
            lp_ref, g_ref = om.logp_grad(th[i], jac)
            assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
            assert np.max(np.abs(g[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), i
    con, Zh, sg = prob.transformed(th[:3])
    for i in range(3):
        fw = om.forward(th[i])
        assert np.allclose(Zh[i], fw['Z_hat'], rtol=1e-11, atol=1e-12) and np.allclose(sg[i], fw['sigma_tot'], rtol=1e-11, atol=1e-13)
    prob.close()


@pytest.mark.parametrize('name', ['PDAC_DRT-TpDDT_outliers', 'DRT-TpDDT-BpDDT_uniform_0.25', 'PDAC_outliers', 'LIB_data'])
def test_every_model_family_through_the_streamed_evaluator(name, monkeypatch):
    """The stored fits of the other families (series + parallel blocks, both outlier models, x_sum prior) evaluated by the streamed
    evaluator (BDRT_BIG=1 forces it on problems that fit the tiles) against the tile evaluator and the oracle."""
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    k = kat_to_model(name)
    tile = Problem(**k['kw'])
    monkeypatch.setenv('BDRT_BIG', '1')
    big = Problem(**k['kw'])
    monkeypatch.delenv('BDRT_BIG')
    assert big.evaluator() == 5 and tile.evaluator() != 5
    om = orc.OracleModel(k['kw']['blocks'], k['kw']['Z'], k['kw']['freq'], **{a: b for a, b in k['kw'].items() if a not in ('blocks', 'Z', 'freq')})
    th0 = tile.unconstrain(k['params'])
    rng = np.random.default_rng(2)
    th = th0[None] + 0.3 * rng.standard_normal((6, tile.D))
    for jac in (True, False):
        lp_t, g_t = tile.logp_grad(th, jacobian=jac)
        lp_b, g_b = big.logp_grad(th, jacobian=jac)
        ok = np.isfinite(lp_t)
        assert np.array_equal(np.isfinite(lp_b), ok)
        assert np.allclose(lp_b[ok], lp_t[ok], rtol=1e-11, atol=1e-9)
        assert np.max(np.abs(g_b[ok] - g_t[ok])) <= 1e-10 * max(1.0, np.max(np.abs(g_t[ok])))
        lp_ref, g_ref = om.logp_grad(th[0], jac)
        if np.isfinite(lp_ref):
            assert abs(lp_b[0] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref))
    tile.close(); big.close()


def test_big_problem_map_and_a_short_nuts_run_against_the_oracle():
    """150 frequencies x 221 basis functions (D = 451): MAP (Newton iteration on the streamed evaluator) reaches a stationary
    point whose log-posterior the oracle confirms; a short NUTS run equals the oracle's draw by draw."""
    from bayes_drt_amd._lib import NutsControl
    from bayes_drt_amd.engine import Sampler, optimize_batch
    from oracle import oracle as orc
    prob, om, _ = _big_problem(nf=150, K=221)
    assert prob.evaluator() == 5
    th0 = np.random.RandomState(3).uniform(-2, 2, (1, prob.D))
    out, rep = optimize_batch(prob, th0)
    assert rep[0]['return_code'] == 0 and rep[0]['grad_inf'] < 1e-6, rep[0]
    lp_ref, g_ref = om.logp_grad(out[0], False)
    assert abs(rep[0]['lp'] - lp_ref) <= 1e-9 * max(1.0, abs(lp_ref)) and np.max(np.abs(g_ref)) < 1e-5
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 5
    with Sampler(prob, 3, 10, 4, 1234, ctrl) as smp:
        assert smp.kind() == 4
        smp.run()
        draws, lp, diag = smp.results()
    octrl = orc.nuts_control(max_treedepth=5)
    for c in range(3):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, 10, 4, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
    prob.close()


def test_ridge_fit_with_303_unknowns():
    """ridge_fit at 200 frequencies x 301 basis functions: 303 unknowns -- the KKT triangle of the QP kernel (coneqp restated,
    bdrt_qp.hip) lives in a global work buffer from n = 201 on, the hyper-lambda loop runs on the device as for the small grids."""
    import warnings
    from bayes_drt_amd.inversion import Inverter
    prob, om, (f, Z, bf) = _big_problem()
    prob.close()
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.ridge_fit(f, Z)
    coef = inv.distribution_fits['DRT']['coef']
    assert coef.shape == (301,) and np.all(np.isfinite(coef)) and np.all(coef >= -1e-9)
    rms = np.sqrt(np.mean(np.abs(inv.predict_Z(f) - Z) ** 2)) / np.std(np.abs(Z))
    assert rms < 0.02, rms
    # the polarisation resistance of the two arcs (1.2 + 0.7) comes back
    assert abs(inv.predict_Rp() - 1.9) < 0.15, inv.predict_Rp()


def test_inverter_fit_on_a_grid_beyond_the_tiles():
    """Inverter.fit end to end on 200 frequencies with a 301-point basis: MAP and a short HMC run follow the spectrum."""
    import warnings
    from bayes_drt_amd.inversion import Inverter
    prob, om, (f, Z, bf) = _big_problem()
    prob.close()
    inv = Inverter(basis_freq=bf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, nonneg=True)
        rms = np.sqrt(np.mean(np.abs(inv.predict_Z(f) - Z) ** 2)) / np.std(np.abs(Z))
        assert rms < 0.02, rms
        inv.fit(f, Z, nonneg=True, mode='sample', warmup=200, samples=50, chains=2)      # (the reference's warm-up length: a shorter one ends far from the mode)
        rms2 = np.sqrt(np.mean(np.abs(inv.predict_Z(f) - Z) ** 2)) / np.std(np.abs(Z))
        assert rms2 < 0.05, rms2
        assert inv._sample_result.n_divergent <= 10


def test_sampler_beyond_1024_parameters_on_the_streamed_path():
    """The streamed path's sampler took D <= 1024 until round 4 (one element pair per thread of its cooperative stage); the
    reference has no limit (three distributions of 301 basis functions are 1821 parameters).  Here: a series and a parallel
    distribution of 301 basis functions on 200 frequencies with the stacked outlier model, D = 1616 -- a short NUTS run equal to the
    oracle's, draw by draw (nuts_big_kernel<4>: four elements per thread, D <= 2048)."""
    from bayes_drt_amd import matrices as gm
    from bayes_drt_amd._lib import NutsControl
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    nf, K = 200, 301
    f = np.logspace(6, -3, nf)
    bf = np.logspace(8, -5, K)
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    blocks = []
    for par in (False, True):
        kw = dict(tau=tau, epsilon=eps) if not par else dict(tau=tau, epsilon=eps, kernel='DDT', dist_type='parallel', symmetry='planar', bc='transmissive')
        A = np.vstack([gm.construct_A(f, 'real', **kw), gm.construct_A(f, 'imag', **kw)])
        L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
        blocks.append(dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True, parallel=par, x_scale=0.8 if par else 1.0))
    w = 2 * np.pi * f
    z = 1.0 + 1.2 / (1 + (1j * w * 3e-3) ** 0.8) + 1.0 / (0.5 * np.sqrt(1j * w * 2.0) * np.tanh(np.sqrt(1j * w * 2.0)) + 1e-12)
    rs = np.random.RandomState(2)
    z = z + 0.003 * (rs.normal(size=nf) + 1j * rs.normal(size=nf))
    z = z / (np.std(np.abs(z)) / np.sqrt(nf / 81))
    Z = np.concatenate([z.real, z.imag])
    kw = dict(sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, outlier_mode=2, so_lambda=5.0, use_x_sum=True, x_sum_invscale=0.1)
    prob = Problem(blocks, Z, f, **kw)
    om = orc.OracleModel(blocks, Z, f, **kw)
    assert prob.evaluator() == 5 and prob.D == 2 + 2 * K + 4 + 2 * nf + 2 * K + 6 == 1616
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 4
    with Sampler(prob, 2, 6, 3, 1234, ctrl) as smp:
        assert smp.kind() == 4
        smp.run()
        draws, lp, diag = smp.results()
    octrl = orc.nuts_control(max_treedepth=4)
    for c in range(2):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, 6, 3, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
    prob.close()


@pytest.mark.parametrize('K,D', [(1200, 2409), (2100, 4209)])
def test_sampler_beyond_2048_parameters_on_the_streamed_path(K, D):
    """Round 6: the streamed path's sampler takes D <= 8192 (nuts_big_kernel<8> / <16>: eight / sixteen elements per thread of the
    cooperative stage; 2048 until round 5) -- one distribution of 1200 basis functions is 2409 parameters, of 2100 it is 4209.
    A short NUTS run equal to the oracle's, draw by draw."""
    from bayes_drt_amd._lib import NutsControl
    from bayes_drt_amd.engine import Sampler
    from oracle import oracle as orc
    prob, om, _ = _big_problem(nf=100, K=K)
    assert prob.evaluator() == 5 and prob.D == D
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 4
    with Sampler(prob, 2, 6, 3, 1234, ctrl) as smp:
        assert smp.kind() == 4
        smp.run()
        draws, lp, diag = smp.results()
    octrl = orc.nuts_control(max_treedepth=4)
    for c in range(2):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, 6, 3, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
    prob.close()
