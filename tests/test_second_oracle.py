"""The package-form Series outlier models (S2) and the single parallel distribution (S6) are pinned by no stored Stan
result (SURVEY 8(c) "unpinned").  Here the C oracle -- and, on the GPU box, the HIP path through the C ABI -- is
checked against an INDEPENDENT transcription of the Stan text into torch with autograd gradients (tests/stan_torch.py):
lp, gradient and the transformed parameters, both Jacobian modes.  S1 runs through the same harness as a control (S1 is
additionally pinned by the 36 stored `optimizing` results)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import stan_torch as st
from tests.helpers import load

TOL = 1e-10


def _series_outlier_case(pos, tag='K81', mode='sample'):
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    so = load('dat_%s_outlier_scalars' % mode)
    nf = len(d['freq'])
    dat = dict(N=nf, K=int(d['K']), A=d['A'], Z=d['Z'], freq=d['freq'], L0=d['L0'], L1=d['L1'], L2=d['L2'],
               sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
               induc_scale=float(d['induc_scale']), sigma_out_lambda=float(so['sigma_out_lambda']),
               sigma_out_alpha=float(so['sigma_out_alpha']), sigma_out_beta=float(so['sigma_out_beta']))
    blocks = [dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)]
    kw = dict(sigma_min=dat['sigma_min'], ups_alpha=dat['ups_alpha'], ups_beta=dat['ups_beta'], induc_scale=dat['induc_scale'],
              outlier_mode=1, so_lambda=dat['sigma_out_lambda'], so_alpha=dat['sigma_out_alpha'], so_beta=dat['sigma_out_beta'])
    return dat, blocks, d['Z'], d['freq'], kw, (st.series_outliers_lp, (dat, None, pos))


def _cmp(fn, args_of_jac, m, theta, jac):
    lp_t, g_t, tp = st.lp_and_grad(fn, theta, *args_of_jac(jac))
    lp_o, g_o = m.logp_grad(theta, jac)
    fw = m.forward(theta)
    assert abs(lp_o - lp_t) <= TOL * max(1.0, abs(lp_t)), (lp_o, lp_t)
    assert np.max(np.abs(g_o - g_t)) <= TOL * max(1.0, np.max(np.abs(g_t)))
    assert np.allclose(fw['Z_hat'], tp['Z_hat'], rtol=1e-12, atol=1e-13)
    assert np.allclose(fw['sigma_tot'], tp['sigma_tot'], rtol=1e-12, atol=0)
    assert np.allclose(fw['q'], tp['q'], rtol=1e-10, atol=1e-13)
    return lp_t, g_t


@pytest.mark.parametrize('pos', [True, False])
@pytest.mark.parametrize('jac', [True, False])
@pytest.mark.parametrize('mode', ['sample', 'optimize'])
def test_series_outliers_package_form_oracle_vs_stan_text(pos, jac, mode):
    dat, blocks, Z, freq, kw, _ = _series_outlier_case(pos, mode=mode)
    m = orc.OracleModel(blocks, Z, freq, **kw)
    assert m.D == 2 * dat['K'] + 9 + 2 * dat['N']
    rng = np.random.default_rng(11)
    for _ in range(3):
        theta = rng.uniform(-1.5, 1.5, m.D)
        _cmp(st.series_outliers_lp, lambda j: (dat, j, pos), m, theta, jac)


@pytest.mark.parametrize('jac', [True, False])
@pytest.mark.parametrize('mode', ['sample', 'optimize'])
def test_parallel_model_oracle_vs_stan_text(jac, mode):
    dat, blocks, Z, freq, kw = parallel_case(mode)
    m = orc.OracleModel(blocks, Z, freq, **kw)
    assert m.D == 2 * dat['K'] + 9
    rng = np.random.default_rng(12)
    for _ in range(3):
        theta = rng.uniform(-1.5, 1.5, m.D)
        _cmp(st.parallel_lp, lambda j: (dat, j), m, theta, jac)


@pytest.mark.parametrize('pos', [True, False])
@pytest.mark.parametrize('jac', [True, False])
def test_series_control_oracle_vs_stan_text(pos, jac):
    d = load('dat_sample_2ZARC_uniform_0.25_K161')
    dat = dict(N=int(d['N']), K=int(d['K']), A=d['A'], Z=d['Z'], freq=d['freq'], L0=d['L0'], L1=d['L1'], L2=d['L2'],
               sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
               induc_scale=float(d['induc_scale']))
    m = orc.OracleModel([dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)], d['Z'], d['freq'],
                        sigma_min=dat['sigma_min'], ups_alpha=dat['ups_alpha'], ups_beta=dat['ups_beta'])
    theta = np.random.default_rng(13).uniform(-2, 2, m.D)
    _cmp(st.series_lp, lambda j: (dat, j, pos), m, theta, jac)


def parallel_case(mode):
    """Stan data of the Parallel model for a transmissive planar DDT, as Inverter._prep_stan_data assembles it: the
    admittance matrices come from the reference (golden), the L scalings from its hyper-parameter table."""
    g = load('ddt_toeplitz_81x161')
    base = load('dat_%s_2ZARC_uniform_0.25_K161' % mode)
    A = np.vstack([g['A_re_tp_parallel'], g['A_im_tp_parallel']])
    nf, K = g['A_re_tp_parallel'].shape
    rng = np.random.default_rng(3)
    # a plausible admittance-generated spectrum: Z = 1/(A x) + offsets + noise
    x_true = 0.3 * np.exp(-0.5 * ((np.arange(K) - 0.55 * K) / 9.0) ** 2) + 1e-3
    Y = A @ x_true
    Zc = 1.0 / (Y[:nf] + 1j * Y[nf:]) + 0.4
    Zs = np.concatenate([Zc.real, Zc.imag]) + 0.002 * rng.standard_normal(2 * nf)
    dat = dict(N=2 * nf, K=K, A=A, Z=Zs, freq=g['freq'], L0=base['L0'], L1=base['L1'], L2=base['L2'],
               sigma_min=0.002, ups_alpha=float(base['ups_alpha']), ups_beta=float(base['ups_beta']), induc_scale=1.0)
    blocks = [dict(A=A, L0=base['L0'], L1=base['L1'], L2=base['L2'], parallel=True, nonneg=True, x_scale=1.0)]
    kw = dict(sigma_min=0.002, ups_alpha=dat['ups_alpha'], ups_beta=dat['ups_beta'], induc_scale=1.0, use_x_sum=False)
    return dat, blocks, Zs, g['freq'], kw


# ------------------------------------------------------------------------------------------------ through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize('family', ['series_pos_outliers', 'series_outliers', 'parallel'])
@pytest.mark.parametrize('jac', [True, False])
def test_hip_path_vs_stan_text(family, jac):
    """libbdrt.so (bdrt_logp_grad) against the torch transcription directly -- not via the C oracle."""
    from bayes_drt_amd.model import Problem
    rng = np.random.default_rng(21)
    if family == 'parallel':
        dat, blocks, Z, freq, kw = parallel_case('sample')
        fn, args = st.parallel_lp, lambda j: (dat, j)
    else:
        pos = family == 'series_pos_outliers'
        dat, blocks, Z, freq, kw, _ = _series_outlier_case(pos, tag='K161')
        fn, args = st.series_outliers_lp, lambda j: (dat, j, pos)
    prob = Problem(blocks, Z, freq, **kw)
    theta = rng.uniform(-1.5, 1.5, (5, prob.D))
    lp, g = prob.logp_grad(theta, jacobian=jac)
    _, Zh, sg = prob.transformed(theta)
    for i in range(5):
        lp_t, g_t, tp = st.lp_and_grad(fn, theta[i], *args(jac))
        assert abs(lp[i] - lp_t) <= TOL * max(1.0, abs(lp_t)), (family, lp[i], lp_t)
        assert np.max(np.abs(g[i] - g_t)) <= TOL * max(1.0, np.max(np.abs(g_t)))
        assert np.allclose(Zh[i], tp['Z_hat'], rtol=1e-11, atol=1e-12)
        assert np.allclose(sg[i], tp['sigma_tot'], rtol=1e-11, atol=0)
    prob.close()
