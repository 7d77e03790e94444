"""Oracle log-posterior (oracle/bdrt_oracle.c) pinned against the reference's stored Stan outputs.

* forward model: the 37 `StanModel.optimizing` results (code_EchemActa/map_results/obj_*.pkl ->
  tests/golden/kat_*.npz): Z_hat, sigma_tot, q, ups, dups recomputed from the stored parameters must equal
  the stored transformed parameters (<=1e-12 relative).
* gradient: central finite differences of the oracle's own log-density, and |grad|_inf at the stored MAPs
  equals the values probed in SURVEY.md 8(c)(5).
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import kat_names, kat_to_model, load, L_SCALE

NAMES = kat_names()


def test_all_37_kats_present():
    assert len(NAMES) == 37


@pytest.mark.parametrize('name', NAMES)
def test_forward_kat(name):
    k = kat_to_model(name)
    if k is None:
        pytest.skip('stored object predates the xp*_raw parameterisation or lacks the fitted frequency list')
    m = orc.OracleModel(**k['kw'])
    assert m.D == len(k['params'])
    theta = m.unconstrain(k['params'])
    out = m.forward(theta)
    opt = k['opt']

    def chk(a, b, what):
        b = np.asarray(b, dtype=float).ravel()
        err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        assert err <= 1e-12, (what, err)

    chk(out['Z_hat'], opt['Z_hat'], 'Z_hat')
    chk(out['sigma_tot'], opt['sigma_tot'], 'sigma_tot')
    if 'q' in opt:
        chk(out['q'], opt['q'], 'q'); chk(out['ups'], opt['ups'], 'ups'); chk(out['dups'], opt['dups'], 'dups')
    else:
        sfx = ['s', 'p'] if 'qp' in opt else ['s', 'p1', 'p2']
        chk(out['q'], np.concatenate([opt['q' + s] for s in sfx]), 'q')
        chk(out['ups'], np.concatenate([opt['ups_' + s] for s in sfx]), 'ups')
        chk(out['dups'], np.concatenate([opt['dups_' + s] for s in sfx]), 'dups')
        assert abs(out['x_sum'] - float(opt['x_sum'])) <= 1e-12


def _fd_check(m, theta, jac, rng, n=40, h=1e-5, rtol=2e-6):
    lp, g = m.logp_grad(theta, jacobian=jac)
    assert np.isfinite(lp)
    idx = rng.choice(m.D, size=min(n, m.D), replace=False)
    for j in idx:
        e = np.zeros(m.D); e[j] = h
        fd = (m.logp(theta + e, jac) - m.logp(theta - e, jac)) / (2 * h)
        assert abs(fd - g[j]) <= rtol * max(1.0, abs(g[j]), abs(fd)) + 1e-6 * np.max(np.abs(g)) * 1e-3, (j, fd, g[j])
    # directional derivative over all coordinates
    v = rng.standard_normal(m.D); v /= np.linalg.norm(v)
    fd = (m.logp(theta + h * v, jac) - m.logp(theta - h * v, jac)) / (2 * h)
    assert abs(fd - g @ v) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize('name', ['RC-ZARC_uniform_0.25', 'trunc_uniform_0.25', 'DRT-2-TpDDT_uniform_0.25',
                                  'DRT-TpDDT-BpDDT_uniform_0.25', 'PDAC_DRT-TpDDT_outliers', 'PDAC_outliers'])
@pytest.mark.parametrize('jac', [False, True])
def test_gradient_fd_at_kat(name, jac):
    k = kat_to_model(name)
    assert k is not None
    kw = dict(k['kw'])
    rng = np.random.default_rng(5)
    if not k['has_Z']:
        kw['Z'] = np.asarray(k['opt']['Z_hat']).ravel() + 0.01 * rng.standard_normal(len(kw['Z']))
    if kw['use_x_sum']:
        kw['x_sum_invscale'] = 0.3            # exercise the x_sum term (0 in optimize mode)
    m = orc.OracleModel(**kw)
    theta = m.unconstrain(k['params']) + 0.05 * rng.standard_normal(m.D)
    _fd_check(m, theta, jac, rng)


@pytest.mark.parametrize('mode', ['optimize', 'sample'])
def test_gradient_fd_benchmark_shape(mode):
    # the headline model: Series_pos, Nf=81, K=161 (D=331), built from the reference's own dat dict
    d = load('dat_%s_2ZARC_uniform_0.25_K161' % mode)
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    m = orc.OracleModel([blk], d['Z'], d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']),
                        ups_beta=float(d['ups_beta']), induc_scale=float(d['induc_scale']))
    assert m.D == 331 and int(d['N']) == 162 and int(d['K']) == 161
    rng = np.random.default_rng(1)
    theta = rng.uniform(-2, 2, m.D)
    _fd_check(m, theta, mode == 'sample', rng, h=1e-6, rtol=2e-5)


def test_outlier_mode1_gradient():
    # package-form Series_pos_outliers (Series_pos_outliers_modelcode.txt): unpinned by any stored fit,
    # checked by finite differences only
    d = load('dat_sample_2ZARC_uniform_0.25_K81')
    so = load('dat_sample_outlier_scalars')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    m = orc.OracleModel([blk], d['Z'], d['freq'], ups_alpha=1.0, ups_beta=0.1, outlier_mode=1,
                        so_lambda=float(so['sigma_out_lambda']), so_alpha=float(so['sigma_out_alpha']),
                        so_beta=float(so['sigma_out_beta']))
    assert m.D == 2 * 81 + 9 + 2 * 81
    rng = np.random.default_rng(2)
    _fd_check(m, rng.uniform(-1, 1, m.D), True, rng)


SURVEY_STATIONARITY = {'RC-ZARC_uniform_0.25': 0.12422, 'RC-ZARC_noiseless': 0.37255,
                       'RC-ZARC_Macdonald_1.0': 0.07686}


@pytest.mark.parametrize('name', sorted(SURVEY_STATIONARITY))
def test_stationarity_at_stored_map(name):
    """max|grad| (no Jacobian = `optimizing`) at the reference's stored MAP: small, and equal to the value
    probed during the survey with an independent torch-autograd transcription."""
    k = kat_to_model(name)
    assert k['has_Z']
    m = orc.OracleModel(**k['kw'])
    theta = m.unconstrain(k['params'])
    lp, g = m.logp_grad(theta, jacobian=False)
    assert abs(np.max(np.abs(g)) - SURVEY_STATIONARITY[name]) < 2e-4
    lpj, gj = m.logp_grad(theta, jacobian=True)
    # with the Jacobian every lower=0 coordinate gains +1
    pos = m.layout()['is_pos']
    np.testing.assert_allclose(gj - g, pos.astype(float), atol=1e-12)
    assert abs((lpj - lp) - theta[pos].sum()) < 1e-9


def test_x_sum_rejection():
    k = kat_to_model('DRT-2-TpDDT_uniform_0.25')
    kw = dict(k['kw']); kw['blocks'] = [dict(b) for b in kw['blocks']]
    kw['blocks'][0]['nonneg'] = False          # Series-Parallel (non-pos): xs unconstrained
    m = orc.OracleModel(**kw)
    lay = m.layout()
    p = k['params'].copy()
    p[lay['x'][0]:lay['x'][0] + m.Ks[0]] = -10.0
    theta = m.unconstrain(p)
    lp, g = m.logp_grad(theta, jacobian=True)
    assert lp == -np.inf and np.all(g == 0)
