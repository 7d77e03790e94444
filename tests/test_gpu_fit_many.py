"""GPU tests (-m gpu) of Inverter.fit_many: the reference's workload -- a loop of `fit` calls over spectra measured on one
grid (code_EchemActa/Run fits.ipynb cells 4-5; bayes_drt/inversion.py:1072-1081, :1218-1221) -- as one batch, and that batch
sharded over the ranks of a process group.  Each spectrum's result must be the one a separate `fit` call gives."""
import os
import socket
import warnings

import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu
TAU_PLOT = np.logspace(-7, 2, 120)


def _spectra(n=3):
    c = load('csv_2ZARC_uniform_0.25')
    Z = c['Z']
    f, z0 = Z[:, 0], Z[:, 1] + 1j * Z[:, 2]
    rs = np.random.RandomState(5)
    zs = [z0] + [z0 * (1.0 + 0.3 * k) + 0.003 * (rs.standard_normal(len(f)) + 1j * rs.standard_normal(len(f))) for k in range(1, n)]
    return f, zs


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('init_from_ridge', [False, True])
def test_fit_many_sample_equals_separate_fit_calls(init_from_ridge):
    """Same seeds => same draws: units (spectrum, chain) of the batch carry the RNG streams (seed, chain) of the separate calls,
    every spectrum is scaled as `fit` scales it, and with init_from_ridge each spectrum starts from its own ridge solution."""
    from bayes_drt_amd.inversion import Inverter
    f, zs = _spectra(3)
    kw = dict(nonneg=True, mode='sample', warmup=30, samples=20, chains=2, random_seed=77, init_from_ridge=init_from_ridge)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        base = Inverter(basis_freq=f)
        views = base.fit_many(f, zs, **kw)
        assert len(views) == 3 and not hasattr(base, '_sample_result')
        for Z, v in zip(zs, views):
            one = Inverter(basis_freq=f)
            one.fit(f, Z, **kw)
            assert v.fit_type == 'bayes' and v.stan_model_name == one.stan_model_name
            assert np.array_equal(v._sample_result.theta, one._sample_result.theta)
            assert np.array_equal(v._sample_result.lp, one._sample_result.lp)
            assert v._Z_scale == one._Z_scale and v.R_inf == one.R_inf
            assert np.array_equal(v.predict_distribution('DRT', eval_tau=TAU_PLOT), one.predict_distribution('DRT', eval_tau=TAU_PLOT))
            assert np.array_equal(v.coef_percentile('DRT', 97.5), one.coef_percentile('DRT', 97.5))
            assert np.array_equal(v.predict_Z(f, percentile=50), one.predict_Z(f, percentile=50))
    assert not np.array_equal(views[0]._sample_result.theta, views[1]._sample_result.theta)


def test_fit_many_map_equals_separate_fit_calls():
    """mode='optimize': every (spectrum, start) fit in one lock-step batch; the start that wins per spectrum and its stationary
    point are those of the separate call (the batched Newton iteration reproduces single fits bit for bit)."""
    from bayes_drt_amd.inversion import Inverter
    f, zs = _spectra(4)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        views = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
        for Z, v in zip(zs, views):
            one = Inverter(basis_freq=f)
            one.fit(f, Z, nonneg=True, mode='optimize')
            assert v.fit_type == 'map' and v._opt_report['return_code'] == 0
            assert v._opt_report['start'] == one._opt_report['start'] and len(v._opt_report['starts']) == len(one._opt_report['starts'])
            assert np.allclose(v.distribution_fits['DRT']['coef'], one.distribution_fits['DRT']['coef'], rtol=1e-9, atol=1e-12)
            assert v.R_inf == pytest.approx(one.R_inf, rel=1e-10)
            assert np.allclose(v.predict_Z(f), one.predict_Z(f), rtol=1e-9)


def test_fit_many_groups_spectra_by_model_and_rejects_mixed_grids():
    from bayes_drt_amd.inversion import Inverter
    f, zs = _spectra(2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        views = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, outliers=True, mode='optimize', n_starts=1)
    assert all(v.stan_model_name == 'Series_pos_outliers_StanModel.pkl' for v in views)
    assert all('sigma_out' in v.error_fit for v in views)
    assert Inverter(basis_freq=f).fit_many(f, [], nonneg=True) == []
    with pytest.raises(ValueError):
        Inverter(basis_freq=f).fit_many(f, [zs[0][:-1]], nonneg=True)


def _rank_main(rank, world, port, mode, q):
    """One process per rank with the real GPU worker on the box's single device (gloo carries the collectives: RCCL refuses two
    ranks on one device); every rank calls fit_many with the same arguments."""
    import torch.distributed as dist
    from bayes_drt_amd.inversion import Inverter
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        f, zs = _spectra(3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if mode == 'sample':
                views = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='sample', warmup=30, samples=20, chains=2, random_seed=77)
                out = [(v._sample_result.theta, v._sample_result.lp, v.predict_distribution('DRT', eval_tau=TAU_PLOT)) for v in views]
            else:
                views = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
                out = [(v.distribution_fits['DRT']['coef'], v._opt_report['lp'], v.predict_distribution('DRT', eval_tau=TAU_PLOT)) for v in views]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('mode', ['sample', 'optimize'])
def test_fit_many_sharded_over_two_ranks_equals_one_process(mode):
    """Inside a process group the batch is sharded (whole spectra per rank for the sampler, rows of the start table for the
    optimiser); every rank returns the full list, equal to the single-process call."""
    import torch.multiprocessing as mp
    from bayes_drt_amd.inversion import Inverter
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=400) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    f, zs = _spectra(3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if mode == 'sample':
            ref = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='sample', warmup=30, samples=20, chains=2, random_seed=77)
            exp = [(v._sample_result.theta, v._sample_result.lp, v.predict_distribution('DRT', eval_tau=TAU_PLOT)) for v in ref]
        else:
            ref = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
            exp = [(v.distribution_fits['DRT']['coef'], v._opt_report['lp'], v.predict_distribution('DRT', eval_tau=TAU_PLOT)) for v in ref]
    for rank in (0, 1):
        for a, b in zip(got[rank], exp):
            for x, y in zip(a, b):
                if mode == 'sample':
                    assert np.array_equal(np.asarray(x), np.asarray(y))
                else:
                    assert np.allclose(np.asarray(x), np.asarray(y), rtol=1e-9, atol=1e-12)


def test_view_returned_by_fit_many_can_be_refitted():
    """A view of fit_many(mode='optimize') is an ordinary Inverter: a later `fit` on it solves its own ridge starting point
    (the deferral of the batch is an argument of the batch's prepare step, not a property left on the views)."""
    from bayes_drt_amd.inversion import Inverter
    f, zs = _spectra(2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        views = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
        coef = views[0].distribution_fits['DRT']['coef'].copy()
        assert not hasattr(views[0], '_defer_ridge_start')
        views[0].fit(f, zs[0], nonneg=True, mode='optimize')
        assert len(views[0]._opt_report['starts']) == 2
        assert np.allclose(views[0].distribution_fits['DRT']['coef'], coef, rtol=1e-9, atol=1e-12)
        views[1].fit(f, zs[0], nonneg=True, mode='optimize')
        assert np.allclose(views[1].distribution_fits['DRT']['coef'], coef, rtol=1e-9, atol=1e-12)


def _truncated_study():
    """The reference's truncated-spectrum study (code_EchemActa/Run fits.ipynb cells 12-14): spectra of 53 frequencies and the
    full-range one of 91 on the basis of 81, sigma_min = 0.005 for the noiseless spectra and 0.002 otherwise."""
    names = ['trunc_uniform_0.25', 'trunc_noiseless', 'trunc_noiseless_FullRange', 'trunc_Orazem_1.0', 'trunc_Macdonald_2.5']
    fs, zs, sm = [], [], []
    for n in names:
        d = load('kat_' + n)
        fs.append(np.array(d['data_freq'], dtype=float)); zs.append(np.array(d['data_Z']))
        sm.append(0.005 if 'noiseless' in n else 0.002)
    return names, fs, zs, sm


@pytest.mark.parametrize('mode', ['optimize', 'sample'])
def test_fit_many_over_mixed_grids_and_options_equals_separate_fit_calls(mode):
    """One call for spectra of different lengths and options: grouped by (grid, basis, model, options), one batch per group,
    results in input order and equal to the separate `fit` calls of the notebook's loop."""
    from bayes_drt_amd.inversion import Inverter
    names, fs, zs, sm = _truncated_study()
    assert sorted({len(f) for f in fs}) == [53, 91]
    kw = dict(mode='optimize') if mode == 'optimize' else dict(mode='sample', warmup=30, samples=20, chains=2, random_seed=11)
    basis = np.logspace(6, -2, 81)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        base = Inverter(basis_freq=basis)
        views = base.fit_many(fs, zs, sigma_min=sm, nonneg=False, **kw)
        assert len(views) == len(names) and np.array_equal(np.atleast_1d(base.f_train), np.atleast_1d(Inverter(basis_freq=basis).f_train))
        for f, Z, s, v in zip(fs, zs, sm, views):
            one = Inverter(basis_freq=basis)
            one.fit(f, Z, sigma_min=s, nonneg=False, **kw)
            assert len(v.f_train) == len(f) and v._stan_input['sigma_min'] == s
            if mode == 'sample':
                assert np.array_equal(v._sample_result.theta, one._sample_result.theta)
                assert np.array_equal(v.predict_distribution('DRT', eval_tau=TAU_PLOT, percentile=97.5),
                                      one.predict_distribution('DRT', eval_tau=TAU_PLOT, percentile=97.5))
            else:
                assert v._opt_report['start'] == one._opt_report['start']
                assert np.allclose(v.distribution_fits['DRT']['coef'], one.distribution_fits['DRT']['coef'], rtol=1e-8, atol=1e-11)
            assert np.allclose(v.predict_Z(f), one.predict_Z(f), rtol=1e-8)
            assert np.allclose(v.error_fit['sigma_min'], one.error_fit['sigma_min'])


def test_fit_many_leaves_a_fitted_instance_and_its_matrix_cache_alone():
    """`fit_many` on another grid must not leave matrices of that grid in the cache of the instance it was called on."""
    from bayes_drt_amd.inversion import Inverter
    f, zs = _spectra(1)
    _, fs, zt, _ = _truncated_study()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv = Inverter(basis_freq=f)
        inv.fit(f, zs[0], nonneg=True, mode='optimize')
        A0 = inv.distribution_matrices['DRT']['A_re'].copy()
        zhat = inv.predict_Z(f)
        inv.fit_many([fs[0]], [zt[0]], nonneg=True, mode='optimize', n_starts=1)
        assert np.array_equal(inv.distribution_matrices['DRT']['A_re'], A0)
        assert np.array_equal(inv.predict_Z(f), zhat)
