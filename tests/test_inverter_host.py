"""Host logic of the Inverter mirror (no GPU): scaling, default tau grid, weights, the Stan data dict with its
hyper-parameter table, model selection and the prediction algebra -- against golden vectors produced by the
reference's own Inverter (tests/golden/dat_*.npz, predict_*.npz).  The GPU matrix builders are replaced by the CPU
oracle's builders through monkeypatching: test-only injection, the product has no such path."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load


@pytest.fixture
def inv_mod(monkeypatch):
    from bayes_drt_amd import inversion

    def cA(frequencies, part, tau=None, basis='gaussian', fit_inductance=False, epsilon=1, kernel='DRT',
           dist_type='series', symmetry='planar', bc=None, ct=False, k_ct=None, integrate_method='trapz'):
        return orc.construct_A(frequencies, part, tau=tau, epsilon=epsilon, kernel=kernel, dist_type=dist_type,
                               symmetry=symmetry, bc=bc if bc else None, ct=ct, k_ct=k_ct)

    def cL(frequencies, tau=None, basis='gaussian', epsilon=1, order=1):
        return orc.construct_L(tau, epsilon, order)

    def cM(frequencies, basis='gaussian', order=1, epsilon=1):
        return orc.construct_M(1 / (2 * np.pi * np.asarray(frequencies)), epsilon, order)
    monkeypatch.setattr(inversion, 'construct_A', cA)
    monkeypatch.setattr(inversion, 'construct_L', cL)
    monkeypatch.setattr(inversion, 'construct_M', cM)
    return inversion


@pytest.mark.parametrize('tag', ['K101', 'K161', 'K81'])
@pytest.mark.parametrize('mode', ['optimize', 'sample'])
def test_stan_data_matches_reference(inv_mod, tag, mode):
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    f, Z = d['freq_in'], d['Z_in']
    bf = {'K101': None, 'K161': np.logspace(10, -6, 161), 'K81': f}[tag]
    inv = inv_mod.Inverter(basis_freq=bf)
    fs, Zs, WZ_re, WZ_im, W_re, W_im, dm = inv._prep_matrices(f, Z, 'both', weights=None, dZ=False, scale_Z=True,
                                                              penalty='discrete', fit_type='map')
    assert inv._Z_scale == pytest.approx(float(d['Z_scale']), rel=1e-14)
    np.testing.assert_allclose(inv.distributions['DRT']['tau'], d['tau'], rtol=1e-13)
    assert len(inv.distributions['DRT']['tau']) == int(tag[1:])          # default grid length incl. int() truncation
    assert inv.distributions['DRT']['epsilon'] == pytest.approx(float(d['epsilon']), rel=1e-12)
    dat = inv._prep_stan_data(fs, Zs, 'both', 'Series', dm, False, 0.002, mode=mode, inductance_scale=1,
                              outlier_lambda=None, fitY=False, SA=False, SASY=False)
    for k in ('N', 'K', 'sigma_min', 'ups_alpha', 'ups_beta', 'induc_scale'):
        assert float(dat[k]) == float(d[k]), k
    np.testing.assert_allclose(dat['freq'], d['freq'], rtol=0)
    np.testing.assert_allclose(dat['Z'], d['Z'], rtol=1e-14)
    for k in ('A', 'L0', 'L1', 'L2'):
        assert np.max(np.abs(dat[k] - d[k])) <= 1e-11 * np.max(np.abs(d[k])), k


@pytest.mark.parametrize('mode', ['optimize', 'sample'])
@pytest.mark.parametrize('outl', [False, True])
def test_series_parallel_stan_data(inv_mod, mode, outl):
    d = load('dat_%s_DRT-TpDDT_%s' % (mode, 'outliers' if outl else 'plain'))
    dists = {'DRT': {'kernel': 'DRT'},
             'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
    inv = inv_mod.Inverter(basis_freq=np.logspace(10, -6, 161), distributions=dists)
    fs, Zs, *_, dm = inv._prep_matrices(d['freq_in'], d['Z_in'], 'both', weights=None, dZ=False, scale_Z=True,
                                        penalty='discrete', fit_type='map')
    model, name = None, None
    dat = inv._prep_stan_data(fs, Zs, 'both', 'Series-Parallel', dm, outl, 0.002, mode=mode, inductance_scale=1,
                              outlier_lambda=None, fitY=False, SA=False, SASY=False)
    for k in d.files:
        if k in ('Z_scale', 'freq_in', 'Z_in'):
            continue
        ref = d[k]
        assert k in dat, k
        if ref.ndim == 0:
            assert float(dat[k]) == pytest.approx(float(ref), rel=1e-14), k
        else:
            assert np.max(np.abs(np.asarray(dat[k]) - ref)) <= 1e-10 * max(np.max(np.abs(ref)), 1e-300), k
    assert inv._Z_scale == pytest.approx(float(d['Z_scale']), rel=1e-14)


def test_model_selection_names(inv_mod, monkeypatch):
    seen = []
    monkeypatch.setattr(inv_mod, 'load_pickle', lambda p: seen.append(p) or object())
    inv = inv_mod.Inverter()
    assert inv._get_stan_model(False, False, False, None, False, False)[1] == 'Series_StanModel.pkl'
    assert inv._get_stan_model(True, True, False, None, False, False)[1] == 'Series_pos_outliers_StanModel.pkl'
    d2 = {'DRT': {'kernel': 'DRT'}, 'P': {'kernel': 'DDT', 'dist_type': 'parallel'}}
    assert inv_mod.Inverter(distributions=d2)._get_stan_model(True, False, False, None, False, False)[1] == \
        'Series-Parallel_pos_StanModel.pkl'
    d3 = dict(d2, Q={'kernel': 'DDT', 'dist_type': 'parallel', 'bc': 'transmissive'})
    assert inv_mod.Inverter(distributions=d3)._get_stan_model(True, False, False, None, False, False)[1] == \
        'Series-2Parallel_pos_StanModel.pkl'
    dp = {'P': {'kernel': 'DDT', 'dist_type': 'parallel'}}
    assert inv_mod.Inverter(distributions=dp)._get_stan_model(True, False, False, None, False, False)[1] == \
        'Parallel_StanModel.pkl'


def test_ddt_defaults_and_validation(inv_mod):
    inv = inv_mod.Inverter(distributions={'D': {'kernel': 'DDT'}})
    assert inv.distributions['D'] == {'dist_type': 'parallel', 'symmetry': 'planar', 'bc': 'blocking', 'ct': False,
                                      'kernel': 'DDT'}
    with pytest.raises(ValueError):
        inv_mod.Inverter(distributions={'D': {'kernel': 'DDT', 'ct': True}})
    with pytest.raises(ValueError):
        inv_mod.Inverter(distributions={'D': {'kernel': 'DDT', 'symmetry': 'cubic'}})
    with pytest.raises(ValueError):
        inv_mod.Inverter().fit(np.ones(3), np.ones(4))


def test_prediction_algebra(inv_mod):
    p = load('predict_2ZARC_K161')
    inv = inv_mod.Inverter(basis_freq=np.logspace(10, -6, 161))
    Z = load('csv_2ZARC_uniform_0.25')['Z']
    f_in, Z_in = Z[:, 0], Z[:, 1] + 1j * Z[:, 2]
    inv._prep_matrices(f_in, Z_in, 'both', weights=None, dZ=False, scale_Z=True, penalty='discrete', fit_type='map')
    assert inv._Z_scale == pytest.approx(float(p['Z_scale']), rel=1e-14)
    inv.distribution_fits = {'DRT': {'coef': p['coef']}}
    inv.R_inf, inv.inductance = float(p['R_inf']), float(p['inductance'])
    inv.fit_type, inv.stan_model_name = 'map', 'Series_pos_StanModel.pkl'
    inv.error_fit = {k[4:]: float(p[k]) for k in p.files if k.startswith('err_')}
    np.testing.assert_allclose(inv.predict_distribution('DRT', eval_tau=p['tau_plot']), p['gamma'], rtol=1e-12)
    np.testing.assert_allclose(inv.predict_Z(p['freq']), p['Z_pred'], rtol=1e-10)
    np.testing.assert_allclose(inv.predict_Z(p['f_alt']), p['Z_pred_alt'], rtol=1e-10)
    s_re, s_im = inv.predict_sigma(p['f_alt'])
    np.testing.assert_allclose(s_re, p['sigma_re_alt'], rtol=1e-10)
    np.testing.assert_allclose(s_im, p['sigma_im_alt'], rtol=1e-10)
    assert inv.predict_Rp() == pytest.approx(float(p['Rp']), rel=1e-12)
    with pytest.raises(ValueError):
        inv.predict_distribution('DRT', percentile=50)


def test_weights(inv_mod):
    inv = inv_mod.Inverter()
    f = np.logspace(3, 0, 4); Z = np.array([1 + 1j, 2 - 1j, 3 + 0.5j, 1 - 2j])
    np.testing.assert_allclose(inv._format_weights(f, Z, None, 'both'), np.ones(4) * (1 + 1j))
    np.testing.assert_allclose(inv._format_weights(f, Z, 'modulus', 'both'), (1 + 1j) / np.abs(Z))
    np.testing.assert_allclose(inv._format_weights(f, Z, 'Orazem', 'both'), (1 + 1j) / (np.abs(Z.real) + np.abs(Z.imag)))
    np.testing.assert_allclose(inv._format_weights(f, Z, 2.0, 'real'), 2.0 + 1j * np.ones(4))
    with pytest.raises(ValueError):
        inv._format_weights(f, Z, 'nope', 'both')
    with pytest.raises(ValueError):
        inv._format_weights(f, Z, np.ones(3), 'both')


def test_format_weights_matches_reference(inv_mod):
    """Every named scheme / scalar form x part against the reference's own _format_weights (golden host_weights.npz);
    array weights (which the reference itself cannot take under numpy >= 1.25) by their documented meaning."""
    g = load('host_weights')
    f, Z = g['freq'], g['Z']
    inv = inv_mod.Inverter()
    cases = {'none': None, 'unity': 'unity', 'modulus': 'modulus', 'Orazem': 'Orazem', 'proportional': 'proportional',
             'prop_adj': 'prop_adj', 'float': 0.7, 'int': 3, 'complex': 0.3 + 1.2j}
    n = 0
    for name, w in cases.items():
        for part in ('both', 'real', 'imag'):
            key = 'w_%s_%s' % (name, part)
            if key in g.files:
                got = np.asarray(inv._format_weights(f, Z, w, part), dtype=complex)
                assert np.array_equal(got, g[key]), key
                n += 1
    assert n == 27
    ar, ac = g['arr_real'], g['arr_cplx']
    assert np.array_equal(inv._format_weights(f, Z, ar, 'both'), ar + 1j * ar)
    assert np.array_equal(inv._format_weights(f, Z, ar, 'imag'), 1 + 1j * ar)
    assert np.array_equal(inv._format_weights(f, Z, ar, 'real'), ar + 1j)
    assert np.array_equal(inv._format_weights(f, Z, ac, 'both'), ac)
    assert np.array_equal(inv._format_weights(f, Z, ac, 'imag'), ac)
    assert np.array_equal(inv._format_weights(f, Z, ac, 'real'), ac.real + 1j)
    with pytest.raises(ValueError):
        inv._format_weights(f, Z, 'nonsense', 'both')
    with pytest.raises(ValueError):
        inv._format_weights(f, Z, ar[:-1], 'both')
    with pytest.raises(ValueError):
        inv._format_weights(f, Z, None, 'neither')


def test_distribution_defaults_match_reference(inv_mod):
    g = load('host_distributions')
    dists = {'DRT': {'kernel': 'DRT'}, 'tp': {'kernel': 'DDT', 'bc': 'transmissive', 'dist_type': 'parallel'},
             'bs': {'kernel': 'DDT', 'symmetry': 'spherical'}, 'ct': {'kernel': 'DDT', 'ct': True, 'k_ct': 2.0, 'dist_type': 'series'}}
    for name, info in dists.items():
        got = inv_mod.Inverter(distributions={name: dict(info)}).distributions[name]
        assert ';'.join('%s=%s' % (k, got[k]) for k in sorted(got)) == str(g[name]), name
    for bad in ({'kernel': 'DDT', 'dist_type': 'diagonal'}, {'kernel': 'DDT', 'symmetry': 'cubic'}, {'kernel': 'DDT', 'bc': 'open'},
                {'kernel': 'DDT', 'ct': 'yes'}, {'kernel': 'DDT', 'ct': True}, {'kernel': 'XYZ'}):
        with pytest.raises(ValueError):
            inv_mod.Inverter(distributions={'d': bad})
    with pytest.warns(UserWarning):
        inv_mod.Inverter(distributions={'d': {'kernel': 'DRT', 'dist_type': 'parallel'}})
    with pytest.warns(UserWarning):
        inv_mod.Inverter(distributions={'d': {'kernel': 'DRT', 'bc': 'blocking'}})


def test_scale_Z_admittance_branch_matches_reference(inv_mod):
    """A single parallel planar DDT scales the ADMITTANCE to a fixed spread (reference inversion.py:2417-2434)."""
    g = load('host_scale_parallel')
    for bc in ('transmissive', 'blocking'):
        iv = inv_mod.Inverter(distributions={'d': {'kernel': 'DDT', 'dist_type': 'parallel', 'symmetry': 'planar', 'bc': bc}})
        Zs = iv._scale_Z(g['Z'], 'map')
        assert abs(iv._Z_scale - float(g['scale_' + bc])) <= 1e-14 * abs(float(g['scale_' + bc]))
        assert np.allclose(Zs, g['Zs_' + bc], rtol=1e-14, atol=0)
        iv._scale_Z(g['Z'], 'ridge')
        assert abs(iv._Z_scale - float(g['scale_ridge_' + bc])) <= 1e-14 * abs(float(g['scale_ridge_' + bc]))


def test_stan_data_for_single_part_fits_matches_reference(inv_mod):
    """fit(part='real' / 'imag'): the two-distribution models keep N = 2 Nf and zero the rows of the part that is not
    fitted (reference inversion.py:1892-1905; golden host_dat_parts.npz); for a single distribution the reference's data is
    dimensionally inconsistent (pystan would reject it), which is an error here too."""
    g = load('host_dat_parts')
    dists = {'DRT': {'kernel': 'DRT'},
             'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
    for part in ('real', 'imag'):
        inv = inv_mod.Inverter(basis_freq=g['basis_freq'], distributions=dists)
        fs, Zs, _, _, _, _, dm = inv._prep_matrices(g['freq'], g['Z'], part, weights=None, dZ=False, scale_Z=True,
                                                    penalty='discrete', fit_type='map')
        dat = inv._prep_stan_data(fs, Zs, part, 'Series-Parallel', dm, False, 0.002, mode='optimize', inductance_scale=1,
                                  outlier_lambda=None, fitY=False, SA=False, SASY=False)
        assert int(dat['N']) == int(g[part + '_N'])
        for k in ('Z', 'As', 'Ap'):
            assert np.allclose(dat[k], g['%s_%s' % (part, k)], rtol=1e-10, atol=1e-13), (part, k)
        half = len(fs)
        zero = slice(half, None) if part == 'real' else slice(0, half)
        assert not np.any(dat['Z'][zero]) and not np.any(dat['As'][zero]) and not np.any(dat['Ap'][zero])
    inv = inv_mod.Inverter(basis_freq=g['basis_freq'])
    fs, Zs, _, _, _, _, dm = inv._prep_matrices(g['freq'], g['Z'], 'real', weights=None, dZ=False, scale_Z=True,
                                                penalty='discrete', fit_type='map')
    with pytest.raises(ValueError):
        inv._prep_stan_data(fs, Zs, 'real', 'Series', dm, False, 0.002, mode='optimize', inductance_scale=1,
                            outlier_lambda=None, fitY=False, SA=False, SASY=False)


def test_ascending_basis_is_flagged(inv_mod):
    """The reference never sorts basis_freq; an ascending one flips the sign of its ln(tau) integrals.  Same here, with a warning."""
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        inv_mod.Inverter(basis_freq=np.logspace(6, -2, 21))
        assert not w
        inv_mod.Inverter(basis_freq=np.logspace(-2, 6, 21))
        assert len(w) == 1 and 'descending' in str(w[0].message)


def test_non_finite_spectra_are_refused_before_any_kernel(inv_mod):
    """(runs without a GPU: the check comes first)"""
    f = np.logspace(5, -1, 31)
    Z = 1.0 + 2.0 / (1 + (2j * np.pi * f * 1e-2) ** 0.9)
    for bad in (np.nan, np.inf):
        Zb = Z.copy(); Zb[4] = bad
        with pytest.raises(ValueError, match='finite'):
            inv_mod.Inverter(basis_freq=f).fit(f, Zb, nonneg=True)
    with pytest.raises(ValueError, match='positive'):
        inv_mod.Inverter(basis_freq=f).fit(np.r_[f[:-1], -1.0], Z)
    with pytest.raises(ValueError, match='equal'):
        inv_mod.Inverter(basis_freq=f).fit(f[:-1], Z)
