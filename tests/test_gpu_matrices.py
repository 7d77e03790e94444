"""GPU parity (-m gpu): bdrt_build_A / _L / _M (HIP) vs the reference's golden matrices and the CPU oracle."""
import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu

DDT = {'tp': ('transmissive', 'planar'), 'bp': ('blocking', 'planar'), 'bs': ('blocking', 'spherical')}


def _close(a, b, rtol=1e-11):
    scale = np.max(np.abs(b))
    assert np.max(np.abs(a - b)) <= rtol * scale, (np.max(np.abs(a - b)), scale)


@pytest.mark.parametrize('case', ['81x81', '81x161', '41x51', 'general_12x9'])
def test_drt_matrices_vs_golden(case):
    from bayes_drt_amd import matrices as gm
    d = load('mat_drt_' + case)
    f, tau, eps = d['freq'], d['tau'], float(d['epsilon'])
    tau_arg = None if ('tau_is_none' in d.files and bool(d['tau_is_none'])) else tau
    for part in ('real', 'imag'):
        _close(gm.construct_A(f, part, tau=tau_arg, epsilon=eps), d['A_' + part[:2]])
    fb = 1 / (2 * np.pi * tau)
    for o in (0, 1, 2):
        _close(gm.construct_L(fb, tau=tau, epsilon=eps, order=o), d['L%d' % o])
        _close(gm.construct_M(fb, order=o, epsilon=eps), d['M%d' % o])
    if case == 'general_12x9':
        _close(gm.construct_L(fb, tau=tau, epsilon=eps, order=3), d['L3'])
        _close(gm.construct_L(fb, tau=tau, epsilon=eps, order=0.5), d['L0p5'])
        _close(gm.construct_L(fb, tau=tau, epsilon=eps, order=[0.2, 0.5, 0.3]), d['Lmix'])
        _close(gm.construct_M(fb, order=[0.2, 0.5, 0.3], epsilon=eps), d['Mmix'])


def _cmp_complex(Are, Aim, rre, rim, tol=1e-11):
    mag = np.sqrt(rre ** 2 + rim ** 2)
    ok = np.isfinite(rre) & np.isfinite(rim)
    err = np.maximum(np.abs(Are - rre), np.abs(Aim - rim))
    assert np.all(np.isfinite(Are[ok])) and np.all(err[ok] <= tol * mag[ok] + 1e-300), float(np.nanmax(err[ok] / mag[ok]))


@pytest.mark.parametrize('tag', ['tp', 'bp', 'bs'])
@pytest.mark.parametrize('dt', ['parallel', 'series'])
def test_ddt_toeplitz_vs_golden(tag, dt):
    from bayes_drt_amd import matrices as gm
    d = load('ddt_toeplitz_81x161')
    f, tau, eps = d['freq'], d['tau'], float(d['epsilon'])
    bc, sym = DDT[tag]
    A = [gm.construct_A(f, p, tau=tau, epsilon=eps, kernel='DDT', dist_type=dt, symmetry=sym, bc=bc) for p in ('real', 'imag')]
    # blocking-spherical: tanh x/(x - tanh x) cancels catastrophically for |x| << 1 in the reference itself
    # (x - tanh x ~ x^3/3): two correct fp64 evaluations differ by ~eps/|x|^2, hence the looser bound
    _cmp_complex(A[0], A[1], d['A_re_%s_%s' % (tag, dt)], d['A_im_%s_%s' % (tag, dt)], tol=1e-6 if tag == 'bs' else 1e-11)


@pytest.mark.parametrize('tag', ['tp', 'bp', 'bs'])
@pytest.mark.parametrize('ct', [0, 1])
def test_ddt_general_vs_golden(tag, ct):
    from bayes_drt_amd import matrices as gm
    d = load('ddt_general_7x6')
    f, tau, eps = d['freq'], d['tau'], float(d['epsilon'])
    bc, sym = DDT[tag]
    for dt in ('parallel', 'series'):
        A = [gm.construct_A(f, p, tau=tau, epsilon=eps, kernel='DDT', dist_type=dt, symmetry=sym, bc=bc, ct=bool(ct),
                            k_ct=float(d['k_ct']) if ct else None) for p in ('real', 'imag')]
        _cmp_complex(A[0], A[1], d['A_re_%s_%s_ct%d' % (tag, dt, ct)], d['A_im_%s_%s_ct%d' % (tag, dt, ct)],
                     tol=1e-6 if tag == 'bs' else 1e-11)


def test_full_size_general_path_vs_toeplitz():
    """Size-independent property at the benchmark shape: the general (nf*k wavefronts) path and the Toeplitz
    shortcut produce the same matrix on a log-uniform grid."""
    from bayes_drt_amd import _lib
    from bayes_drt_amd._lib import ptr
    lib = _lib.require_gpu()
    d = load('mat_drt_81x161')
    f, tau, eps = np.ascontiguousarray(d['freq']), np.ascontiguousarray(d['tau']), float(d['epsilon'])
    for part in (0, 1):
        out_t = np.empty((81, 161)); out_g = np.empty((81, 161))
        assert lib.bdrt_build_A(ptr(f), 81, ptr(tau), 161, eps, 0, part, 1, 0, 0.0, 1, ptr(out_t)) == 0
        assert lib.bdrt_build_A(ptr(f), 81, ptr(tau), 161, eps, 0, part, 1, 0, 0.0, 0, ptr(out_g)) == 0
        assert np.max(np.abs(out_t - out_g)) <= 1e-12 * np.max(np.abs(out_t))


def test_bad_arguments():
    from bayes_drt_amd import matrices as gm
    f = np.logspace(3, 0, 7)
    with pytest.raises(ValueError):
        gm.construct_A(f, 'real', kernel='XYZ')
    with pytest.raises(ValueError):
        gm.construct_A(f, 'real', kernel='DDT', bc='transmissive', symmetry='spherical', dist_type='parallel')
    with pytest.raises(ValueError):
        gm.construct_A(f, 'real', kernel='DDT', bc='blocking', symmetry='planar', dist_type='parallel', ct=True)
    with pytest.raises(ValueError):
        gm.construct_A(f, 'real', basis='lorentzian')
    with pytest.raises(ValueError):
        gm.construct_L(f, basis='Cole-Cole')              # the reference defines no derivative for it (matrices.py:316-318)
    with pytest.raises(ValueError):
        gm.construct_L(f, basis='Zic', order=1)


@pytest.mark.parametrize('basis', ['Cole-Cole', 'Zic'])
def test_other_basis_functions_vs_golden(basis):
    """the Cole-Cole and Zic basis functions of get_basis_func (matrices.py:14-21) through construct_A on
    the GPU, against the reference's own outputs: Toeplitz path, collocated default tau, general path, a DDT kernel."""
    from bayes_drt_amd import matrices as gm
    d = load('basis_functions')
    tag = basis.replace('-', '')
    eps = float(d['eps_' + tag])
    for part in ('real', 'imag'):
        for got, key in ((gm.construct_A(d['f_lu'], part, tau=d['tau_sup'], epsilon=eps, basis=basis), 'toep'),
                         (gm.construct_A(d['f_lu'], part, epsilon=eps, basis=basis), 'coll'),
                         (gm.construct_A(d['f_ir'], part, tau=d['tau_ir'], epsilon=eps, basis=basis), 'gen'),
                         (gm.construct_A(d['f_ir'], part, tau=d['tau_ir'], epsilon=eps, basis=basis, kernel='DDT',
                                         dist_type='parallel', symmetry='planar', bc='transmissive'), 'ddt')):
            want = d['A_%s_%s_%s' % (tag, key, part)]
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want)), (basis, part, key)


def test_construct_L_non_collocated_and_zic_vs_golden():
    """construct_L for any frequencies against any tau (matrices.py:268-325): [Nf x K], all orders; Zic order 0."""
    from bayes_drt_amd import matrices as gm
    d = load('basis_functions')
    for order, tag in ((0, '0'), (1, '1'), (2, '2'), (3, '3'), (0.5, 'h'), (1.25, 'q'), ([0.2, 0.5, 0.3], 'mix')):
        got = gm.construct_L(d['f_ir'], tau=d['tau_sup'], epsilon=2.5, order=order)
        want = d['L_rect_' + tag]
        assert got.shape == want.shape == (9, 33)
        assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want)), tag
    got = gm.construct_L(d['f_lu'], tau=d['tau_sup'], basis='Zic', epsilon=1.0, order=0)
    assert np.max(np.abs(got - d['L_Zic_0'])) <= 1e-11 * np.max(np.abs(d['L_Zic_0']))
