"""construct_A / construct_L / construct_M with the reference's signatures (bayes_drt/matrices.py:120, :268,
:366), computed on the GPU through libbdrt.so (bdrt_build_A / _L / _M)."""
import numpy as np

from . import _lib
from ._lib import check, f64, ptr
from .utils import is_loguniform, rel_round

_KERNEL_ID = {('blocking', 'planar'): 1, ('blocking', 'spherical'): 2, ('transmissive', 'planar'): 3}
_BASIS_ID = {'gaussian': 0, 'Cole-Cole': 1, 'Zic': 2}            # get_basis_func (reference matrices.py:8-24)


def _basis_id(basis):
    if basis not in _BASIS_ID:
        raise ValueError(f'Invalid basis {basis}. Options are gaussian')
    return _BASIS_ID[basis]


def _kernel_id(kernel, dist_type, symmetry, bc, ct, k_ct):
    if ct is True and k_ct is None:
        raise ValueError('k_ct must be supplied if ct==True')
    if kernel == 'DRT':
        if dist_type != 'series':
            raise ValueError('dist_type for DRT kernel must be series')
        return 0
    if kernel != 'DDT':
        raise ValueError(f'Invalid kernel {kernel}. Options are DRT and DDT')
    if bc == 'blocking' and symmetry not in ('planar', 'spherical'):
        raise ValueError(f'Invalid symmetry {symmetry}. Options are planar or spherical for bc=blocking')
    if bc == 'transmissive' and symmetry != 'planar':
        raise ValueError(f'Invalid symmetry {symmetry}. Symmetry must be planar for bc=transmissive')
    if dist_type not in ('series', 'parallel'):
        raise ValueError(f'Invalid dist_type {dist_type}. Options are series and parallel')
    return _KERNEL_ID[(bc, symmetry)]


def _toeplitz_decision(frequencies, tau, tau_given, ct):
    """The reference decides between the Toeplitz shortcut and the full double loop from the two grids alone
    (matrices.py:147-205): log-uniform frequencies, no charge transfer, and tau == 1/omega or one grid a
    contiguous run of the other with log-uniform tau."""
    omega = frequencies * 2 * np.pi
    r_inv_om = rel_round(1 / omega, 10)
    r_tau = rel_round(tau, 10)
    same = (not tau_given) or (len(tau) == len(omega) and bool(np.min(r_tau == r_inv_om)))
    sub = False
    first = r_tau == rel_round(1 / omega[0], 10)
    if first.sum() > 1:
        raise Exception('Repeated tau values')
    if first.sum() == 1:
        i0 = int(np.argmax(first))
        run = r_tau[i0:i0 + len(omega)]
        sub = len(run) == len(omega) and bool(np.min(run == r_inv_om))
    if not sub:
        first = r_inv_om == r_tau[0]
        if first.sum() > 1:
            raise Exception('Repeated omega values')
        if first.sum() == 1:
            i0 = int(np.argmax(first))
            run = rel_round(omega[i0:i0 + len(tau)], 10)
            sub = len(run) == len(tau) and bool(np.min(run == rel_round(1 / tau, 10)))
    if is_loguniform(frequencies) and not ct:
        return bool(same or (sub and is_loguniform(tau)))
    return False


def construct_A(frequencies, part, tau=None, basis='gaussian', fit_inductance=False, epsilon=1, kernel='DRT',
                dist_type='series', symmetry='planar', bc=None, ct=False, k_ct=None, integrate_method='trapz'):
    """A' / A'' matrix ([len(frequencies) x len(tau)]).  Same arguments as the reference: basis 'gaussian' (what Inverter
    uses), 'Cole-Cole' (0 < epsilon < 1) or 'Zic'; the trapezoid quadrature (the reference's default and the only one any
    caller uses) is the one implemented."""
    bid = _basis_id(basis)
    if integrate_method != 'trapz':
        raise ValueError("only integrate_method='trapz' (the reference default) is implemented")
    if part not in ('real', 'imag'):
        raise ValueError(f"Invalid part {part}. Options are 'real' or 'imag'")
    kid = _kernel_id(kernel, dist_type, symmetry, bc, ct, k_ct)
    f = f64(frequencies)
    tau_given = tau is not None
    t = f64(tau) if tau_given else f64(1 / (f * 2 * np.pi))
    toep = _toeplitz_decision(f, t, tau_given, bool(ct))
    out = np.empty((len(f), len(t)))
    lib = _lib.require_gpu()
    rc = lib.bdrt_build_A_basis(ptr(f), len(f), ptr(t), len(t), float(epsilon), kid, 0 if part == 'real' else 1,
                                int(dist_type == 'series'), int(bool(ct)), float(k_ct) if k_ct is not None else 0.0,
                                int(toep), bid, ptr(out))
    if rc == -2:
        raise Exception('First entries of first row and column are not equal')
    check(rc, 'bdrt_build_A')
    return out


def _order_coefs(order, n):
    c = np.zeros(n)
    if type(order) == list:
        c[:3] = order
    elif order in (0, 1, 2, 3) and order < n:
        c[int(order)] = 1.0
    elif 0 < order < 1:
        c[0], c[1] = 1 - order, order
    elif 1 < order < 2:
        c[1], c[2] = 2 - order, order - 1
    else:
        raise ValueError('Order must be between 0 and 3' if n == 4 else f'Invalid order {order}')
    return c


def construct_L(frequencies, tau=None, basis='gaussian', epsilon=1, order=1):
    """Differentiation matrix [len(frequencies) x len(tau)]: L@coef gives the order-th derivative of the distribution at
    ln tau = -ln(2 pi frequencies) (reference :268-325).  Inverter calls it collocated, frequencies = 1/(2 pi tau)
    (inversion.py:2302-2307); any other pair of grids works the same way.  basis: 'gaussian' (orders 0-3, fractional, 3-list
    mixes) or 'Zic' (order 0, the only case the reference defines, :316-318)."""
    bid = _basis_id(basis)
    if basis == 'Cole-Cole' or (basis == 'Zic' and not (type(order) != list and order == 0)):
        raise ValueError(f'construct_L: no derivative of order {order} is defined for the {basis} basis')
    f = f64(frequencies)
    t = f64(tau) if tau is not None else f64(1 / (2 * np.pi * f))
    out = np.empty((len(f), len(t)))
    lib = _lib.require_gpu()
    check(lib.bdrt_build_L_rect(ptr(f), len(f), ptr(t), len(t), float(epsilon), ptr(_order_coefs(order, 4)), bid, ptr(out)),
          'bdrt_build_L')
    return out


def construct_M(frequencies, basis='gaussian', order=1, epsilon=1):
    """Integrated-penalty matrix: x^T M x = integral of the squared order-th derivative over ln(tau)."""
    if basis != 'gaussian':
        raise ValueError(f'Invalid basis {basis}')
    f = f64(frequencies)
    t = f64(1 / (2 * np.pi * f))
    out = np.empty((len(t), len(t)))
    lib = _lib.require_gpu()
    check(lib.bdrt_build_M(ptr(t), len(t), float(epsilon), ptr(_order_coefs(order, 3)), int(is_loguniform(f)),
                           ptr(out)), 'bdrt_build_M')
    return out
