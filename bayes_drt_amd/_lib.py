"""ctypes binding of libbdrt.so (C ABI: include/bdrt.h).

This is the binding INTEGRATION.md shows for the reference side: plain pointers and sizes, no torch types.
The library is loaded lazily and loudly: a missing libbdrt.so or a missing GPU is an error, never a fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAXB = 3


class BdrtError(RuntimeError):
    pass


class Dat(C.Structure):
    """bdrt_dat (include/bdrt.h) = the Stan data block assembled by Inverter._prep_stan_data."""
    _fields_ = [('nf', C.c_int), ('nblocks', C.c_int),
                ('K', C.c_int * MAXB), ('is_parallel', C.c_int * MAXB), ('nonneg', C.c_int * MAXB),
                ('x_scale', C.c_double * MAXB),
                ('A', C.c_void_p * MAXB), ('L0', C.c_void_p * MAXB), ('L1', C.c_void_p * MAXB),
                ('L2', C.c_void_p * MAXB),
                ('freq', C.c_void_p), ('n_spectra', C.c_int), ('Z', C.c_void_p),
                ('sigma_min', C.c_double), ('ups_alpha', C.c_double), ('ups_beta', C.c_double),
                ('induc_scale', C.c_double),
                ('outlier_mode', C.c_int),
                ('so_lambda', C.c_double), ('so_alpha', C.c_double), ('so_beta', C.c_double),
                ('use_x_sum', C.c_int), ('x_sum_invscale', C.c_double)]


class OptOptions(C.Structure):
    _fields_ = [('max_iter', C.c_int), ('history', C.c_int), ('init_alpha', C.c_double), ('tol_obj', C.c_double),
                ('tol_rel_obj', C.c_double), ('tol_grad', C.c_double), ('tol_rel_grad', C.c_double),
                ('tol_param', C.c_double), ('newton_max_iter', C.c_int), ('lbfgs_before_newton', C.c_int),
                ('newton_tol', C.c_double)]


class OptReport(C.Structure):
    _fields_ = [('iterations', C.c_int), ('n_evals', C.c_int), ('return_code', C.c_int), ('lp', C.c_double),
                ('grad_norm', C.c_double), ('newton_iterations', C.c_int), ('grad_inf', C.c_double)]


class RidgeOptions(C.Structure):
    _fields_ = [('n', C.c_int), ('K', C.c_int), ('off', C.c_int), ('penalty', C.c_int), ('max_iter', C.c_int),
                ('hyper_lambda', C.c_int), ('zero_delta1', C.c_int), ('xtol', C.c_double), ('hl_fbeta', C.c_double),
                ('reg_ord', C.c_double * 3)]


class NutsControl(C.Structure):
    _fields_ = [('adapt_delta', C.c_double), ('adapt_t0', C.c_double), ('adapt_gamma', C.c_double),
                ('adapt_kappa', C.c_double), ('max_treedepth', C.c_int), ('init_buffer', C.c_int),
                ('term_buffer', C.c_int), ('base_window', C.c_int), ('init_radius', C.c_double),
                ('max_deltaH', C.c_double), ('stepsize0', C.c_double)]


class ChainDiag(C.Structure):
    _fields_ = [('n_leapfrog', C.c_int64), ('n_divergent', C.c_int), ('n_max_treedepth', C.c_int),
                ('stepsize', C.c_double), ('mean_accept', C.c_double)]


# every symbol include/bdrt.h declares (tests/test_abi.py checks the library exports each of them)
SYMBOLS = [
    'bdrt_build_A', 'bdrt_build_A_basis', 'bdrt_build_L', 'bdrt_build_L_rect', 'bdrt_build_M',
    'bdrt_problem_create', 'bdrt_problem_destroy', 'bdrt_num_params', 'bdrt_problem_evaluator', 'bdrt_param_is_pos', 'bdrt_problem_set_Z',
    'bdrt_logp_grad', 'bdrt_logp_grad_dev', 'bdrt_transformed',
    'bdrt_opt_defaults', 'bdrt_optimize',
    'bdrt_nuts_defaults', 'bdrt_sampler_create', 'bdrt_sampler_destroy', 'bdrt_sampler_advance', 'bdrt_sampler_sync',
    'bdrt_sampler_run', 'bdrt_sampler_results', 'bdrt_sampler_tail_units', 'bdrt_sampler_compactions', 'bdrt_sampler_kind', 'bdrt_sampler_total_leapfrogs', 'bdrt_sampler_kernel_time',
    'bdrt_sampler_phase_profile',
    'bdrt_sample',
    'bdrt_gram', 'bdrt_qp_box', 'bdrt_qp_box_batch', 'bdrt_ridge',
    'bdrt_percentiles', 'bdrt_sampler_percentiles', 'bdrt_sampler_summary', 'bdrt_summary', 'bdrt_sampler_draws_dev',
    'bdrt_last_error', 'bdrt_device_count', 'bdrt_set_device', 'bdrt_version',
]


def library_path():
    # BDRT_LIBRARY: another build of the same library (A/B measurements of kernel variants, tools/build_variant.sh)
    return os.environ.get('BDRT_LIBRARY') or os.path.join(_HERE, 'libbdrt.so')


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; libbdrt.so is linked
    against the system ROCm.  Loaded in the order "libbdrt first, torch later" the process ends up with BOTH runtimes and the
    second one finds no GPU ("No HIP GPUs are available" from torch).  In the order "torch first" libbdrt's dependency resolves
    to the runtime torch brought (same SONAME) -- the configuration bench.py, the tests and parallel.py run in.  So when
    PyTorch is installed, its runtime is loaded before libbdrt.so whatever the caller imports first (without importing torch).
    BDRT_HIP_RUNTIME=system skips this (a process that will never import torch)."""
    import importlib.util
    import sys
    if 'torch' in sys.modules or os.environ.get('BDRT_HIP_RUNTIME') == 'system':
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                   # fall back to the system runtime libbdrt.so was linked against


def load_library():
    """Load libbdrt.so.  Raises BdrtError when it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise BdrtError('libbdrt.so not found at %s: build it with `python -c "import __graft_entry__ as g; '
                        'g.build()"` or `make -C bayes_drt_amd/csrc`' % path)
    _preload_hip_runtime()
    lib = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p
    lib.bdrt_last_error.restype = C.c_char_p
    lib.bdrt_version.restype = C.c_char_p
    lib.bdrt_problem_create.restype = vp
    lib.bdrt_problem_create.argtypes = [C.POINTER(Dat)]
    lib.bdrt_problem_destroy.argtypes = [vp]
    lib.bdrt_problem_destroy.restype = None
    lib.bdrt_num_params.argtypes = [vp]
    lib.bdrt_problem_evaluator.argtypes = [vp]
    lib.bdrt_problem_evaluator.restype = C.c_int
    lib.bdrt_param_is_pos.argtypes = [vp, vp]
    lib.bdrt_problem_set_Z.argtypes = [vp, vp, C.c_int]
    lib.bdrt_logp_grad.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.bdrt_logp_grad_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.bdrt_transformed.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    lib.bdrt_build_A.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_int, vp]
    lib.bdrt_build_A_basis.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_int, C.c_int, vp]
    lib.bdrt_build_L.argtypes = [vp, C.c_int, C.c_double, vp, vp]
    lib.bdrt_build_L_rect.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, vp, C.c_int, vp]
    lib.bdrt_build_M.argtypes = [vp, C.c_int, C.c_double, vp, C.c_int, vp]
    lib.bdrt_opt_defaults.argtypes = [C.POINTER(OptOptions)]
    lib.bdrt_opt_defaults.restype = None
    lib.bdrt_optimize.argtypes = [vp, vp, vp, C.c_int, C.POINTER(OptOptions), vp, vp]
    lib.bdrt_nuts_defaults.argtypes = [C.POINTER(NutsControl)]
    lib.bdrt_nuts_defaults.restype = None
    lib.bdrt_sampler_create.restype = vp
    lib.bdrt_sampler_create.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_uint64, vp, C.POINTER(NutsControl)]
    lib.bdrt_sampler_destroy.argtypes = [vp]
    lib.bdrt_sampler_destroy.restype = None
    lib.bdrt_sampler_advance.argtypes = [vp, C.c_int, vp]
    lib.bdrt_sampler_sync.argtypes = [vp]
    lib.bdrt_sampler_run.argtypes = [vp]
    lib.bdrt_sampler_results.argtypes = [vp, vp, vp, vp]
    lib.bdrt_sampler_tail_units.argtypes = [vp]
    lib.bdrt_sampler_compactions.argtypes = [vp]
    lib.bdrt_sampler_compactions.restype = C.c_int
    lib.bdrt_sampler_kind.argtypes = [vp]
    lib.bdrt_sampler_kind.restype = C.c_int
    lib.bdrt_sampler_tail_units.restype = C.c_int
    lib.bdrt_sampler_total_leapfrogs.argtypes = [vp]
    lib.bdrt_sampler_total_leapfrogs.restype = C.c_int64
    lib.bdrt_sampler_kernel_time.argtypes = [vp, vp, vp, C.c_int]
    lib.bdrt_sampler_phase_profile.argtypes = [vp, C.c_int, vp]
    lib.bdrt_sample.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_uint64, vp, C.POINTER(NutsControl), vp, vp,
                                vp]
    lib.bdrt_gram.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.bdrt_qp_box.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    lib.bdrt_qp_box_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.bdrt_ridge.argtypes = [C.POINTER(RidgeOptions), C.c_int, C.c_int] + [vp] * 20
    lib.bdrt_percentiles.argtypes = [vp, C.c_int, C.c_int, C.c_long, vp, C.c_int, vp, vp, C.c_int, vp]
    lib.bdrt_sampler_percentiles.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
    lib.bdrt_sampler_summary.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    lib.bdrt_summary.argtypes = [vp, C.c_int, C.c_int, C.c_long, vp, vp, C.c_int, vp, vp]
    lib.bdrt_sampler_draws_dev.argtypes = [vp]
    lib.bdrt_sampler_draws_dev.restype = vp
    lib.bdrt_set_device.argtypes = [C.c_int]
    _LIB = lib
    return lib


def check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise BdrtError('%s failed (%s): %s' % (what, rc, load_library().bdrt_last_error().decode()))
    return rc


def require_gpu():
    lib = load_library()
    if lib.bdrt_device_count() < 1:
        raise BdrtError('no HIP device visible: bayes_drt_amd has no CPU fallback')
    return lib


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
