"""GPU tests (-m gpu) of the general one-chain-per-workgroup path (bdrt_solo_wide.h): several distributions, parallel blocks,
the outlier error model -- the reference's own call shape (2-4 chains) for BASELINE config 5."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu


def _family(name):
    from bayes_drt_amd.engine import blocks_from_dat
    if name == 'series_outliers':
        from tests.test_gpu_engine import _bench_problem
        blk, Z, f, kw, d = _bench_problem('sample', 'K161')
        so = load('dat_sample_outlier_scalars')
        kw = dict(kw, outlier_mode=1, so_lambda=float(so['sigma_out_lambda']), so_alpha=float(so['sigma_out_alpha']),
                  so_beta=float(so['sigma_out_beta']))
        return dict(blocks=[blk], Z=Z, freq=f, **kw)
    if name == 'series_plain':
        from tests.test_gpu_engine import _bench_problem
        blk, Z, f, kw, d = _bench_problem('sample', 'K161')
        return dict(blocks=[blk], Z=Z, freq=f, **kw)
    if name == 'series_irregular_frequencies':
        # a single DRT whose measurement frequencies are not log-uniform (an instrument's rounded list): A is a general
        # matrix (construct_A's quadrature path), L stays banded Toeplitz
        from bayes_drt_amd import matrices as gm
        rng = np.random.default_rng(11)
        nf, K = 67, 121
        f = np.sort(np.logspace(5.5, -1.5, nf) * np.exp(0.08 * rng.standard_normal(nf)))[::-1]
        bf = np.logspace(7.5, -3.5, K)
        tau = 1 / (2 * np.pi * bf)
        eps = 1 / np.mean(np.diff(np.log(tau)))
        A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
        L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
        x = np.exp(-0.5 * ((np.log(tau) + 5) / 1.2) ** 2)
        Z = A @ x + np.concatenate([np.full(nf, 0.4), np.zeros(nf)])
        Z = Z / np.std(np.hypot(Z[:nf], Z[nf:])) + 0.01 * rng.standard_normal(2 * nf)
        return dict(blocks=[dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], Z=Z, freq=f, sigma_min=0.002,
                    ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)
    if name in ('kat_2parallel', 'kat_series_parallel_outliers'):
        # stored Stan fits of the reference: three blocks (series + two parallel) / two blocks with the stacked outlier model
        from tests.helpers import kat_to_model
        return kat_to_model({'kat_2parallel': 'DRT-TpDDT-BpDDT_uniform_0.25',
                             'kat_series_parallel_outliers': 'PDAC_DRT-TpDDT_outliers'}[name])['kw']
    dd = load('dat_sample_DRT-TpDDT_plain')
    blocks, kw2, _ = blocks_from_dat('Series-Parallel_pos_StanModel.pkl', {k: dd[k] for k in dd.files})
    if name == 'series_parallel_outliers':               # BASELINE config 5's model: the plain matrices + the outlier scalars
        do = load('dat_sample_DRT-TpDDT_outliers')
        kw2 = dict(kw2, outlier_mode=2, so_lambda=float(do['so_invscale']))
    return dict(blocks=blocks, Z=dd['Z'], freq=dd['freq'], **kw2)


def _wide1_logp_grad(prob, theta, jac):
    lib = prob._lib
    fn = lib.bdrt_debug_wide1_logp_grad
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    lp = np.empty(len(theta)); g = np.empty_like(theta)
    rc = fn(prob.handle, theta.ctypes.data, None, len(theta), int(jac), lp.ctypes.data, g.ctypes.data)
    if rc == -2:
        return None, None                                  # the problem does not take this evaluator
    assert rc == 0, lib.bdrt_last_error().decode()
    return lp, g


@pytest.mark.parametrize('family', ['series_plain', 'series_outliers', 'series_parallel', 'series_parallel_outliers',
                                    'kat_2parallel', 'kat_series_parallel_outliers', 'series_irregular_frequencies'])
@pytest.mark.parametrize('jac', [True, False])
def test_one_chain_evaluator_matches_the_batched_evaluator_and_the_oracle(family, jac):
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    args = _family(family)
    prob = Problem(**args)
    om = orc.OracleModel(**args)
    rng = np.random.default_rng(5)
    theta = rng.uniform(-2, 2, (5, prob.D))
    lp, g = _wide1_logp_grad(prob, theta, jac)
    # ('kat_series_parallel_outliers': a measured spectrum on its own frequency grid, 106 points, K = 101 -- A is not Toeplitz
    #  and the evaluator reads plain copies of it instead of generators)
    assert lp is not None
    lp16, g16 = prob.logp_grad(theta, jacobian=jac)
    for i in range(len(theta)):
        lp_ref, g_ref = om.logp_grad(theta[i], jac)
        if not np.isfinite(lp_ref):
            assert lp[i] == lp_ref
            continue
        assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
        assert np.max(np.abs(g[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), (i, np.argmax(np.abs(g[i] - g_ref)))
    fin = np.isfinite(lp16)
    assert np.allclose(lp[fin], lp16[fin], rtol=1e-11, atol=1e-9) and np.allclose(g[fin], g16[fin], rtol=1e-9, atol=1e-9)
    prob.close()


@pytest.mark.parametrize('family', ['series_parallel_outliers', 'kat_2parallel', 'series_outliers', 'series_irregular_frequencies',
                                    'kat_series_parallel_outliers'])
def test_few_chains_take_the_one_chain_kernel_and_equal_the_16_chain_kernel(family, monkeypatch):
    """Four chains of a multi-distribution / outlier model run on the kernel of bdrt_solo_wide.h: bit-identical across launch
    slicing, and the same chains as on the 16-chain kernel (BDRT_WIDE1=0) up to summation order."""
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd._lib import NutsControl
    prob = Problem(**_family(family))
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 6

    def run(slice_rounds):
        with Sampler(prob, 4, 10, 6, 21, ctrl) as smp:
            kind = smp.kind()
            smp.run(slice_rounds)
            return smp.results() + (kind,)

    d_a, lp_a, g_a, kind_a = run(None)
    d_b, lp_b, g_b, kind_b = run(13)
    assert kind_a == 2 and kind_b == 2
    assert np.array_equal(d_a, d_b) and np.array_equal(lp_a, lp_b)
    monkeypatch.setenv('BDRT_WIDE1', '0')
    d_c, lp_c, g_c, kind_c = run(None)
    assert kind_c == 0
    assert [x['n_leapfrog'] for x in g_a] == [x['n_leapfrog'] for x in g_c]
    assert np.max(np.abs(d_a - d_c)) <= 1e-6 * np.max(np.abs(d_c))
    prob.close()


@pytest.mark.parametrize('wave', [False, True])
def test_tail_of_a_large_run_of_a_general_model_moves_to_the_one_chain_kernels(wave, monkeypatch):
    """A large run of the config 5 model starts on the 16-chain kernel; `bdrt_sampler_run` hands the last live chains to a one-chain
    kernel: the one-chain-per-wave kernel (<= four chains per CU, round 5) or -- without it, BDRT_WAVE=0 -- the kernel of
    bdrt_solo_wide.h (<= 2.75 per CU).  The run equals the one without the hand-over chain by chain."""
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd._lib import NutsControl
    if not wave:
        monkeypatch.setenv('BDRT_WAVE', '0')
    prob = Problem(**_family('series_parallel_outliers'))
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 5
    n_units = 1200 if wave else 900

    def run():
        with Sampler(prob, n_units, 16, 8, 3, ctrl) as smp:
            kind0 = smp.kind()
            smp.run()
            return smp.results() + (kind0, smp.kind(), smp.tail_units())

    d1, lp1, g1, k0, k1, tail1 = run()
    assert k0 == 0 and k1 == (3 if wave else 2) and 0 < tail1 <= (1024 if wave else 704), (k0, k1, tail1)
    monkeypatch.setenv('BDRT_TAIL_MIGRATION', '0')
    d0, lp0, g0, k0b, k1b, tail0 = run()
    assert k1b == 0 and tail0 == 0
    assert np.all(np.isfinite(d1))
    err = np.max(np.abs(d1 - d0), axis=(1, 2)) / np.max(np.abs(d0))
    # (the chains that were handed over continue in another summation order: the longer the stretch on the other kernel -- the wave
    #  kernel takes over at 1024 live chains, the workgroup kernel at 704 --, the more of them drift past 1e-6 in these 24 iterations)
    assert np.mean(err < 1e-6) > (0.8 if wave else 0.9), np.mean(err < 1e-6)
    assert np.median(err) < 1e-12
    prob.close()
