"""GPU tests (-m gpu) of the one-chain-per-WAVE sampler (bdrt_wave.h): a chain is a workgroup of one wavefront, theta / momentum /
gradient / metric in registers, Toeplitz A, no workgroup barrier.  Same model code and transition logic as the other kernels
(reference call site bayes_drt/inversion.py:1218-1221), another thread mapping: results agree up to summation order."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu


def _problem(tag, mode='sample', pos=True):
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
              induc_scale=float(d['induc_scale']))
    return Problem([blk], d['Z'], d['freq'], **kw), orc.OracleModel([blk], d['Z'], d['freq'], **kw)


def _wave_logp_grad(prob, theta, jac):
    lib = prob._lib
    fn = lib.bdrt_debug_wave_logp_grad
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    lp = np.empty(len(theta)); g = np.empty_like(theta)
    rc = fn(prob.handle, theta.ctypes.data, None, len(theta), int(jac), lp.ctypes.data, g.ctypes.data)
    assert rc == 0, lib.bdrt_last_error().decode()
    return lp, g


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('tag', ['K161', 'K81', 'K101'])
@pytest.mark.parametrize('mode,pos', [('sample', True), ('optimize', True), ('sample', False)])
def test_wave_evaluator_matches_oracle(tag, mode, pos):
    prob, om = _problem(tag, mode, pos)
    rng = np.random.default_rng(3)
    theta = rng.uniform(-2, 2, (70, prob.D))
    jac = mode == 'sample'
    lp, g = _wave_logp_grad(prob, theta, jac)
    lp16, g16 = prob.logp_grad(theta, jacobian=jac)
    for i in range(0, len(theta), 7):
        lp_ref, g_ref = om.logp_grad(theta[i], jac)
        assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
        assert np.max(np.abs(g[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), i
    assert np.allclose(lp, lp16, rtol=1e-11, atol=1e-9) and np.allclose(g, g16, rtol=1e-9, atol=1e-9)
    prob.close()


def _ctrl(lib, **kw):
    from bayes_drt_amd._lib import NutsControl
    c = NutsControl(); lib.bdrt_nuts_defaults(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize('tag', ['K161', 'K81'])
def test_wave_nuts_matches_oracle_and_the_other_kernels(tag):
    from bayes_drt_amd.engine import Sampler, sample_units
    from oracle import oracle as orc
    prob, om = _problem(tag)
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    warm, nd = 25, 8                       # short on purpose: the dynamics amplify the 1e-13 evaluation noise
    with _env(BDRT_WAVE='1', BDRT_SOLO=None):
        with Sampler(prob, 4, warm, nd, 1234, ctrl) as smp:
            assert smp.kind() == 3
            smp.run()
            draws, lp, diag = smp.results()
    with _env(BDRT_WAVE='0', BDRT_SOLO='1'):
        d1, lp1, dg1 = sample_units(prob, 4, warm, nd, 1234, ctrl)
    with _env(BDRT_WAVE='0', BDRT_SOLO='0'):
        d16, lp16, dg16 = sample_units(prob, 4, warm, nd, 1234, ctrl)
    octrl = orc.nuts_control(max_treedepth=6)
    for c in range(4):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, warm, nd, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'] == dg16[c]['n_leapfrog'] == dg1[c]['n_leapfrog'], (c, dr, diag[c], dg16[c])
        assert dr['n_divergent'] == diag[c]['n_divergent']
        assert abs(dr['stepsize'] - diag[c]['stepsize']) < 1e-6 * dr['stepsize']
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
        assert np.max(np.abs(draws[c] - d16[c])) < 1e-6 * np.max(np.abs(ref)), c
        assert np.max(np.abs(draws[c] - d1[c])) < 1e-6 * np.max(np.abs(ref)), c
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)
    prob.close()


@pytest.mark.parametrize('tag', ['K161', 'K81', 'K101'])
def test_wave_schedules_for_one_and_two_waves_per_simd_give_the_same_bits(tag):
    """A launch with at most four chains per CU takes the instantiations scheduled for ONE wave per SIMD (operands of the Toeplitz
    products requested an iteration ahead, band coefficients from LDS, branch-free neighbour terms: bdrt_wave.h, OCC = 1), a fuller one
    the default ones; a chain changes schedule between launches as the other chains finish.  Same operations in the same order:
    evaluator and sampler agree bit for bit (BDRT_WAVE_OCC forces a schedule), and both match the oracle."""
    from bayes_drt_amd.engine import sample_units
    prob, om = _problem(tag)
    rng = np.random.default_rng(11)
    theta = rng.uniform(-2, 2, (40, prob.D))
    out = {}
    for occ in ('1', '2'):
        with _env(BDRT_WAVE_OCC=occ):
            out[occ] = _wave_logp_grad(prob, theta, True)
    assert np.array_equal(out['1'][0], out['2'][0]) and np.array_equal(out['1'][1], out['2'][1])
    lp_ref, g_ref = om.logp_grad(theta[5], True)
    assert abs(out['1'][0][5] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref))
    assert np.max(np.abs(out['1'][1][5] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref)))
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    runs = {}
    for occ in ('1', '2'):
        with _env(BDRT_WAVE='1', BDRT_WAVE_OCC=occ):
            runs[occ] = sample_units(prob, 5, 30, 10, 21, ctrl)
    assert np.array_equal(runs['1'][0], runs['2'][0]) and np.array_equal(runs['1'][1], runs['2'][1])
    assert [d['n_leapfrog'] for d in runs['1'][2]] == [d['n_leapfrog'] for d in runs['2'][2]]
    prob.close()


@pytest.mark.parametrize('hot', [None, '0', '3'])
def test_wave_is_independent_of_launch_slicing_packing_and_resident_rows(hot):
    """The chain state survives the register / LDS <-> HBM round trip at every launch boundary bit for bit, whatever the number
    of LDS-resident rows and of chains per CU."""
    from bayes_drt_amd.engine import sample_units
    prob, om = _problem('K161')
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    ids = np.array([5, 0, 3], dtype=np.int32)
    with _env(BDRT_WAVE='1', BDRT_WAVE_HOT=None, BDRT_WAVE_PER_CU=None):
        one, lp1, dg1 = sample_units(prob, 3, 30, 10, 7, ctrl, chain_ids=ids)
    with _env(BDRT_WAVE='1', BDRT_WAVE_HOT=hot, BDRT_WAVE_PER_CU='8' if hot else None):
        cut, lp2, dg2 = sample_units(prob, 3, 30, 10, 7, ctrl, chain_ids=ids, rounds_per_launch=7)
        solo1, _, _ = sample_units(prob, 1, 30, 10, 7, ctrl, chain_ids=ids[1:2])
    assert np.array_equal(one, cut) and np.array_equal(lp1, lp2)
    assert np.array_equal(solo1[0], one[1])
    with _env(BDRT_WAVE='0', BDRT_SOLO='1'):
        other, _, dg3 = sample_units(prob, 3, 30, 10, 7, ctrl, chain_ids=ids)
    assert not np.array_equal(other, one) and np.allclose(other, one, rtol=0, atol=1e-4 * np.max(np.abs(one)))
    assert [d['n_leapfrog'] for d in dg1] == [d['n_leapfrog'] for d in dg3]
    prob.close()


def test_wave_many_units_full_depth_against_the_16_chain_kernel():
    """More chains than one turn of the machine holds (9 per CU asked of a kernel that runs 8), full tree depth, several spectra:
    every chain equals the 16-chain kernel's to the usual tolerance of a short run and the leapfrog counts agree exactly."""
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.model import Problem
    d = load('dat_sample_2ZARC_uniform_0.25_K161')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    rng = np.random.default_rng(5)
    Z = d['Z'][None, :] * (1.0 + 0.01 * rng.standard_normal((3, len(d['Z']))))
    prob = Problem([blk], Z, d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']))
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    n = 2400
    spec = (np.arange(n) % 3).astype(np.int32)
    with _env(BDRT_WAVE='1'):
        a, lpa, da = sample_units(prob, n, 12, 3, 99, ctrl, spec=spec)
    with _env(BDRT_WAVE='0', BDRT_SOLO='0'):
        b, lpb, db = sample_units(prob, n, 12, 3, 99, ctrl, spec=spec)
    same = [x['n_leapfrog'] == y['n_leapfrog'] for x, y in zip(da, db)]
    assert np.mean(same) > 0.995, np.mean(same)            # (a chain whose U-turn test sits at rounding level may differ)
    ok = np.array(same)
    scale = np.max(np.abs(b))
    assert np.max(np.abs(a[ok] - b[ok])) < 1e-5 * scale
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(lpa))
    prob.close()


def test_wave_long_run_statistics_match_the_16_chain_kernel():
    """Posterior means / spreads of a longer run (through all metric-adaptation windows) agree between the two kernels
    within Monte-Carlo error: 4 chains x (300 + 300), K = 81."""
    from bayes_drt_amd.engine import sample_units
    prob, om = _problem('K81')
    ctrl = _ctrl(prob._lib)
    with _env(BDRT_WAVE='1'):
        a, _, da = sample_units(prob, 4, 300, 300, 11, ctrl)
    with _env(BDRT_WAVE='0', BDRT_SOLO='0'):
        b, _, db = sample_units(prob, 4, 300, 300, 12, ctrl)
    A, B = a.reshape(-1, prob.D), b.reshape(-1, prob.D)
    sd = 0.5 * (A.std(axis=0) + B.std(axis=0))
    z = np.abs(A.mean(axis=0) - B.mean(axis=0)) / (sd / np.sqrt(60.0))      # ~60 effective draws per run, conservatively
    assert np.max(z) < 6.0, (np.argmax(z), np.max(z))
    r = np.abs(A.std(axis=0) / B.std(axis=0) - 1)          # heavy-tailed coordinates (d strengths, ups of empty regions) are noisy
    assert np.median(r) < 0.15 and np.quantile(r, 0.9) < 0.5 and np.max(r) < 3.0, (np.median(r), np.quantile(r, 0.9), np.max(r))
    assert sum(d['n_divergent'] for d in da) < 10 and np.mean([d['mean_accept'] for d in da]) > 0.7
    prob.close()


def _outlier_problem(tag, om_mode):
    """Series_pos with the outlier error model (two further parameters per frequency; bdrt_tile_hw.h): mode 1 = the Stan files' form
    (sigma_out = raw x scale), mode 2 = one parameter per part."""
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    d = load('dat_sample_2ZARC_uniform_0.25_%s' % tag)
    so = load('dat_sample_outlier_scalars')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
              induc_scale=float(d['induc_scale']), outlier_mode=om_mode, so_lambda=float(so['sigma_out_lambda']),
              so_alpha=float(so['sigma_out_alpha']), so_beta=float(so['sigma_out_beta']))
    return Problem([blk], d['Z'], d['freq'], **kw), orc.OracleModel([blk], d['Z'], d['freq'], **kw)


@pytest.mark.parametrize('tag', ['K161', 'K81'])
@pytest.mark.parametrize('om_mode', [1, 2])
def test_wave_evaluator_with_the_outlier_model_matches_oracle(tag, om_mode):
    prob, om = _outlier_problem(tag, om_mode)
    rng = np.random.default_rng(5)
    theta = rng.uniform(-2, 2, (40, prob.D))
    for jac in (True, False):
        lp, g = _wave_logp_grad(prob, theta, jac)
        lp16, g16 = prob.logp_grad(theta, jacobian=jac)
        for i in range(0, len(theta), 5):
            lp_ref, g_ref = om.logp_grad(theta[i], jac)
            assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
            assert np.max(np.abs(g[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), i
        assert np.allclose(lp, lp16, rtol=1e-11, atol=1e-9) and np.allclose(g, g16, rtol=1e-9, atol=1e-9)
    prob.close()


@pytest.mark.parametrize('om_mode', [1, 2])
def test_wave_nuts_with_the_outlier_model_matches_oracle_and_the_16_chain_kernel(om_mode):
    """Mid-occupancy runs of the outlier models (what a batch of a few hundred spectra x 4 chains is) take the one-chain-per-wave kernel
    with the outlier parameters as further slots of the lanes, four chains per CU: draw by draw the oracle's chains."""
    from bayes_drt_amd.engine import Sampler, sample_units
    from oracle import oracle as orc
    prob, om = _outlier_problem('K161', om_mode)
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    warm, nd = 20, 6
    with _env(BDRT_WAVE='1', BDRT_SOLO=None, BDRT_WIDE1=None):
        with Sampler(prob, 5, warm, nd, 4321, ctrl) as smp:
            assert smp.kind() == 3
            smp.run()
            draws, lp, diag = smp.results()
    with _env(BDRT_WAVE=None, BDRT_SOLO=None, BDRT_WIDE1=None):
        # the default choice for this family: the general one-chain kernel up to one chain per CU, the wave kernel from there to four
        with Sampler(prob, 5, 2, 2, 4321, ctrl) as few:
            assert few.kind() == 2
        with Sampler(prob, 300, 2, 2, 4321, ctrl) as mid:
            assert mid.kind() == 3
    with _env(BDRT_WAVE='0', BDRT_SOLO='0', BDRT_WIDE1='0'):
        d16, lp16, dg16 = sample_units(prob, 5, warm, nd, 4321, ctrl)
    octrl = orc.nuts_control(max_treedepth=6)
    for c in range(5):
        ref, lpr, dr = orc.nuts_sample(om, c, 4321, warm, nd, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'] == dg16[c]['n_leapfrog'], (c, dr, diag[c], dg16[c])
        assert dr['n_divergent'] == diag[c]['n_divergent']
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
        assert np.max(np.abs(draws[c] - d16[c])) < 1e-6 * np.max(np.abs(ref)), c
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)
    prob.close()


def test_tail_of_a_large_run_of_an_outlier_model_moves_to_the_wave_kernel(monkeypatch):
    """More than four chains per CU of the outlier family start on the 16-chain kernel; `bdrt_sampler_run` hands the last live chains
    (<= four per CU) to the one-chain-per-wave kernel.  The run equals the one without the hand-over chain by chain."""
    from bayes_drt_amd.engine import Sampler
    prob, om = _outlier_problem('K81', 1)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    n_units = 1200

    def run():
        with Sampler(prob, n_units, 16, 8, 3, ctrl) as smp:
            kind0 = smp.kind()
            smp.run()
            return smp.results() + (kind0, smp.kind(), smp.tail_units())

    d1, lp1, g1, k0, k1, tail1 = run()
    assert k0 == 0 and k1 == 3 and 0 < tail1 <= 1024, (k0, k1, tail1)
    monkeypatch.setenv('BDRT_TAIL_MIGRATION', '0')
    d0, lp0, g0, k0b, k1b, tail0 = run()
    assert k1b == 0 and tail0 == 0
    assert np.all(np.isfinite(d1))
    err = np.max(np.abs(d1 - d0), axis=(1, 2)) / np.max(np.abs(d0))
    assert np.mean(err < 1e-6) > 0.9, np.mean(err < 1e-6)
    assert [x['n_leapfrog'] for x in g1[:50]] == [x['n_leapfrog'] for x in g0[:50]]
    prob.close()


# ---- several distributions (wave_eval_nb, bdrt_wave_nb.hip) ---------------------------------------------------------------------------------
MULTI = ['series_parallel', 'series_parallel_outliers', 'kat_2parallel']        # (two blocks, two blocks + outlier model = BASELINE config 5's, three blocks)


def _multi(name):
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    from tests.test_gpu_solo_wide import _family
    args = _family(name)
    return Problem(**args), orc.OracleModel(**args)


@pytest.mark.parametrize('name', MULTI)
def test_wave_evaluator_of_the_multi_distribution_models_matches_oracle(name):
    """Series-Parallel (two blocks, +- the outlier model: BASELINE config 5's model) and Series-2Parallel (three blocks) on the lane
    mapping of the one-chain-per-wave kernel: lp and gradient against the oracle and the tile evaluator."""
    prob, om = _multi(name)
    rng = np.random.default_rng(9)
    theta = rng.uniform(-2, 2, (24, prob.D))
    for jac in (True, False):
        lp, g = _wave_logp_grad(prob, theta, jac)
        lp16, g16 = prob.logp_grad(theta, jacobian=jac)
        for i in range(0, len(theta), 4):
            lp_ref, g_ref = om.logp_grad(theta[i], jac)
            if not np.isfinite(lp_ref):
                assert not np.isfinite(lp[i])                # (x_sum_raw < 0: rejected point)
                continue
            assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
            assert np.max(np.abs(g[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref))), (i, np.max(np.abs(g[i] - g_ref)))
        fin = np.isfinite(lp16)
        assert np.array_equal(fin, np.isfinite(lp))
        assert np.allclose(lp[fin], lp16[fin], rtol=1e-11, atol=1e-9) and np.allclose(g[fin], g16[fin], rtol=1e-9, atol=1e-9)
    prob.close()


@pytest.mark.parametrize('name', ['series_parallel', 'series_parallel_outliers', 'kat_2parallel'])
def test_wave_nuts_of_the_multi_distribution_models_matches_oracle_and_the_16_chain_kernel(name):
    from bayes_drt_amd.engine import Sampler, sample_units
    from oracle import oracle as orc
    prob, om = _multi(name)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    warm, nd = 12, 5
    with _env(BDRT_WAVE='1', BDRT_SOLO=None, BDRT_WIDE1=None):
        with Sampler(prob, 3, warm, nd, 77, ctrl) as smp:
            assert smp.kind() == 3
            smp.run()
            draws, lp, diag = smp.results()
    with _env(BDRT_WAVE='0', BDRT_SOLO='0', BDRT_WIDE1='0'):
        d16, lp16, dg16 = sample_units(prob, 3, warm, nd, 77, ctrl)
    octrl = orc.nuts_control(max_treedepth=5)
    for c in range(3):
        ref, lpr, dr = orc.nuts_sample(om, c, 77, warm, nd, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'] == dg16[c]['n_leapfrog'], (c, dr, diag[c], dg16[c])
        assert dr['n_divergent'] == diag[c]['n_divergent']
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
        assert np.max(np.abs(draws[c] - d16[c])) < 1e-6 * np.max(np.abs(ref)), c
    with _env(BDRT_WAVE=None, BDRT_SOLO=None, BDRT_WIDE1=None):
        with Sampler(prob, 3, 2, 2, 77, ctrl) as few:
            assert few.kind() == 2                           # up to one chain per CU: the general one-chain kernel
        with Sampler(prob, 400, 2, 2, 77, ctrl) as mid:
            assert mid.kind() == 3                           # from there to four per CU: one chain per wave
    prob.close()
