"""GPU parity (-m gpu): HIP log-posterior + gradient (libbdrt.so via the C ABI) vs the CPU oracle.

fp64 tolerance: the two sides sum the same products in different orders (MFMA k-blocked vs sequential), so
lp agrees to ~1e-13 relative; asserted: |lp - lp_ref| <= 1e-10 * max(1, |lp_ref|) and
max|g - g_ref| <= 1e-10 * max(1, max|g_ref|).
"""
import numpy as np
import pytest

from tests.helpers import kat_names, kat_to_model, load

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_evaluators(monkeypatch):
    """These tests pin the 16-column tile evaluators.  Batches of up to one point per CU would otherwise take the
    one-workgroup-per-point evaluators (launch_logp_grad_few), which have their own tests below and in test_gpu_solo*.py."""
    monkeypatch.setenv('BDRT_FEW_POINTS', '0')


def _mods():
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    return Problem, orc


def _compare(prob, om, thetas, jac, spec=None, oms=None, gtol=1e-10):
    lp, g = prob.logp_grad(thetas, jacobian=jac, spec=spec)
    for i, th in enumerate(thetas):
        m = om if oms is None else oms[spec[i]]
        lp_ref, g_ref = m.logp_grad(th, jacobian=jac)
        if not np.isfinite(lp_ref):
            assert lp[i] == lp_ref
            continue
        assert abs(lp[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref)), (i, lp[i], lp_ref)
        err = np.max(np.abs(g[i] - g_ref))
        assert err <= gtol * max(1.0, np.max(np.abs(g_ref))), (i, err, int(np.argmax(np.abs(g[i] - g_ref))))


def _bench_blocks(mode='sample', tag='K161'):
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
              induc_scale=float(d['induc_scale']))
    return d, blk, kw


@pytest.mark.parametrize('mode', ['sample', 'optimize'])
@pytest.mark.parametrize('B', [1, 5, 16, 33])
def test_benchmark_shape_vs_oracle(mode, B):
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks(mode)
    prob = Problem([blk], d['Z'], d['freq'], **kw)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    assert prob.D == om.D == 331
    rng = np.random.default_rng(B)
    thetas = rng.uniform(-2, 2, (B, prob.D))
    _compare(prob, om, thetas, mode == 'sample')
    _compare(prob, om, thetas, mode != 'sample')


@pytest.mark.parametrize('tag,K', [('K161', 161), ('K81', 81)])
def test_both_operand_paths_of_the_one_block_tile(tag, K, monkeypatch):
    """81 x 161 and 81 x 81 on log-uniform grids: the GEMMs take their A operands from the Toeplitz generator table in LDS
    (evaluator 4); BDRT_STREAM_A=1 pins the streamed fragments (evaluator 2).  Both against the oracle, and against each other."""
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample', tag)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    rng = np.random.default_rng(K)
    thetas = rng.uniform(-2, 2, (37, om.D))
    out = {}
    for stream in ('0', '1'):
        if stream == '1': monkeypatch.setenv('BDRT_STREAM_A', '1')
        else: monkeypatch.delenv('BDRT_STREAM_A', raising=False)
        prob = Problem([blk], d['Z'], d['freq'], **kw)
        assert prob.evaluator() == (2 if stream == '1' else 4)
        for jac in (True, False):
            _compare(prob, om, thetas, jac)
        out[stream] = prob.logp_grad(thetas, jacobian=True)
    assert np.max(np.abs(out['0'][0] - out['1'][0]) / np.maximum(1.0, np.abs(out['1'][0]))) < 1e-11
    assert np.max(np.abs(out['0'][1] - out['1'][1])) <= 1e-11 * max(1.0, np.max(np.abs(out['1'][1])))
    assert not np.array_equal(out['0'][1], out['1'][1])          # (another summation order: the switch did switch)


def _log_uniform_problem(nf, K, nonneg=True, **kw):
    """A synthetic single-DRT problem on log-uniform grids of equal spacing (A_re, A_im exactly Toeplitz), nf frequencies, K basis points."""
    from bayes_drt_amd import matrices as gm
    ppd = 10.0
    f = 10.0 ** (6.0 - np.arange(nf) / ppd)
    bf = 10.0 ** (6.0 + ((K - nf) // 2) / ppd - np.arange(K) / ppd)         # the basis brackets the measured range, on the same grid
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    rng = np.random.default_rng(nf * 1000 + K)
    Z = np.concatenate([1.0 + rng.random(nf), -rng.random(nf)]) + 0.01 * rng.standard_normal(2 * nf)
    blk = dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=nonneg)
    return blk, Z, f, dict(ups_alpha=1.0, ups_beta=0.1, **kw)


@pytest.mark.parametrize('nf', [80, 81, 82])
@pytest.mark.parametrize('K', [80, 81, 82, 160, 161, 162])
def test_every_shape_of_the_table_path_vs_oracle(nf, K):
    """The Toeplitz-table GEMMs take nf = 80..82 and K = 80..82 / 160..162: zero, one or two rows of each part of A (and of A^T)
    beyond the 16-row tiles go through the VALU dot products, and the rows of g beyond the chunks of four through the odd chunk of
    the backward GEMM.  Every combination against the oracle, with and without the sign constraint, and with the outlier error
    models (whose extra parameters the same evaluator handles outside the sampler)."""
    Problem, orc = _mods()
    for nonneg, extra in ((True, {}), (False, {}), (True, dict(outlier_mode=1, so_lambda=10.0, so_alpha=5.0, so_beta=1.0)),
                          (True, dict(outlier_mode=2, so_lambda=10.0))):
        blk, Z, f, kw = _log_uniform_problem(nf, K, nonneg, **extra)
        prob = Problem([blk], Z, f, **kw)
        # (with outlier parameters and K >= 160 the table does not fit beside the sampler's theta rows: streamed fragments, evaluator 2)
        assert prob.evaluator() == 4 or (extra and K >= 160 and prob.evaluator() == 2), (nf, K, extra, prob.evaluator())
        om = orc.OracleModel([blk], Z, f, **kw)
        rng = np.random.default_rng(nf + K)
        th = rng.uniform(-2, 2, (19, prob.D))
        _compare(prob, om, th, True)
        _compare(prob, om, th[:5], False)
        prob.close()


def test_shapes_beside_the_default_table_shapes():
    """nf = 41 / K = 51 (not blocks of 80; basis and measurement grids offset): whichever evaluator takes it, against the oracle
    (the general table routine has its own file, tests/test_gpu_toep_gen.py)."""
    Problem, orc = _mods()
    from bayes_drt_amd import matrices as gm
    f = np.logspace(5, 1, 41)
    bf = np.logspace(6, 1, 51); tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    rng = np.random.default_rng(5)
    Z = rng.standard_normal(82)
    blk = dict(A=A, L0=L[0], L1=L[1], L2=L[2], nonneg=True)
    prob = Problem([blk], Z, f, ups_alpha=1.0, ups_beta=0.1)
    assert prob.evaluator() in (2, 3, 4)
    om = orc.OracleModel([blk], Z, f, ups_alpha=1.0, ups_beta=0.1)
    _compare(prob, om, rng.uniform(-2, 2, (9, prob.D)), True)


def test_signed_x_series():
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample', 'K81')
    blk = dict(blk, nonneg=False)
    prob = Problem([blk], d['Z'], d['freq'], **kw)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    rng = np.random.default_rng(3)
    _compare(prob, om, rng.uniform(-2, 2, (7, prob.D)), True)


def test_multi_spectrum_batch():
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample')
    rng = np.random.default_rng(11)
    Zs = np.stack([d['Z'] * (1 + 0.05 * s) + 0.01 * rng.standard_normal(len(d['Z'])) for s in range(5)])
    prob = Problem([blk], Zs, d['freq'], **kw)
    oms = [orc.OracleModel([blk], Zs[s], d['freq'], **kw) for s in range(5)]
    spec = rng.integers(0, 5, 40).astype(np.int32)
    thetas = rng.uniform(-2, 2, (40, prob.D))
    _compare(prob, None, thetas, True, spec=spec, oms=oms)
    with pytest.raises(Exception):
        prob.logp_grad(thetas[:2], spec=np.array([0, 5], dtype=np.int32))


@pytest.mark.parametrize('name', ['RC-ZARC_uniform_0.25', 'trunc_uniform_0.25', 'LIB_data', 'PDAC_outliers',
                                  'DRT-2-TpDDT_uniform_0.25', 'PDAC_DRT-TpDDT_outliers',
                                  'DRT-TpDDT-BpDDT_uniform_0.25'])
def test_model_families_vs_oracle(name):
    """Series, Series_pos, old Series_outliers (stacked), Series-Parallel_pos (+outliers), Series-2Parallel_pos at the
    reference's stored MAP points and at perturbed points."""
    Problem, orc = _mods()
    k = kat_to_model(name)
    assert k is not None
    kw = dict(k['kw'])
    rng = np.random.default_rng(5)
    if not k['has_Z']:
        kw['Z'] = np.asarray(k['opt']['Z_hat']).ravel() + 0.01 * rng.standard_normal(len(kw['Z']))
    if kw['use_x_sum']:
        kw['x_sum_invscale'] = 0.3
    prob = Problem(**kw)
    om = orc.OracleModel(**kw)
    assert prob.D == om.D
    th0 = om.unconstrain(k['params'])
    thetas = np.stack([th0] + [th0 + 0.1 * rng.standard_normal(om.D) for _ in range(4)])
    for jac in (False, True):
        # at the stored MAPs the gradient is a near-total cancellation of O(1e4) terms (e.g. 100*sum(g_Zhat) for
        # Rinf_raw): the summation-order noise floor is ~1e-9 absolute, hence the looser bound here
        _compare(prob, om, thetas, jac, gtol=1e-8)
    # transformed parameters vs the stored Stan outputs (KAT through the GPU path)
    params, Zh, sg = prob.transformed(th0[None])
    np.testing.assert_allclose(params[0], k['params'], rtol=1e-12)
    ref = np.asarray(k['opt']['Z_hat']).ravel()
    assert np.max(np.abs(Zh[0] - ref)) <= 1e-11 * np.max(np.abs(ref))
    ref = np.asarray(k['opt']['sigma_tot']).ravel()
    assert np.max(np.abs(sg[0] - ref)) <= 1e-11 * np.max(np.abs(ref))


def test_outlier_mode1_and_parallel_only():
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample', 'K81')
    so = load('dat_sample_outlier_scalars')
    kw1 = dict(kw, outlier_mode=1, so_lambda=float(so['sigma_out_lambda']), so_alpha=float(so['sigma_out_alpha']),
               so_beta=float(so['sigma_out_beta']))
    prob = Problem([blk], d['Z'], d['freq'], **kw1)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw1)
    rng = np.random.default_rng(8)
    _compare(prob, om, rng.uniform(-1, 1, (9, prob.D)), True)
    # Parallel model (Parallel_modelcode.txt): a single parallel block built from the DDT golden matrices
    dd = load('ddt_toeplitz_81x161')
    A = np.vstack([dd['A_re_tp_parallel'], dd['A_im_tp_parallel']])
    m161 = load('mat_drt_81x161')
    blkp = dict(A=A, L0=m161['L0'], L1=m161['L1'], L2=0.75 * m161['L2'], parallel=True, x_scale=1.0)
    Zy = 1.0 / (d['Z'][:81] + 1j * d['Z'][81:])
    Zp = np.concatenate([Zy.real, Zy.imag]) * 0 + d['Z']
    prob = Problem([blkp], Zp, d['freq'], use_x_sum=False, **kw)
    om = orc.OracleModel([blkp], Zp, d['freq'], use_x_sum=False, **kw)
    _compare(prob, om, rng.uniform(-2, 0, (6, prob.D)), True)


def test_x_sum_rejection_flag():
    Problem, orc = _mods()
    k = kat_to_model('DRT-2-TpDDT_uniform_0.25')
    kw = dict(k['kw']); kw['blocks'] = [dict(b) for b in kw['blocks']]
    kw['blocks'][0]['nonneg'] = False
    prob = Problem(**kw)
    p = k['params'].copy()
    lay = prob.layout()
    p[lay['x'][0]:lay['x'][0] + prob.Ks[0]] = -10.0
    lp, g = prob.logp_grad(prob.unconstrain(p)[None], jacobian=True)
    assert lp[0] == -np.inf


def test_structured_L_path_equals_dense_path(monkeypatch):
    """The banded-Toeplitz convolution path (chosen automatically for log-uniform tau grids) and the dense MFMA path
    (any grid; forced with BDRT_DENSE_L=1) evaluate the same density: lp and gradient agree to fp64 round-off."""
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample')
    rng = np.random.default_rng(21)
    thetas = rng.uniform(-2, 2, (19, 331))
    fast = Problem([blk], d['Z'], d['freq'], **kw)
    lp_f, g_f = fast.logp_grad(thetas, jacobian=True)
    monkeypatch.setenv('BDRT_DENSE_L', '1')
    dense = Problem([blk], d['Z'], d['freq'], **kw)
    lp_d, g_d = dense.logp_grad(thetas, jacobian=True)
    assert np.max(np.abs(lp_f - lp_d) / np.maximum(1.0, np.abs(lp_d))) < 1e-12
    assert np.max(np.abs(g_f - g_d)) < 1e-11 * max(1.0, np.max(np.abs(g_d)))
    assert not np.array_equal(g_f, g_d)          # really two different code paths


@pytest.mark.gpu
@pytest.mark.parametrize('jacobian', [False, True])
def test_fast_s1_tile_equals_generic_tile(monkeypatch, jacobian):
    """The half-wave-per-chain evaluator of the headline family (bdrt_tile_s1.h) and the generic block evaluator
    (forced with BDRT_GENERIC_TILE=1) are the same function: lp, gradient, constrained parameters, Z_hat, sigma_tot."""
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample')
    rng = np.random.default_rng(22)
    thetas = rng.uniform(-2, 2, (37, 331))          # 37: two full tiles + a ragged one
    fast = Problem([blk], d['Z'], d['freq'], **kw)
    lp_f, g_f = fast.logp_grad(thetas, jacobian=jacobian)
    tr_f = fast.transformed(thetas)
    monkeypatch.setenv('BDRT_GENERIC_TILE', '1')
    gen = Problem([blk], d['Z'], d['freq'], **kw)
    lp_g, g_g = gen.logp_grad(thetas, jacobian=jacobian)
    tr_g = gen.transformed(thetas)
    assert np.max(np.abs(lp_f - lp_g) / np.maximum(1.0, np.abs(lp_g))) < 1e-12
    assert np.max(np.abs(g_f - g_g)) < 1e-11 * max(1.0, np.max(np.abs(g_g)))
    for a, b in zip(tr_f, tr_g):
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))
    assert not np.array_equal(g_f, g_g)          # really two different code paths


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [1, 2])
def test_fast_s1_tile_with_outlier_parameters_equals_generic_tile(monkeypatch, mode):
    """Series / Series_pos with the outlier error model (package form: raw[Nf], scale[Nf]; paper form: raw[2Nf]) also takes
    the half-wave evaluator; same function as the generic block evaluator and as the oracle."""
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample', 'K161')
    so = load('dat_sample_outlier_scalars')
    kw = dict(kw, outlier_mode=mode, so_lambda=float(so['sigma_out_lambda']), so_alpha=float(so['sigma_out_alpha']),
              so_beta=float(so['sigma_out_beta']))
    rng = np.random.default_rng(40 + mode)
    fast = Problem([blk], d['Z'], d['freq'], **kw)
    thetas = rng.uniform(-1.5, 1.5, (21, fast.D))
    assert fast.D == 331 + 162
    lp_f, g_f = fast.logp_grad(thetas, jacobian=True)
    tr_f = fast.transformed(thetas)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    _compare(fast, om, thetas[:6], True)
    monkeypatch.setenv('BDRT_GENERIC_TILE', '1')
    gen = Problem([blk], d['Z'], d['freq'], **kw)
    lp_g, g_g = gen.logp_grad(thetas, jacobian=True)
    tr_g = gen.transformed(thetas)
    assert np.max(np.abs(lp_f - lp_g) / np.maximum(1.0, np.abs(lp_g))) < 1e-12
    assert np.max(np.abs(g_f - g_g)) < 1e-11 * max(1.0, np.max(np.abs(g_g)))
    for a, b in zip(tr_f, tr_g):
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))
    assert not np.array_equal(g_f, g_g)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['DRT-2-TpDDT_uniform_0.25', 'DRT-TpDDT-BpDDT_uniform_0.25', 'PDAC_DRT-TpDDT_outliers'])
def test_general_half_wave_tile_equals_generic_tile(monkeypatch, name):
    """Several distributions / parallel blocks / x_sum prior / outlier parameters: the general half-wave evaluator
    (bdrt_tile_hw.h) and the generic block evaluator (BDRT_GENERIC_TILE=1) are the same function."""
    Problem, orc = _mods()
    k = kat_to_model(name)
    rng = np.random.default_rng(77)
    fast = Problem(**k['kw'])
    th0 = fast.unconstrain(k['params'])
    thetas = th0[None] + 0.3 * rng.standard_normal((19, fast.D))
    lp_f, g_f = fast.logp_grad(thetas, jacobian=True)
    tr_f = fast.transformed(thetas)
    monkeypatch.setenv('BDRT_GENERIC_TILE', '1')
    gen = Problem(**k['kw'])
    lp_g, g_g = gen.logp_grad(thetas, jacobian=True)
    tr_g = gen.transformed(thetas)
    fin = np.isfinite(lp_g)
    assert np.array_equal(fin, np.isfinite(lp_f)) and fin.sum() >= 10
    assert np.max(np.abs(lp_f[fin] - lp_g[fin]) / np.maximum(1.0, np.abs(lp_g[fin]))) < 1e-12
    assert np.max(np.abs(g_f[fin] - g_g[fin])) < 1e-11 * max(1.0, np.max(np.abs(g_g[fin])))
    for a, b in zip(tr_f, tr_g):
        assert np.max(np.abs(a[fin] - b[fin])) <= 1e-12 * max(1.0, np.max(np.abs(b[fin])))
    if 'PDAC' not in name:                       # (the experimental PDAC grid is not log-uniform: generic evaluator either way)
        assert not np.array_equal(g_f, g_g)      # really two different code paths


@pytest.mark.parametrize('B', [1, 4, 100, 256, 257, 1280, 1281])
def test_few_points_take_the_one_workgroup_evaluator_headline_family(monkeypatch, B):
    """bdrt_logp_grad with up to five points per CU: a workgroup per point with Toeplitz products (24 us -> 8 us at B = 1;
    beyond one point per CU the register-tight variant, three workgroups to a CU, in a grid-stride loop); B = 1281 is back on
    the tiles.  Against the oracle, and against the tile evaluator on the same points."""
    Problem, orc = _mods()
    d, blk, kw = _bench_blocks('sample')
    prob = Problem([blk], d['Z'], d['freq'], **kw)
    om = orc.OracleModel([blk], d['Z'], d['freq'], **kw)
    thetas = np.random.default_rng(100 + B).uniform(-2, 2, (B, prob.D))
    monkeypatch.delenv('BDRT_FEW_POINTS', raising=False)
    lp1, g1 = prob.logp_grad(thetas, jacobian=True)
    for i in (0, B // 2, B - 1):
        lp_ref, g_ref = om.logp_grad(thetas[i], jacobian=True)
        assert abs(lp1[i] - lp_ref) <= 1e-10 * max(1.0, abs(lp_ref))
        assert np.max(np.abs(g1[i] - g_ref)) <= 1e-10 * max(1.0, np.max(np.abs(g_ref)))
    monkeypatch.setenv('BDRT_FEW_POINTS', '0')
    lp0, g0 = prob.logp_grad(thetas, jacobian=True)
    assert np.max(np.abs(lp1 - lp0)) <= 1e-11 * np.max(np.abs(lp0))
    assert np.max(np.abs(g1 - g0)) <= 1e-10 * np.max(np.abs(g0))
    if B <= 1280:
        assert not np.array_equal(g1, g0)          # really another kernel: same numbers in another summation order
    else:
        assert np.array_equal(g1, g0) and np.array_equal(lp1, lp0)


@pytest.mark.parametrize('name', ['DRT-2-TpDDT_uniform_0.25', 'DRT-TpDDT-BpDDT_uniform_0.25'])
def test_few_points_general_block_models(monkeypatch, name):
    Problem, orc = _mods()
    k = kat_to_model(name)
    prob = Problem(**k['kw'])
    om = orc.OracleModel(**k['kw'])
    th0 = prob.unconstrain(k['params'])
    thetas = th0[None] + 0.05 * np.random.default_rng(3).standard_normal((6, prob.D))
    monkeypatch.delenv('BDRT_FEW_POINTS', raising=False)
    _compare(prob, om, thetas, True)
    _compare(prob, om, thetas, False)
