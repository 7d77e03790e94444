"""GPU tests (-m gpu) of the Toeplitz-table GEMMs on ANY log-uniform shape (bdrt_tile_s1.h::toep_gemm_gen, DevProblem::toepA == 2):
partial 16-row tiles, reductions of any length, the tiles beyond the rounds of eight shared in halves.  The model code is the one of
the headline family (reference bayes_drt/stan_model_files/Series_pos_modelcode.txt; matrices as bayes_drt/inversion.py:2127-2209
builds them on the reference's default grids), the evaluator reports code 4 as on the default shapes."""
import ctypes as C

import numpy as np
import pytest

from tests.test_gpu_model import _compare, _log_uniform_problem, _mods

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_evaluators(monkeypatch):
    """These tests pin the 16-column tile evaluator (small batches would otherwise take the one-workgroup-per-point evaluators)."""
    monkeypatch.setenv('BDRT_FEW_POINTS', '0')

# frequencies x basis functions: the review's table (package default 81 x 101; 81 x 81; 41 x 51; 53 x 81; 106 x 101), the limits
# (32 x 32, 96 x 171, 128 x 150; below 32 a reduction has fewer than two whole quads: streamed fragments), and shapes that reach every branch
# of the schedule: R = 1..7 tiles beyond the rounds of eight, reductions of 2..12 quads with 0..3 single chunks, two or fewer rows
# beyond the whole tiles (VALU rows) against three or more (a partial tile)
SHAPES = [(81, 101), (81, 81), (41, 51), (53, 81), (106, 101), (32, 32), (128, 192), (33, 47), (100, 96), (64, 64), (35, 190),
          (36, 63), (66, 115), (113, 49), (97, 131), (48, 179), (125, 32), (34, 144), (96, 171), (128, 150), (16, 16), (31, 80), (80, 31)]


@pytest.mark.parametrize('nf,K', SHAPES)
def test_any_shape_takes_the_table_and_matches_the_oracle(nf, K):
    Problem, orc = _mods()
    for nonneg, extra in ((True, {}), (False, {}), (True, dict(outlier_mode=1, so_lambda=10.0, so_alpha=5.0, so_beta=1.0)),
                          (True, dict(outlier_mode=2, so_lambda=10.0))):
        blk, Z, f, kw = _log_uniform_problem(nf, K, nonneg, **extra)
        prob = Problem([blk], Z, f, **kw)
        # (more than 352 parameters: the table does not fit beside the sampler's theta rows -- streamed fragments, evaluator 2)
        small = nf < 32 or K < 32
        assert prob.evaluator() == (2 if small else 4) or (not extra and K > 171 and prob.evaluator() == 2), (nf, K, extra, prob.evaluator())
        om = orc.OracleModel([blk], Z, f, **kw)
        rng = np.random.default_rng(nf + K)
        th = rng.uniform(-2, 2, (37, prob.D))              # (three workgroups, the last with five of sixteen columns)
        _compare(prob, om, th, True)
        _compare(prob, om, th[:5], False)
        prob.close()


@pytest.mark.parametrize('nf,K', [(81, 161), (80, 160), (82, 82), (81, 81), (81, 101), (41, 51)])
def test_general_routine_against_the_streamed_fragments_and_the_default_shapes_routine(nf, K, monkeypatch):
    """The same problem through the general routine (BDRT_TOEP_GEN=1 also on the default shapes), the default shapes' routine
    and the streamed fragments: equal up to summation order, and not bit-equal (the switches did switch)."""
    Problem, orc = _mods()
    blk, Z, f, kw = _log_uniform_problem(nf, K)
    rng = np.random.default_rng(K)
    th = rng.uniform(-2, 2, (21, 2 * K + 9))
    out = {}
    for name, env in (('gen', dict(BDRT_TOEP_GEN='1')), ('default', {}), ('stream', dict(BDRT_STREAM_A='1'))):
        for k in ('BDRT_TOEP_GEN', 'BDRT_STREAM_A'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = Problem([blk], Z, f, **kw)
        assert prob.evaluator() == (2 if name == 'stream' else 4)
        out[name] = prob.logp_grad(th, jacobian=True)
        prob.close()
    for a, b in (('gen', 'stream'), ('default', 'stream')):
        assert np.max(np.abs(out[a][0] - out[b][0]) / np.maximum(1.0, np.abs(out[b][0]))) < 1e-11
        assert np.max(np.abs(out[a][1] - out[b][1])) <= 1e-11 * max(1.0, np.max(np.abs(out[b][1])))
    if (nf, K) != (41, 51):          # (there the two happen to add in the same order)
        assert not np.array_equal(out['gen'][1], out['stream'][1])
    if (nf, K) in ((82, 82), (81, 81)):       # (at K = 160 .. 162 the longer table of the general routine does not fit: the default one runs)
        assert not np.array_equal(out['gen'][1], out['default'][1])


@pytest.mark.parametrize('nf,K', [(81, 101), (41, 51), (106, 101), (53, 81), (113, 49), (64, 171)])
def test_sixteen_chain_sampler_on_the_general_routine_vs_oracle(nf, K, monkeypatch):
    """The 16-chain sampler kernel with the general table GEMMs: a few chains draw by draw against the recursive CPU oracle
    (reference call site bayes_drt/inversion.py:1218-1221)."""
    from bayes_drt_amd import _lib
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    lib = _lib.require_gpu()
    monkeypatch.setenv('BDRT_SOLO', '0'); monkeypatch.setenv('BDRT_WIDE1', '0')
    blk, Z, f, kw = _log_uniform_problem(nf, K)
    prob = Problem([blk], Z, f, **kw)
    assert prob.evaluator() == 4
    ctrl = _lib.NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl))
    ctrl.max_treedepth = 5
    warm, nd, n_units = 6, 4, 19                            # (two workgroups, the second with three columns)
    spec = np.zeros(n_units, dtype=np.int32); cid = np.arange(n_units, dtype=np.int32)
    with Sampler(prob, n_units, warm, nd, 4321, ctrl, spec=spec, chain_ids=cid) as smp:
        assert smp.kind() == 0
        smp.run()
        draws, lp, diag = smp.results()
    om = orc.OracleModel([blk], Z, f, **kw)
    octrl = orc.nuts_control(max_treedepth=5)
    for c in (0, 7, 15, 16, 18):
        ref, lpr, dr = orc.nuts_sample(om, c, 4321, warm, nd, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)
    prob.close()
