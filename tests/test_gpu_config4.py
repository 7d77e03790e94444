"""GPU tests (-m gpu) at BASELINE config 4's FULL size: 512 synthetic spectra x 8 chains = 4096 units on one GPU (one
16-chain workgroup per CU) -- the workload bench.py times, here with its results checked:
  * every draw of every unit finite, every chain past warm-up;
  * >= 8 (spectrum, chain) units, including spectrum 511 and a unit in the last workgroup, compared draw by draw with the
    recursive CPU oracle NUTS (same Philox streams => identical tree shapes, draws equal to 1e-6);
  * per-spectrum posterior summaries reduced on the device (bdrt_sampler_summary) equal numpy on the constrained draws."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('operands', ['table', 'streamed'])
def test_config4_full_size_draws_match_oracle_and_summaries_match_numpy(operands, monkeypatch):
    """operands: the A operands of the evaluator's GEMMs from the Toeplitz generator table in LDS (the default for this shape) or
    streamed as packed fragments from L2 (BDRT_STREAM_A=1, the path of every other shape)."""
    import bench
    if operands == 'streamed': monkeypatch.setenv('BDRT_STREAM_A', '1')
    else: monkeypatch.delenv('BDRT_STREAM_A', raising=False)
    from bayes_drt_amd import _lib
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.parallel import make_units
    from oracle import oracle as orc
    lib = _lib.require_gpu()
    kw = bench.build_problem_kwargs(bench.N_SPECTRA)
    blocks, Z, freq = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, freq, **kw)
    assert prob.D == 331 and Z.shape == (512, 162)
    assert prob.evaluator() == (4 if operands == 'table' else 2)
    spec, chain = make_units(bench.N_SPECTRA, bench.CHAINS_PER_SPECTRUM)
    ctrl = _lib.NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl))
    ctrl.adapt_delta, ctrl.adapt_t0, ctrl.max_treedepth = 0.9, 10.0, 5
    warm, nd = 6, 4
    with Sampler(prob, len(spec), warm, nd, 1234, ctrl, spec=spec, chain_ids=chain) as smp:
        smp.run()
        draws, lp, diag = smp.results()
        assert draws.shape == (4096, nd, 331) and np.all(np.isfinite(draws)) and np.all(np.isfinite(lp))
        assert all(d['n_leapfrog'] > 0 for d in diag)
        # (spectrum, chain) units spread over the grid: first / middle / last workgroups, spectrum 511 twice
        picks = [(0, 0), (0, 7), (37, 3), (255, 5), (256, 0), (400, 6), (511, 0), (511, 7), (129, 2)]
        octrl = orc.nuts_control(adapt_delta=0.9, adapt_t0=10.0, max_treedepth=5)
        for s, c in picks:
            u = s * bench.CHAINS_PER_SPECTRUM + c
            assert spec[u] == s and chain[u] == c
            om = orc.OracleModel(blocks, Z[s], freq, **kw)
            ref, lpr, dr = orc.nuts_sample(om, c, 1234, warm, nd, control=octrl)
            assert dr['n_leapfrog'] == diag[u]['n_leapfrog'], (s, c, dr, diag[u])
            assert dr['n_divergent'] == diag[u]['n_divergent']
            assert np.max(np.abs(draws[u] - ref)) < 1e-6 * np.max(np.abs(ref)), (s, c)
            assert np.allclose(lp[u], lpr, rtol=1e-8, atol=1e-6)
        # device summaries of whole spectra (what a multi-GPU run gathers) vs numpy on the constrained draws
        q = [2.5, 50.0, 97.5]
        for s in (0, 300, 511):
            mean, pct = smp.summary(8 * s, 8 * s + 8, q)
            cons = prob.constrain(draws[8 * s:8 * s + 8].reshape(-1, prob.D))
            assert np.allclose(mean, cons.mean(axis=0), rtol=1e-13, atol=0)
            assert np.allclose(pct, np.percentile(cons, q, axis=0), rtol=1e-14, atol=0)
    prob.close()


def test_summary_on_host_draws_equals_summary_on_the_sampler():
    """post.summary (host buffer; used when the chains of a spectrum were sampled on several GPUs) performs the arithmetic
    of Sampler.summary: bit-identical."""
    from bayes_drt_amd import post
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from tests.helpers import load
    d = load('dat_sample_2ZARC_uniform_0.25_K81')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    prob = Problem([blk], d['Z'], d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=1.0, ups_beta=0.1)
    ctrl = None
    with Sampler(prob, 4, 8, 16, 7, ctrl) as smp:
        smp.run()
        draws, _, _ = smp.results()
        m1, p1 = smp.summary(0, 4, [5.0, 50.0, 95.0])
        m2, p2 = post.summary(draws.reshape(-1, prob.D), [5.0, 50.0, 95.0], prob.is_pos)
        assert np.array_equal(m1, m2) and np.array_equal(p1, p2)
        m3, p3 = smp.summary(1, 3, [50.0])
        m4, p4 = post.summary(draws[1:3].reshape(-1, prob.D), [50.0], prob.is_pos)
        assert np.array_equal(m3, m4) and np.array_equal(p3, p4)
        cons = prob.constrain(draws.reshape(-1, prob.D))
        assert np.allclose(p1, np.percentile(cons, [5.0, 50.0, 95.0], axis=0), rtol=1e-14, atol=0)
    prob.close()


def test_tail_of_a_large_run_moves_to_the_one_chain_kernels(monkeypatch):
    """A run with more than eight chains per CU starts on the 16-chain kernel; once eight per CU or fewer are still alive
    `bdrt_sampler_run` hands them to the one-chain layout (nuts_migrate_kernel), where the one-chain-per-wave kernel advances
    them while more than 2.5 per CU are running and the 512-thread one-chain kernels finish the rest.  Same random numbers, same
    arithmetic up to summation order: the run with the hand-over equals the run without it (BDRT_TAIL_MIGRATION=0) chain by
    chain."""
    import bench
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    ns = 300
    kw = bench.build_problem_kwargs(ns)
    blocks, Z, freq = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, freq, **kw)
    n_units, warm, nd = ns * 8, 24, 12
    spec = np.repeat(np.arange(ns, dtype=np.int32), 8)
    cid = np.tile(np.arange(8, dtype=np.int32), ns)
    from bayes_drt_amd._lib import NutsControl
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 6

    def run():
        with Sampler(prob, n_units, warm, nd, 77, ctrl, spec=spec, chain_ids=cid) as smp:
            assert smp.kind() == 0
            smp.run()
            tail = smp.tail_units()
            return smp.results() + (tail, smp.kind())

    d1, lp1, g1, tail1, kind1 = run()
    monkeypatch.setenv('BDRT_TAIL_MIGRATION', '0')
    d0, lp0, g0, tail0, kind0 = run()
    assert tail0 == 0 and kind0 == 0 and 0 < tail1 <= 2048 and kind1 in (1, 3), (tail0, tail1, kind0, kind1)
    assert np.all(np.isfinite(d1)) and np.all(np.isfinite(lp1))
    err = np.max(np.abs(d1 - d0), axis=(1, 2)) / np.max(np.abs(d0))
    # chains that had finished before the hand-over are untouched; the others continue with other summation orders
    assert np.mean(err == 0.0) > 0.1 and np.mean(err < 1e-6) > 0.9, (np.mean(err == 0.0), np.mean(err < 1e-6))
    n1, n0 = sum(x['n_leapfrog'] for x in g1), sum(x['n_leapfrog'] for x in g0)
    assert abs(n1 - n0) < 0.01 * n0


@pytest.mark.parametrize('nf,K', [(81, 101), (41, 51)])
def test_hand_over_and_compaction_at_the_other_row_strides(nf, K, monkeypatch):
    """The 16-chain kernel instantiated by basis length keeps its state rows with a stride of 32 x (4, 6, 7) doubles instead of
    32 x 11: the hand-over to the one-chain layout (nuts_migrate_kernel; 2400 units) and the compaction of an oversubscribed
    run (nuts_compact_kernel; 5120 units) at those strides, against the runs without either."""
    import bench
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd._lib import NutsControl
    for ns, warm, nd, depth in ((300, 24, 12, 6), (640, 16, 8, 5)):
        prob = Problem(**bench.shape_problem_kwargs(nf, K, ns))
        assert prob.evaluator() == 4
        n_units = ns * 8
        spec = np.repeat(np.arange(ns, dtype=np.int32), 8)
        cid = np.tile(np.arange(8, dtype=np.int32), ns)
        ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = depth

        def run():
            with Sampler(prob, n_units, warm, nd, 77, ctrl, spec=spec, chain_ids=cid) as smp:
                assert smp.kind() == 0
                smp.run()
                return smp.results() + (smp.tail_units(), smp.compactions())

        monkeypatch.delenv('BDRT_TAIL_MIGRATION', raising=False); monkeypatch.delenv('BDRT_COMPACTION', raising=False)
        d1, lp1, g1, tail1, nc1 = run()
        monkeypatch.setenv('BDRT_TAIL_MIGRATION', '0'); monkeypatch.setenv('BDRT_COMPACTION', '0')
        d0, lp0, g0, tail0, nc0 = run()
        assert tail0 == 0 and nc0 == 0 and (tail1 > 0 if ns == 300 else nc1 >= 1), (ns, tail0, nc0, tail1, nc1)
        assert np.all(np.isfinite(d1)) and np.all(np.isfinite(lp1))
        err = np.max(np.abs(d1 - d0), axis=(1, 2)) / np.max(np.abs(d0))
        assert np.mean(err < 1e-6) > 0.9, (ns, np.mean(err < 1e-6))
        n1, n0 = sum(x['n_leapfrog'] for x in g1), sum(x['n_leapfrog'] for x in g0)
        assert abs(n1 - n0) < 0.01 * n0
        prob.close()


def test_compaction_of_an_oversubscribed_run_is_bit_identical(monkeypatch):
    """More than 16 units per CU (3 x BASELINE config 4's shard: 1536 spectra x 8 chains = 12288 units = 768 workgroups):
    `run` re-packs the live chains into fewer full workgroups as chains finish (nuts_compact_kernel), so that the 16-column
    MFMA tiles stay full.  Every chain continues bit for bit: draws, lp and per-chain diagnostics equal the run with the
    layout frozen (BDRT_COMPACTION=0), and picks equal the same chains sampled alone."""
    import bench
    from bayes_drt_amd import _lib
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.parallel import make_units
    lib = _lib.require_gpu()
    ns = 3 * bench.N_SPECTRA
    kw = bench.build_problem_kwargs(ns)
    blocks, Z, freq = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, freq, **kw)
    spec, chain = make_units(ns, bench.CHAINS_PER_SPECTRUM)
    ctrl = _lib.NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl))
    ctrl.adapt_delta, ctrl.adapt_t0, ctrl.max_treedepth = 0.9, 10.0, 6
    warm, nd = 24, 8                                   # chain lengths spread over a factor > 2: columns empty out early

    def run(compaction):
        monkeypatch.setenv('BDRT_COMPACTION', '1' if compaction else '0')
        monkeypatch.setenv('BDRT_TAIL_MIGRATION', '0')          # (the hand-over to the one-chain kernel re-orders sums: own test)
        with Sampler(prob, len(spec), warm, nd, 77, ctrl, spec=spec, chain_ids=chain) as smp:
            assert smp.kind() == 0
            smp.run()
            return smp.results() + (smp.compactions(),)
    d1, lp1, g1, nc1 = run(True)
    d0, lp0, g0, nc0 = run(False)
    assert nc0 == 0 and nc1 >= 2, (nc0, nc1)
    assert np.all(np.isfinite(d1)) and np.array_equal(d1, d0) and np.array_equal(lp1, lp0)
    assert [x['n_leapfrog'] for x in g1] == [x['n_leapfrog'] for x in g0]
    assert [x['stepsize'] for x in g1] == [x['stepsize'] for x in g0]
    lens = np.array([x['n_leapfrog'] for x in g1])
    assert lens.max() > 1.5 * np.median(lens)
    # and the slot a chain runs in never mattered: three of them alone
    pick = [5, 6001, 12287]
    monkeypatch.setenv('BDRT_SOLO', '0')                        # (three chains would otherwise take a one-chain kernel)
    monkeypatch.setenv('BDRT_WIDE1', '0')
    with Sampler(prob, 3, warm, nd, 77, ctrl, spec=spec[pick], chain_ids=chain[pick]) as smp:
        assert smp.kind() == 0
        smp.run()
        da, lpa, _ = smp.results()
    assert np.array_equal(da, d1[pick]) and np.array_equal(lpa, lp1[pick])
    prob.close()


@pytest.mark.parametrize('nf,K', [(80, 160), (82, 162), (82, 80), (80, 81)])
def test_sixteen_chain_kernel_on_the_other_shapes_of_the_table_path_vs_oracle(nf, K, monkeypatch):
    """The sampler's fast path with the Toeplitz-table GEMMs at the shapes beside 81 x 161 (no / two odd rows per part, no / two odd
    rows of A^T): a few chains forced onto the 16-chain kernel, draw by draw against the recursive CPU oracle."""
    from tests.test_gpu_model import _log_uniform_problem
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
    warm, nd, n_units = 6, 4, 5
    spec = np.zeros(n_units, dtype=np.int32); cid = np.arange(n_units, dtype=np.int32)
    with Sampler(prob, n_units, warm, nd, 4321, ctrl, spec=spec, chain_ids=cid) as smp:
        assert smp.kind() == 0
        smp.run()
        draws, lp, diag = smp.results()
    om = orc.OracleModel([blk], Z, f, **kw)
    octrl = orc.nuts_control(max_treedepth=5)
    for c in range(n_units):
        ref, lpr, dr = orc.nuts_sample(om, c, 4321, warm, nd, control=octrl)
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref)), c
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)
    prob.close()
