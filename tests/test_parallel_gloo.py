"""N > 1 path on CPU: world_size-2 gloo process group.  The GPU sampler is replaced by a deterministic stand-in whose
output depends only on (seed, spectrum data, chain id), so the test checks the distribution logic itself: the
problem broadcast from rank 0, whole-spectrum block partition, the single all-gather, and that the assembled result
is identical for world sizes 1, 2 and 3 (sharding-independence)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayes_drt_amd import parallel as par


def _fake_sampler(kw, spec, chain_ids, warmup, n_draws, seed, control):
    Z = np.atleast_2d(kw['Z'])
    D = 2 * kw['blocks'][0]['A'].shape[1] + 9
    out = np.empty((len(spec), n_draws, D)); lp = np.empty((len(spec), n_draws)); st = np.zeros((len(spec), 5))
    for u, (s, c) in enumerate(zip(spec, chain_ids)):
        rs = np.random.RandomState((seed * 1000003 + int(c)) % (2 ** 31))
        out[u] = 0.3 * rs.standard_normal((n_draws, D)) + 0.01 * Z[s].sum() + float(kw['sigma_min'])
        lp[u] = out[u].sum(axis=1)
        st[u, 0] = 10 * c
    return out, lp, st


class FakeWorker:
    """Stand-in for parallel.GpuWorker (test-only injection): same protocol, numpy arithmetic, records which rank ran."""

    def __init__(self, kw):
        self.kw = kw
        self.D = 2 * kw['blocks'][0]['A'].shape[1] + 9

    def run(self, spec, chain_ids, warmup, n_draws, seed, control, init_theta=None):
        self._draws, self._lp, self._stats = _fake_sampler(self.kw, spec, chain_ids, warmup, n_draws, seed, control)
        if init_theta is not None:                     # the start point a unit was handed shows up in its first draw
            assert init_theta.shape == (len(spec), self.D)
            self._draws[:, 0, :] = init_theta
        self._stats[:, 1] = dist.get_rank()            # "n_divergent" column abused as the rank that sampled the unit
        self._stats[:, 2] = np.atleast_2d(self.kw['Z']).shape[0]     # "n_max_treedepth": spectra this rank was handed

    def lp(self):
        return self._lp

    def stats(self):
        return self._stats

    def draws(self, device):
        return self._draws

    def is_pos(self):
        m = np.zeros(self.D, dtype=bool); m[2:self.D - 3:2] = True
        return m

    def summary(self, lo, hi, q):
        return FakeWorker.reduce(self._draws[lo:hi].reshape(-1, self.D), self.is_pos(), q)

    @staticmethod
    def reduce(block, is_pos, q):
        c = np.where(is_pos, np.exp(block), block)
        return c.mean(axis=0), np.percentile(c, q, axis=0)

    def close(self):
        pass


def _problem(n_spectra):
    rs = np.random.RandomState(0)
    nf, K = 5, 4
    blk = dict(A=rs.standard_normal((2 * nf, K)), L0=np.eye(K), L1=rs.standard_normal((K, K)), L2=np.eye(K) * 2,
               nonneg=True)
    return dict(blocks=[blk], Z=rs.standard_normal((n_spectra, 2 * nf)), freq=np.logspace(2, 0, nf), sigma_min=0.002,
                ups_alpha=1.0, ups_beta=0.1, outlier_mode=0, use_x_sum=False)


def _worker(rank, world, port, n_spectra, chains, gather, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem(n_spectra) if rank == 0 else None
        res = par.sample_sharded(pk, n_spectra, chains, 5, 4, seed=11, worker_cls=FakeWorker, gather=gather)
        if rank == world - 1:
            q.put(res)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(world, n_spectra, chains=3, gather='draws'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_spectra, chains, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _same(a, b, skip=('stats',)):
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape, k
        if k not in skip:
            assert np.array_equal(a[k], b[k]), k


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 512):
        for w in (1, 2, 3, 8):
            b = [par.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def test_partition_units_whole_spectra_or_chains():
    # config 4: whole spectra per rank
    p = par.partition_units(512, 8, 8)
    assert p[0] == (0, 512) and p[-1] == (3584, 4096) and all((hi - lo) % 8 == 0 for lo, hi in p)
    # config 3 / 5: one spectrum, 4 chains => chains are spread, 4 of 8 ranks busy
    p = par.partition_units(1, 4, 8)
    assert [hi - lo for lo, hi in p] == [1, 1, 1, 1, 0, 0, 0, 0]
    assert par.partition_units(1, 4, 2) == [(0, 2), (2, 4)]
    assert par.partition_units(0, 4, 2) == [(0, 0), (0, 0)]
    for ns, ch, w in ((5, 3, 2), (2, 4, 3), (3, 2, 8)):
        p = par.partition_units(ns, ch, w)
        assert p[0][0] == 0 and p[-1][1] == ns * ch and all(p[i][1] == p[i + 1][0] for i in range(w - 1))


@pytest.mark.parametrize('n_spectra', [5, 2])
def test_world2_equals_world1(n_spectra):
    ref = _run(1, n_spectra)
    two = _run(2, n_spectra)
    _same(ref, two)
    assert np.array_equal(ref['stats'][:, 0], two['stats'][:, 0])
    assert set(two['stats'][:, 1]) == {0.0, 1.0}                 # both ranks sampled
    if n_spectra >= 2:
        # the spectra are scattered, not broadcast: a rank holds only the rows it samples (3 + 2 of 5; 1 + 1 of 2)
        held = {int(r): int(h) for r, h in zip(two['stats'][:, 1], two['stats'][:, 2])}
        assert held == ({0: 3, 1: 2} if n_spectra == 5 else {0: 1, 1: 1}), held
    pk = _problem(n_spectra)
    spec, chain = par.make_units(n_spectra, 3)
    direct = _fake_sampler(pk, spec, chain, 5, 4, 11, None)
    assert np.array_equal(direct[0], ref['draws'])
    m, p = FakeWorker.reduce(direct[0][:3].reshape(-1, direct[0].shape[2]), FakeWorker(pk).is_pos(), [2.5, 50.0, 97.5])
    assert np.array_equal(ref['mean'][0], m) and np.array_equal(ref['pct'][0], p)


def test_single_spectrum_chains_are_spread_over_ranks():
    """BASELINE configs 3 and 5: one spectrum, 4 chains, 2 ranks => each rank samples 2 chains; same result as 1 rank."""
    ref = _run(1, 1, chains=4)
    two = _run(2, 1, chains=4)
    _same(ref, two)
    assert list(two['stats'][:, 1]) == [0.0, 0.0, 1.0, 1.0]     # chains 0,1 on rank 0, chains 2,3 on rank 1
    three = _run(3, 1, chains=4, gather='summary')             # 4 units on 3 ranks; summaries only
    assert 'draws' not in three
    assert np.array_equal(three['mean'], ref['mean']) and np.array_equal(three['pct'], ref['pct'])


def test_summary_gather_moves_no_draws():
    ref = _run(1, 4, gather='draws')
    summ = _run(2, 4, gather='summary')
    assert 'draws' not in summ
    for k in ('mean', 'pct', 'lp'):
        assert np.array_equal(ref[k], summ[k]), k


def test_world3_with_idle_rank():
    ref = _run(1, 2)
    three = _run(3, 2)          # 2 spectra x 3 chains on 3 ranks: chains are spread (2 units per rank)
    _same(ref, three)
    eight = _run(8, 1, chains=4)        # more ranks than units: four ranks idle
    _same(_run(1, 1, chains=4), eight)


def test_no_spectra_is_an_empty_result():
    res = _run(2, 0)
    assert res['draws'].shape[0] == 0 and res['stats'].shape == (0, 5)


def test_broadcast_rejects_what_float64_cannot_carry():
    with pytest.raises(TypeError):
        par._pack_problem(dict(blocks=[dict(A=np.ones((2, 2)) * 1j, L0=np.eye(2), L1=np.eye(2), L2=np.eye(2))], Z=np.ones(2),
                               freq=np.ones(1)))


def _worker_init(rank, world, port, n_spectra, chains, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem(n_spectra) if rank == 0 else None
        D = 2 * 4 + 9
        it = np.arange(n_spectra * chains * D, dtype=float).reshape(n_spectra * chains, D) if rank == 0 else None
        res = par.sample_sharded(pk, n_spectra, chains, 5, 4, seed=11, worker_cls=FakeWorker, gather='draws', init_theta=it)
        if rank == world - 1:
            q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_spectra,chains', [(1, 3, 2), (2, 3, 2), (2, 1, 4), (3, 2, 2)])
def test_start_points_travel_with_their_units(world, n_spectra, chains):
    """`init_theta` rows (Stan `init=` values of Inverter.fit_many(init_from_ridge=True)) reach the rank that samples the unit, whatever
    the partition: whole spectra per rank, or the chains of one spectrum spread over the ranks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_init, args=(r, world, port, n_spectra, chains, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    D = 2 * 4 + 9
    assert np.array_equal(res['draws'][:, 0, :], np.arange(n_spectra * chains * D, dtype=float).reshape(n_spectra * chains, D))


def _worker_bad(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem(3) if rank == 0 else None        # three spectra handed over, four announced
        try:
            par.sample_sharded(pk, 4, 2, 5, 4, seed=11, worker_cls=FakeWorker)
            q.put((rank, 'no error'))
        except ValueError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_a_bad_argument_on_rank_0_fails_on_every_rank_instead_of_hanging():
    """rank 0 used to raise before the collectives while the other ranks waited in the broadcast until the
    process-group timeout.  The verdict of the argument check now travels first and every rank raises."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert set(got) == {0, 1} and all('Z has 3 spectra' in m for m in got.values())


def _worker_bad_width(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem(1) if rank == 0 else None
        it = np.zeros((2, 2 * 4 + 9 + 1)) if rank == 0 else None      # one column too many
        try:
            par.sample_sharded(pk, 1, 2, 5, 4, seed=11, worker_cls=FakeWorker, init_theta=it)
            q.put((rank, 'no error'))
        except ValueError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_start_points_of_the_wrong_width_fail_on_every_rank_including_the_idle_one():
    """the width of `init_theta` was only checked inside `worker.run`, i.e. on ranks that own units; a rank without
    units (one spectrum x 2 chains on 3 ranks) went on into the gathers and waited for the process-group timeout."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_width, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert set(got) == {0, 1, 2} and all('18 columns' in m and 'D = 17' in m for m in got.values()), got
