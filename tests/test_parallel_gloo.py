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
        out[u] = rs.standard_normal((n_draws, D)) + Z[s].sum() + float(kw['sigma_min'])
        lp[u] = out[u].sum(axis=1)
        st[u, 0] = 10 * c
    return out, lp, st


def _problem(n_spectra):
    rs = np.random.RandomState(0)
    nf, K = 5, 4
    blk = dict(A=rs.standard_normal((2 * nf, K)), L0=np.eye(K), L1=rs.standard_normal((K, K)), L2=np.eye(K) * 2,
               nonneg=True)
    return dict(blocks=[blk], Z=rs.standard_normal((n_spectra, 2 * nf)), freq=np.logspace(2, 0, nf), sigma_min=0.002,
                ups_alpha=1.0, ups_beta=0.1, outlier_mode=0, use_x_sum=False)


def _worker(rank, world, port, n_spectra, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem(n_spectra) if rank == 0 else None
        draws, lp, st = par.sample_sharded(pk, n_spectra, 3, 5, 4, seed=11, sample_fn=_fake_sampler)
        if rank == world - 1:
            q.put((draws, lp, st))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(world, n_spectra):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_spectra, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 512):
        for w in (1, 2, 3, 8):
            b = [par.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


@pytest.mark.parametrize('n_spectra', [5, 1])
def test_world2_equals_world1(n_spectra):
    ref = _run(1, n_spectra)
    two = _run(2, n_spectra)
    for a, b in zip(ref, two):
        assert a.shape == b.shape and np.array_equal(a, b)
    pk = _problem(n_spectra)
    spec, chain = par.make_units(n_spectra, 3)
    direct = _fake_sampler(pk, spec, chain, 5, 4, 11, None)
    assert np.array_equal(direct[0], ref[0])


def test_world3_with_idle_rank():
    ref = _run(1, 2)
    three = _run(3, 2)          # 2 spectra on 3 ranks: one rank has no work
    for a, b in zip(ref, three):
        assert np.array_equal(a, b)
