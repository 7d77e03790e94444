"""GPU test (-m gpu) of parallel.sample_sharded with the real per-rank worker on an RCCL ("nccl") process group of one
rank: problem broadcast, device-resident draws gathered straight from HBM (`__cuda_array_interface__`, no host round trip),
per-spectrum summaries reduced on the device -- equal to a direct run of the same units."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_sample_sharded_on_rccl_group_equals_direct_run():
    import torch
    import torch.distributed as dist
    from bayes_drt_amd import parallel as par
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.model import Problem
    from tests.helpers import load
    d = load('dat_sample_2ZARC_uniform_0.25_K81')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    rs = np.random.RandomState(1)
    Z = np.stack([d['Z'], d['Z'] + 0.002 * rs.standard_normal(d['Z'].shape), d['Z'] * 1.01])
    pk = dict(blocks=[blk], Z=Z, freq=d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=1.0, ups_beta=0.1,
              induc_scale=1.0)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        ctl = {'max_treedepth': 5}
        res = par.sample_sharded(pk, 3, 2, 8, 6, seed=99, control=ctl, gather='draws')
        summ = par.sample_sharded(pk, 3, 2, 8, 6, seed=99, control=ctl, gather='summary')
    finally:
        dist.destroy_process_group()
    prob = Problem([blk], Z, d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=1.0, ups_beta=0.1)
    import ctypes as C
    from bayes_drt_amd import _lib
    c = _lib.NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(c)); c.max_treedepth = 5
    spec, chain = par.make_units(3, 2)
    draws, lp, diag = sample_units(prob, 6, 8, 6, 99, c, spec=spec, chain_ids=chain)
    assert np.array_equal(res['draws'], draws) and np.array_equal(res['lp'], lp)
    assert np.array_equal(res['stats'][:, 0], [x['n_leapfrog'] for x in diag])
    assert 'draws' not in summ and np.array_equal(summ['mean'], res['mean']) and np.array_equal(summ['pct'], res['pct'])
    for s in range(3):
        cons = prob.constrain(draws[2 * s:2 * s + 2].reshape(-1, prob.D))
        assert np.allclose(res['mean'][s], cons.mean(axis=0), rtol=1e-13, atol=0)
        assert np.allclose(res['pct'][s], np.percentile(cons, [2.5, 50.0, 97.5], axis=0), rtol=1e-14, atol=0)
    prob.close()


def _problem_kwargs():
    from tests.helpers import load
    d = load('dat_sample_2ZARC_uniform_0.25_K81')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    rs = np.random.RandomState(1)
    Z = np.stack([d['Z'] * (1 + 0.01 * k) + 0.002 * rs.standard_normal(d['Z'].shape) for k in range(5)])
    return dict(blocks=[blk], Z=Z, freq=d['freq'], sigma_min=float(d['sigma_min']), ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)


def _rank_main(rank, world, port, n_spectra, chains, gather, q):
    """One process per rank, every rank with the real GPU worker on the box's single device; gloo carries the broadcast and
    the gather (RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    from bayes_drt_amd import parallel as par
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pk = _problem_kwargs() if rank == 0 else None
        if pk is not None:
            pk['Z'] = pk['Z'][:n_spectra]
        res = par.sample_sharded(pk, n_spectra, chains, 8, 6, seed=99, control={'max_treedepth': 5}, gather=gather)
        if rank == 0:
            q.put({k: np.asarray(v) for k, v in res.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world,n_spectra,chains', [(2, 5, 2), (3, 5, 2), (2, 1, 4)])
def test_several_ranks_with_the_real_gpu_worker_equal_a_direct_run(world, n_spectra, chains):
    """world_size 2 and 3 with the real per-rank worker (HIP sampler) in every process: whole spectra per rank (config 4), and
    the chains of a single spectrum spread over the ranks (configs 3 / 5) -- the assembled draws, lp, leapfrog counts and the
    device-reduced summaries are those of one direct run of all units."""
    import ctypes as C
    import torch.multiprocessing as mp
    from bayes_drt_amd import _lib, parallel as par
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.model import Problem
    ctx = mp.get_context('spawn')
    out = {}
    for gather in ('draws', 'summary'):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_rank_main, args=(r, world, port, n_spectra, chains, gather, q)) for r in range(world)]
        for p in procs:
            p.start()
        out[gather] = q.get(timeout=400)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    res, summ = out['draws'], out['summary']
    pk = _problem_kwargs()
    Z = pk['Z'][:n_spectra]
    prob = Problem(pk['blocks'], Z, pk['freq'], sigma_min=pk['sigma_min'], ups_alpha=1.0, ups_beta=0.1)
    c = _lib.NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(c)); c.max_treedepth = 5
    spec, chain = par.make_units(n_spectra, chains)
    draws, lp, diag = sample_units(prob, n_spectra * chains, 8, 6, 99, c, spec=spec, chain_ids=chain)
    assert np.array_equal(res['draws'], draws) and np.array_equal(res['lp'], lp)
    assert np.array_equal(res['stats'][:, 0], [x['n_leapfrog'] for x in diag])
    assert 'draws' not in summ and np.array_equal(summ['mean'], res['mean']) and np.array_equal(summ['pct'], res['pct'])
    for s in range(n_spectra):
        cons = prob.constrain(draws[chains * s:chains * (s + 1)].reshape(-1, prob.D))
        assert np.allclose(res['mean'][s], cons.mean(axis=0), rtol=1e-12, atol=0)
    prob.close()
