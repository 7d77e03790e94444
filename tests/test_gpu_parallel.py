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
