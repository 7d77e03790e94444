"""Rank program of tests/test_gpu_nccl.py, started by `python -m torch.distributed.run` (one process per rank).
backend nccl: every rank takes the GPU of its LOCAL_RANK (two or more devices: the real multi-GPU path over RCCL / xGMI);
backend gloo: every rank on device 0 (the 1-GPU box: the same program, the collectives on host tensors).
Runs parallel.sample_sharded (draws and summaries) and Inverter.fit_many (both modes) and writes what rank 0 got to argv[2]."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend, out_path = sys.argv[1], sys.argv[2]
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    import torch
    import torch.distributed as dist
    from bayes_drt_amd import _lib, parallel as par
    from bayes_drt_amd.inversion import Inverter
    from tests.test_gpu_fit_many import _spectra
    from tests.test_gpu_parallel import _problem_kwargs
    dev = local if backend == 'nccl' else 0
    torch.cuda.set_device(dev)
    lib = _lib.require_gpu()
    assert lib.bdrt_set_device(dev) == 0
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo')
    try:
        ones = torch.ones(1, dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(ones)
        ids = [None] * world
        dist.all_gather_object(ids, int(dev))
        pk = _problem_kwargs() if rank == 0 else None
        res = par.sample_sharded(pk, 5, 2, 8, 6, seed=99, control={'max_treedepth': 5}, gather='draws')
        summ = par.sample_sharded(pk, 5, 2, 8, 6, seed=99, control={'max_treedepth': 5}, gather='summary')
        one = par.sample_sharded(None if rank else dict(pk, Z=pk['Z'][:1]), 1, 4, 8, 6, seed=99, control={'max_treedepth': 5}, gather='draws')
        f, zs = _spectra(3)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            vs = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='sample', warmup=30, samples=20, chains=2, random_seed=77)
            vm = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
        if rank == 0:
            np.savez(out_path, ranks_seen=int(ones.item()), device_ids=np.array(ids), draws=res['draws'], lp=res['lp'], stats=res['stats'],
                     mean=res['mean'], pct=res['pct'], s_mean=summ['mean'], s_pct=summ['pct'], one_draws=one['draws'],
                     fit_theta=np.stack([v._sample_result.theta for v in vs]), fit_coef=np.stack([v.distribution_fits['DRT']['coef'] for v in vm]))
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
