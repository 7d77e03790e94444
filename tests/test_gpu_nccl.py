"""The multi-GPU path with REAL ranks (-m gpu): `parallel.sample_sharded`, `Inverter.fit_many` and `bench.py --gpus 2` under the
driver's launcher (`python -m torch.distributed.run`, one process per rank).
  * two or more devices visible: backend "nccl" -- RCCL over xGMI, every rank on its own GPU; asserts that two ranks took part, on
    distinct devices, and that the assembled results equal one direct run.  SKIPPED (not passed) on a 1-GPU box.
  * the same rank program on the 1-GPU box: both ranks on device 0, gloo carrying the collectives (RCCL refuses two ranks on one
    device) -- so the program the multi-GPU test runs is exercised wherever the suite runs."""
import os
import socket
import subprocess
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _device_count():
    from bayes_drt_amd import _lib
    return int(_lib.require_gpu().bdrt_device_count())


def _run_ranks(backend, world, tmp_path):
    out = str(tmp_path / 'rank0.npz')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'nccl_worker.py'), backend, out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def _check_against_direct_run(got, world):
    import ctypes as C
    from bayes_drt_amd import _lib, parallel as par
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.inversion import Inverter
    from bayes_drt_amd.model import Problem
    from tests.test_gpu_fit_many import _spectra
    from tests.test_gpu_parallel import _problem_kwargs
    assert int(got['ranks_seen']) == world
    pk = _problem_kwargs()
    prob = Problem(pk['blocks'], pk['Z'], pk['freq'], sigma_min=pk['sigma_min'], ups_alpha=1.0, ups_beta=0.1)
    c = _lib.NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(c)); c.max_treedepth = 5
    spec, chain = par.make_units(5, 2)
    draws, lp, diag = sample_units(prob, 10, 8, 6, 99, c, spec=spec, chain_ids=chain)
    assert np.array_equal(got['draws'], draws) and np.array_equal(got['lp'], lp)
    assert np.array_equal(got['stats'][:, 0], [x['n_leapfrog'] for x in diag])
    assert np.array_equal(got['s_mean'], got['mean']) and np.array_equal(got['s_pct'], got['pct'])
    for s in range(5):
        cons = prob.constrain(draws[2 * s:2 * s + 2].reshape(-1, prob.D))
        assert np.allclose(got['mean'][s], cons.mean(axis=0), rtol=1e-12, atol=0)
    prob.close()
    p1 = Problem(pk['blocks'], pk['Z'][:1], pk['freq'], sigma_min=pk['sigma_min'], ups_alpha=1.0, ups_beta=0.1)
    d1, _, _ = sample_units(p1, 4, 8, 6, 99, c, spec=np.zeros(4, dtype=np.int32), chain_ids=np.arange(4, dtype=np.int32))
    assert np.array_equal(got['one_draws'], d1)                      # one spectrum, its four chains spread over the ranks
    p1.close()
    f, zs = _spectra(3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        vs = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='sample', warmup=30, samples=20, chains=2, random_seed=77)
        vm = Inverter(basis_freq=f).fit_many(f, zs, nonneg=True, mode='optimize')
    assert np.array_equal(got['fit_theta'], np.stack([v._sample_result.theta for v in vs]))
    assert np.allclose(got['fit_coef'], np.stack([v.distribution_fits['DRT']['coef'] for v in vm]), rtol=1e-9, atol=1e-12)


@pytest.mark.timeout(1500)
def test_two_rccl_ranks_on_two_gpus_equal_a_direct_run(tmp_path):
    if _device_count() < 2:
        pytest.skip('one GPU visible: the RCCL path with more than one rank needs two devices (the gloo variant below runs the same program)')
    got = _run_ranks('nccl', 2, tmp_path)
    assert sorted(int(i) for i in got['device_ids']) == [0, 1]
    _check_against_direct_run(got, 2)


@pytest.mark.timeout(1500)
def test_bench_two_rccl_ranks_on_two_gpus():
    """`bench.py --gpus 2` as the driver launches it, on two devices over RCCL."""
    if _device_count() < 2:
        pytest.skip('one GPU visible')
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'BDRT_BENCH_ONE_DEVICE')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--rounds', '40', '--no-cpu-baseline',
           '--spectra', '32']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and sorted(d['config']['rccl']['device_ids']) == [0, 1]
    assert d['config']['rccl']['backend'].startswith('nccl') and 'test_mode' not in d['config']
    assert d['config']['evals_in_timed_region'] == 2 * 32 * 8 * 40 * 3
    rt = d['config']['dist_roundtrip']
    assert 'error' not in rt and rt['finite']


@pytest.mark.timeout(1500)
def test_the_same_rank_program_on_one_device_over_gloo(tmp_path):
    got = _run_ranks('gloo', 2, tmp_path)
    assert [int(i) for i in got['device_ids']] == [0, 0]
    _check_against_direct_run(got, 2)


@pytest.mark.timeout(1800)
def test_the_same_rank_program_with_eight_ranks_on_one_device_over_gloo(tmp_path):
    """World size 8 -- the size of the node the driver scales to -- on the 1-GPU box: 5 spectra x 2 chains and 1 spectrum x 4
    chains leave ranks with one unit or none (idle ranks must walk through every collective), `fit_many` shards 3 spectra
    (6 units, 6 MAP rows) over 8 ranks.  Same assertions as with two ranks: everything equals one direct run."""
    got = _run_ranks('gloo', 8, tmp_path)
    assert [int(i) for i in got['device_ids']] == [0] * 8
    _check_against_direct_run(got, 8)
