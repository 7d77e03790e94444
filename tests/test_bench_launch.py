"""bench.py's launch contract, on the CPU box (no GPU): `--gpus N` never yields a line with n_gpus = N from fewer than N
ranks.  Started bare it becomes the launcher of a child `torch.distributed.run` (reference: one process per chain,
bayes_drt/inversion.py:1218-1221) or exits non-zero; under a launcher whose WORLD_SIZE differs from --gpus it refuses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
ARGS = ['--steps', '1', '--warmup', '0', '--rounds', '5', '--spectra', '2', '--no-cpu-baseline']


def _run(args, env_extra=None, drop=('RANK', 'WORLD_SIZE', 'LOCAL_RANK')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def _lines(out):
    return [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_bare_multi_gpu_request_without_devices_exits_nonzero():
    out = _run(['--gpus', '2'] + ARGS, drop=('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'BDRT_BENCH_ONE_DEVICE'))
    assert out.returncode != 0 and not _lines(out)
    assert 'device(s) visible' in out.stderr


def test_bare_multi_gpu_request_launches_ranks_as_a_child():
    """With the test switch the device-count check is skipped, so the child launcher really starts two ranks; on this box
    they stop at 'needs a GPU' -- the parent relays the failure (rc != 0, no line).  On the GPU box the same command yields
    the 2-rank line (tests/test_gpu_bench.py)."""
    out = _run(['--gpus', '2'] + ARGS, {'BDRT_BENCH_ONE_DEVICE': '1'})
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(_lines(out)[0])
        assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2
    else:
        assert out.returncode != 0 and not _lines(out)
        assert 'launcher exited' in out.stderr


def test_world_size_mismatch_is_refused():
    out = _run(['--gpus', '8'] + ARGS, {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'}, drop=())
    assert out.returncode != 0 and not _lines(out)
    assert 'WORLD_SIZE=1' in out.stderr
    out = _run(['--gpus', '1'] + ARGS, {'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'}, drop=())
    assert out.returncode != 0 and not _lines(out)


def test_physical_core_count():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
