"""GPU test (-m gpu): bench.py end to end under the launcher the driver uses for N > 1 (`python -m torch.distributed.run`),
with one rank and BDRT_BENCH_FORCE_DIST=1 so that the RCCL code path (init_process_group, barrier, all_reduce of the
timing / evaluation counts, destroy) runs on a 1-GPU box; checks the one-line JSON contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_contract_under_torchrun_single_rank():
    env = dict(os.environ, BDRT_BENCH_FORCE_DIST='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '100',
           '--warmup', '50', '--no-cpu-baseline', '--spectra', '64']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 100 and d['warmup'] == 50 and d['scaling'] == 'weak' and d['dtype'] == 'f64'
    assert d['value'] > 1e6 and d['unit'] == 'evals/s' and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['unit'] == 'TFLOP/s'
    # evaluations counted = units x steps (every chain is active in every round of the timed region)
    assert d['config']['evals_in_timed_region'] == d['config']['units_per_gpu'] * 100
