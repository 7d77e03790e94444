"""GPU tests (-m gpu) of bench.py.
(1) The driver's exact single-GPU command line (`bench.py --gpus 1 --steps 20 --warmup 5`): the timed region must be the
    steady state -- at least 0.3 s of sampling, every unit evaluated once per round -- and the line must carry `roofline`
    and `cpu_baseline` (with the single-core figure and the labelled pystan estimate).
(2) bench.py under the launcher the driver uses for N > 1 (`python -m torch.distributed.run`), with one rank and
    BDRT_BENCH_FORCE_DIST=1 so that the RCCL code path (init_process_group, problem broadcast, barrier, all_reduce of the
    timing / evaluation counts, the sample_sharded round trip, destroy) runs on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    return d


def test_driver_command_line_times_the_steady_state():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--cpu-seconds', '2']
    d = _line(subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900))
    c, r = d['config'], d['roofline']
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['scaling'] == 'weak' and d['dtype'] == 'f64'
    assert c['rounds_per_launch'] >= 1000 and c['units_per_gpu'] == 4096
    assert c['timed_region_s'] >= 0.3, c['timed_region_s']
    assert abs(d['ms_per_step'] * d['steps'] - c['timed_region_s'] * 1e3) < 1e-6 * c['timed_region_s'] * 1e3
    # every chain is active in every round of the timed region: evaluations = units x rounds
    assert c['evals_in_timed_region'] == c['units_per_gpu'] * c['rounds_per_launch'] * d['steps']
    assert abs(d['value'] - c['evals_in_timed_region'] / c['timed_region_s']) <= 1e-9 * d['value']
    assert d['value'] > 5e7
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['launches'] == 20
    assert 'hbm_frac_B1_accounting' not in r
    # the kernel's HIP-event time cannot exceed the wall time of the region that contains it
    assert r['avg_launch_ms'] * r['launches'] <= c['timed_region_s'] * 1e3 * 1.001
    assert c['ranks_seen'] == 1 and c['rccl']['device_ids'] == [0]
    assert 0 < r['mfma_pipe_frac'] < r['frac_executed'] < r['frac'] < 1.1
    # the measured busy fraction of the fp64 pipe rides along from the committed SQ-counter passes (labelled: not collected in this run)
    assert 0 < r['mfma_busy'] < r['pipe_busy'] < 1 and r['pipe_busy_measured'] is False and 'sq_counters' in r['pipe_busy_source']
    fc = c['few_chains']                      # BASELINE config 3 beside the throughput figure, labelled latency-bound
    assert fc['chains'] == 4 and fc['sampler_kind'] == 1 and fc['evals_per_s'] > 2e5 and 'latency' in fc['bound']
    assert 2.0 < fc['us_per_leapfrog_round'] < 20.0
    # beside the headline: fewer chains than the tiles want (which kernel took them: 1 one chain per 512-thread workgroup, 3 one
    # chain per wave), the other shapes of the reference's fits, and config 4 as written (= the headline job at N = 1)
    mo = {m['units']: m for m in c['mid_occupancy']}
    assert set(mo) == {512, 1024, 2048} and mo[1024]['sampler_kind'] == 3 and mo[2048]['sampler_kind'] == 3
    assert mo[512]['evals_per_s'] > 3e7 and mo[1024]['evals_per_s'] > 6e7 and mo[2048]['evals_per_s'] > 9e7
    sh = {(r_['nf'], r_['K']): r_ for r_ in c['shapes']}
    assert set(sh) == {(81, 161), (81, 101), (81, 81), (41, 51), (53, 81), (106, 101)}
    assert all('error' not in r_ and r_['units_4096']['evals_per_s'] > 5e7 for r_ in sh.values())
    assert sh[(81, 161)]['evaluator'] == 4 and sh[(81, 101)]['units_2048']['sampler_kind'] == 3
    assert c['strong_scaling']['value'] == d['value']
    b = d['cpu_baseline']
    assert 'untuned' in b['tuning'] and b['cores'] <= b['logical_cpus']
    assert b['kind'] == 'port' and b['cores'] >= 1 and 0 < b['single_core'] <= b['value']
    assert b['pystan_derived']['value'] > 0 and 'derived' in b['pystan_derived']['source']


def test_bench_contract_under_torchrun_single_rank():
    env = dict(os.environ, BDRT_BENCH_FORCE_DIST='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4',
           '--warmup', '2', '--rounds', '50', '--no-cpu-baseline', '--spectra', '64']
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['scaling'] == 'weak' and d['dtype'] == 'f64'
    assert d['value'] > 1e6 and d['unit'] == 'evals/s' and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['unit'] == 'TFLOP/s'
    assert d['config']['evals_in_timed_region'] == d['config']['units_per_gpu'] * 50 * 4
    rt = d['config']['dist_roundtrip']            # broadcast + short sample_sharded run + summary gather over RCCL
    assert rt['finite'] and rt['ms'] > 0 and rt['gather'] == 'summary'


def test_bench_two_ranks_on_the_one_device():
    """The N = 2 code path of bench.py (unit partition, barriers, max / sum reductions, the sample_sharded round trip) under the
    driver's launcher, both ranks on the box's single GPU over gloo (BDRT_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on one
    device).  Checks the bookkeeping, not a rate."""
    env = dict(os.environ, BDRT_BENCH_ONE_DEVICE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3',
           '--warmup', '1', '--rounds', '40', '--no-cpu-baseline', '--spectra', '32']
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and 'test_mode' in d['config']
    # weak scaling: every rank runs its own 32 spectra x 8 chains; the line counts the evaluations of both
    assert d['config']['units_per_gpu'] == 32 * 8
    assert d['config']['evals_in_timed_region'] == 2 * d['config']['units_per_gpu'] * 40 * 3
    assert d['value'] > 1e5
    rt = d['config']['dist_roundtrip']
    assert 'error' not in rt and rt['finite'] and rt['spectra'] == 16
    # config 4 as written beside the weak-scaling value: the job's spectra IN TOTAL over the ranks (here all 64 of them: a test
    # run has fewer than config 4's 512), timed the same way
    ss = d['config']['strong_scaling']
    assert ss['value'] > 1e5 and ss['unit'] == 'evals/s' and '64 spectra' in ss['workload'] and ss['steps'] >= 2


def test_bare_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts `torch.distributed.run` as a child before any
    GPU call and relays the single line; the line is produced by two ranks (ranks_seen) or not at all."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['BDRT_BENCH_ONE_DEVICE'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--rounds', '30',
           '--no-cpu-baseline', '--spectra', '16']
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['config']['rccl']['device_ids'] == [0, 0]
    assert 'bench.py itself' in d['config']['rccl']['launched_by']
    assert d['config']['evals_in_timed_region'] == 2 * 16 * 8 * 30 * 2
    # without the test switch a 1-GPU box must refuse: fewer devices than ranks
    env.pop('BDRT_BENCH_ONE_DEVICE')
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith('{')]


@pytest.mark.timeout(1800)
def test_bench_eight_ranks_on_the_one_device():
    """`bench.py --gpus 8` under the driver's launcher with all eight ranks on the box's single GPU over gloo: the 8-way partition
    of the weak-scaling job (8 spectra x 8 chains = 64 units per rank), the `strong_scaling` leg (the same 64 spectra in total: 64
    units per rank, one chain per workgroup) and the sharded round trip over 8 ranks.  Bookkeeping, not a rate."""
    env = dict(os.environ, BDRT_BENCH_ONE_DEVICE='1', BDRT_BENCH_ROUNDTRIP_TIMEOUT='600')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
           '127.0.0.1', '--master-port', '29547', os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2',
           '--warmup', '1', '--rounds', '30', '--no-cpu-baseline', '--spectra', '8']
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500))
    c = d['config']
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and 'test_mode' in c
    assert c['ranks_seen'] == 8 and c['rccl']['device_ids'] == [0] * 8
    assert c['units_per_gpu'] == 64 and c['evals_in_timed_region'] == 8 * 64 * 30 * 2
    ss = c['strong_scaling']
    assert '64 spectra' in ss['workload'] and '(64 units per GPU)' in ss['workload'] and ss['sampler_kind_rank0'] == 1
    assert ss['value'] > 1e4 and ss['steps'] >= 2
    rt = c['dist_roundtrip']
    assert 'error' not in rt and rt['finite'] and rt['spectra'] == 64
