#!/usr/bin/env python3
"""bench.py -- HMC log-posterior+gradient evaluations per second, 81 frequencies x 161 tau-basis (BASELINE.json).

Workload (config.workload): BASELINE config 4, weak scaling -- every GPU holds its own shard of 512 synthetic 2-ZARC
spectra (shared frequency / tau grids => one A/L set in HBM) with 8 NUTS chains per spectrum = 4096 units per GPU = 16
chains on each of the 256 CUs.  With N GPUs the job is 512*N spectra, partitioned by `parallel.partition_units`
(whole spectra per rank -- the function `parallel.sample_sharded` uses), so rank r samples spectra [512 r, 512 (r+1)).
Model Series_pos, D = 331, sampling-mode hyper-parameters.  Each unit is a real NUTS chain (device-resident
transitions, bdrt_nuts.hip).  No data-path collective: units are independent.

A STEP is one launch of the sampler kernel that advances every chain by `--rounds` (default 1000) leapfrogs: one
log-posterior+gradient evaluation per unit and round inside the chain's current tree.  With the driver's
`--steps 20 --warmup 5` the timed region is 20 000 rounds (~0.7 s) after 5 000 warm-up rounds (initial points, step-size
search and the first transitions are behind every chain).

Prints ONE JSON line (rank 0).  `value` = gradient evaluations actually executed by all ranks during the timed
region (device counter) / wall time (max over ranks).  `roofline` is for the dominant kernel (nuts_kernel):
achieved = algorithmic FLOPs per launch (4.27e5 per evaluation) / HIP-event launch duration on the
sampler's stream.  `cpu_baseline` = the CPU oracle (plain C port of the same log-posterior+gradient, -O3 -march=native)
inside the oracle's NUTS driver, timed on this box's host cores: one core, and one chain per core on all cores.
N > 1 additionally times, outside the steady-state region, one complete `parallel.sample_sharded` call (broadcast of
the problem + short run + gather of per-spectrum summaries over RCCL) and reports it in `config.dist_roundtrip`.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 4.27e5      # 4*(2Nf*K + 3K^2) + ~12k element-wise, Nf=81, K=161
# what the structured path really executes per evaluation on a log-uniform grid: the A GEMMs (820 MFMA * 2048 flop / 16 chains =
# 1.05e5 with the operands from the LDS table, 1.18e5 on padded 176 x 168 tiles with streamed fragments) + six banded convolutions
# + ~1.5e4 element-wise; main() picks the figure of the evaluator the problem was given
FLOP_PER_EVAL_EXECUTED = 1.66e5 - (924 - 820) * 2048 / 16.0 + 1.3e3
PEAK_F64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak; measured 78.05 by tools/mfma_probe (profiles/)
NF, K = 81, 161
N_SPECTRA, CHAINS_PER_SPECTRUM = 512, 8
ROUNDS_PER_STEP = 1000
# BASELINE.md section 1 / SURVEY section 6: derived from the reference's notebook (4 chains x 1000 iterations in 515 s at
# K = 81 => 6-8 k evals/s over 4 processes), scaled by the dense work ratio to K = 161.  NOT measured here: pystan
# cannot be installed on this box.
PYSTAN_DERIVED = {'value': 2250.0, 'unit': 'evals/s', 'processes': 4,
                  'source': 'derived from reference notebook timings (Run fits.ipynb cell 6: 4x1000 iterations, 515 s, '
                            'K=81), scaled to K=161; not measured on this box'}


# what the CPU figure is: the parity checker, compiled -O3 -march=native, not a tuned CPU implementation (scalar dense
# mat-vecs, buffers allocated per evaluation, no use of the Toeplitz / band structure the GPU path exploits).  The GPU/CPU
# ratio is context, not a quality measure -- roofline.frac is.
CPU_PORT_NOTE = ('untuned checker: oracle/bdrt_oracle.c is the plain-C parity oracle (scalar dense mat-vecs, per-evaluation '
                 'malloc, no Toeplitz/band shortcuts), one chain per physical core')
# MFMA instructions the structured path issues per evaluation of a 16-chain tile (v_mfma_f64_16x16x4_f64, 2048 flop each; DESIGN
# 3.1b; SQ_INSTS_MFMA in profiles/ agrees): with the packed A fragments streamed from L2 (evaluator 2) forward A (11 row tiles x 42
# k-steps) + backward A^T (11 x 42) = 924; with the operands from the LDS-resident Toeplitz table (evaluator 4: this workload) ten
# tiles x 41 chunks each way = 820, the odd rows being ~1.3e3 flop of VALU dot products
MFMA_PER_TILE = {2: 924, 4: 820}
MFMA_FLOP_PER_EVAL = MFMA_PER_TILE[4] * 2048 / 16.0


def synth_spectra(n, seed=20260101, raw=False):
    """n two-ZARC spectra, parameters per SURVEY 8(d) config 4; noise models cycled uniform/Orazem/Macdonald 0.25 %.
    raw=False: [n, 2 NF] stacked (Z', Z'') scaled as Inverter._scale_Z does (the oracle leg's input, which must not depend on the
    product package); raw=True: the complex spectra as measured, for `Inverter.batch_stan_data` / `fit_many`."""
    rs = np.random.RandomState(seed)
    f = np.logspace(6, -2, NF)
    w = 2 * np.pi * f
    Z = np.empty((n, 2 * NF))
    Zraw = []
    for s in range(n):
        R1, R2 = rs.uniform(0.5, 2, 2)
        t1, t2 = 10 ** rs.uniform(-3, -1), 10 ** rs.uniform(-4, -2)
        n1, n2 = rs.uniform(0.6, 0.95, 2)
        z = 1.0 + R1 / (1 + (1j * w * t1) ** n1) + R2 / (1 + (1j * w * t2) ** n2)
        kind = s % 3
        if kind == 0:
            sig = 0.0025 * (z.real.max() - z.real.min()) * np.ones(NF)
        elif kind == 1:
            sig = 0.0025 * (np.abs(z.real) + np.abs(z.imag))
        else:
            sig = 0.0025 * np.abs(z)
        noise = rs.normal(size=(NF, 2))
        z = z + sig * noise[:, 0] + 1j * sig * noise[:, 1]
        Zraw.append(z)
        scale = np.std(np.abs(z)) / np.sqrt(NF / 81)       # Inverter._scale_Z (inversion.py:2437-2441)
        z = z / scale
        Z[s] = np.concatenate([z.real, z.imag])
    return (f, Zraw) if raw else (f, Z)


_CPU_WORKER = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as orc
from bench import synth_spectra, NF, K
f, Z = synth_spectra(1)
tau = 1 / (2 * np.pi * np.logspace(10, -6, K)); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([orc.construct_A(f, 'real', tau=tau, epsilon=eps), orc.construct_A(f, 'imag', tau=tau, epsilon=eps)])
blk = dict(A=A, L0=orc.construct_L(tau, eps, 0), L1=orc.construct_L(tau, eps, 1), L2=0.75 * orc.construct_L(tau, eps, 2), nonneg=True)
m = orc.OracleModel([blk], Z[0], f, ups_alpha=1.0, ups_beta=0.1)
ctrl = orc.nuts_control(adapt_delta=0.9, adapt_t0=10.0)
chain, budget = int(sys.argv[1]), float(sys.argv[2])
n = 0; t0 = time.perf_counter(); it = 6
# the oracle's own NUTS driver (oracle/nuts_oracle.c), warm-up transitions of a real chain; repeated with a fresh chain id
# and a doubled iteration count until the time budget is used
while time.perf_counter() - t0 < budget:
    _, _, d = orc.nuts_sample(m, chain, 1234, it, 0, control=ctrl)
    n += d['n_leapfrog']; chain += 1000; it = min(2 * it, 48)
print(n, time.perf_counter() - t0)
''' % ROOT


_CPU_TUNED_WORKER = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as orc
from bench import synth_spectra, NF, K
f, Z = synth_spectra(1)
tau = 1 / (2 * np.pi * np.logspace(10, -6, K)); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([orc.construct_A(f, 'real', tau=tau, epsilon=eps), orc.construct_A(f, 'imag', tau=tau, epsilon=eps)])
blk = dict(A=A, L0=orc.construct_L(tau, eps, 0), L1=orc.construct_L(tau, eps, 1), L2=0.75 * orc.construct_L(tau, eps, 2), nonneg=True)
banded, budget = int(sys.argv[1]) != 0, float(sys.argv[2])
m = orc.TunedS1(blk, Z[0], f, ups_alpha=1.0, ups_beta=0.1, banded=banded)
assert m.banded == banded
th = np.random.default_rng(int(sys.argv[3])).uniform(-1, 1, m.D)
n = 0; chunk = 200; m.bench(th, 50)
t0 = time.perf_counter()
while time.perf_counter() - t0 < budget:
    m.bench(th, chunk); n += chunk
print(n, time.perf_counter() - t0)
''' % ROOT


def _cpu_tuned_leg(nproc, seconds, banded):
    procs = [subprocess.Popen([sys.executable, '-c', _CPU_TUNED_WORKER, str(int(banded)), str(seconds), str(i)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, cwd=ROOT) for i in range(nproc)]
    total = 0.0
    for p in procs:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('cpu_baseline.tuned worker failed: ' + err.decode()[-400:])
        n, t = out.decode().split()
        total += float(n) / float(t)
    return total


def _cpu_leg(nproc, seconds):
    env = dict(os.environ, BDRT_ORACLE_NATIVE='1')
    procs = [subprocess.Popen([sys.executable, '-c', _CPU_WORKER, str(i), str(seconds)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, cwd=ROOT, env=env) for i in range(nproc)]
    total = 0.0
    for p in procs:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('cpu_baseline worker failed: ' + err.decode()[-400:])
        n, t = out.decode().split()
        total += float(n) / float(t)
    return total


def physical_cores():
    """Physical cores this process may run on (hyper-thread siblings counted once)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    for cpu in allowed:
        try:
            with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % cpu) as fh:
                cores.add(fh.read().strip())
        except OSError:
            cores.add(str(cpu))
    n = max(1, len(cores))
    # a container may see every CPU of the host and still be throttled to a few cores' worth of time: the CPU bandwidth
    # quota of the cgroup bounds what any number of processes can use
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: (t.split()[0], t.split()[1])),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: (t.strip(), None))):
        try:
            with open(path) as fh:
                quota, period = parse(fh.read())
            if period is None:
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh:
                    period = fh.read().strip()
            if quota not in ('max', '-1') and float(quota) > 0:
                n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(seconds=8.0):
    """Oracle log_prob+grad evaluations/s inside the oracle's NUTS driver on the host cores: one core, then one chain
    per core (pystan's own layout: one process per chain)."""
    from oracle import oracle as orc
    orc.build(force=True, native=True)           # -O3 -march=native for THIS host (the checker build is untouched)
    ncores = physical_cores()
    single = _cpu_leg(1, seconds)
    allc = _cpu_leg(ncores, seconds)
    # a box that grants fewer cores than it shows (no readable quota) gives itself away by the aggregate: report the
    # parallelism actually obtained beside the process count
    effective = allc / single if single > 0 else float(ncores)
    # the tuned leg (oracle/bdrt_tuned.c: the same formulas from a preallocated workspace, vectorised products): raw
    # log-posterior + gradient evaluations, dense (SURVEY 8(d)'s accounting: A, A^T and the three K x K penalty operators as full
    # products) and banded (the penalty operators through their 13 diagonals: what the GPU's structured path executes)
    tuned = None
    try:
        orc.tuned_lib(force=True)
        ts = min(seconds, 4.0)
        tuned = dict(unit='evals/s', cores=ncores,
                     dense=dict(single_core=_cpu_tuned_leg(1, ts, False), value=_cpu_tuned_leg(ncores, ts, False)),
                     banded=dict(single_core=_cpu_tuned_leg(1, ts, True), value=_cpu_tuned_leg(ncores, ts, True)),
                     sample='raw log-posterior + gradient evaluations of oracle/bdrt_tuned.c (Series_pos 81x161, jacobian on; checked against the '
                            'oracle by tests/test_oracle_tuned.py), gcc -O3 -march=native -ffp-contract=fast: 1 process, then %d processes, %.0f s each; '
                            'NOT inside a NUTS driver (a leapfrog is one evaluation plus O(D) vector work)' % (ncores, ts))
    except Exception as e:          # noqa: BLE001 -- the tuned figure is context, the line must not depend on it
        tuned = dict(error=str(e)[-200:])
    model = ''
    try:
        with open('/proc/cpuinfo') as fh:
            model = next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), '')
    except OSError:
        pass
    return dict(value=allc, unit='evals/s', cores=ncores, kind='port', tuning=CPU_PORT_NOTE, single_core=single,
                cpu_model=model, logical_cpus=os.cpu_count(), effective_parallelism=effective, pystan_derived=PYSTAN_DERIVED, tuned=tuned,
                sample='leapfrogs of real NUTS warm-up transitions (oracle/nuts_oracle.c driving oracle/bdrt_oracle.c, '
                       'Series_pos 81x161, jacobian on), gcc -O3 -march=native: 1 process x %.0f s (single_core), then %d '
                       'processes x %.0f s, one chain per core (value)' % (seconds, ncores, seconds))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


WATCHDOG_EXIT = 4      # exit code of a run whose rate line is valid but whose multi-rank round trip / teardown hung


def self_launch(n, argv):
    """`bench.py --gpus N` started without a launcher: run the N ranks as a child `python -m torch.distributed.run` (the
    command the driver itself uses), relay its output and return its exit code.  Refuses (non-zero) when the box has fewer
    than N devices, unless BDRT_BENCH_ONE_DEVICE (test mode) is set."""
    if not os.environ.get('BDRT_BENCH_ONE_DEVICE'):
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:
            have = 0
        if have < n:
            sys.stderr.write('bench.py: --gpus %d asked for, %d device(s) visible; not launching\n' % (n, have))
            return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, BDRT_BENCH_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    child = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    for ln in child.stdout.splitlines():
        if ln not in lines:
            sys.stderr.write(ln + '\n')
    if child.returncode != 0 and len(lines) == 1 and '"timed out; the rate above was measured before it"' in lines[0]:
        # a rank's watchdog ended the job (exit code WATCHDOG_EXIT; torchrun reports its own non-zero code): the rate in the line
        # was measured before the sharded round trip hung -- relay the line, and a non-zero code so that this is never a success
        sys.stderr.write('bench.py: the sharded round trip (or the teardown) hung; launcher exited %d; the rate was measured before it\n'
                         % child.returncode)
        print(lines[0])
        return WATCHDOG_EXIT
    if child.returncode != 0 or len(lines) != 1:
        sys.stderr.write('bench.py: launcher exited %d with %d result line(s)\n' % (child.returncode, len(lines)))
        return child.returncode or 3
    print(lines[0])
    return 0


def build_problem_kwargs(n_spectra, lib=None):
    """Global problem of the benchmark: shared 81 x 161 grids, n_spectra synthetic spectra -- the batch that
    `Inverter(basis_freq=...).fit_many(f, spectra, nonneg=True, mode='sample')` runs, taken from the Inverter itself (matrices built
    on the GPU, every spectrum scaled as `fit` scales it, sample-mode hyper-parameters of `_prep_stan_data`, inversion.py:1725-1754)."""
    from bayes_drt_amd import engine
    from bayes_drt_amd.inversion import Inverter
    f, spectra = synth_spectra(n_spectra, raw=True)
    model_name, dat = Inverter(basis_freq=np.logspace(10, -6, K)).batch_stan_data(f, spectra, nonneg=True, mode='sample')
    assert model_name == 'Series_pos_StanModel.pkl'
    blocks, kw, _ = engine.blocks_from_dat(model_name, dat)
    kw = {k: v for k, v in kw.items() if k in ('sigma_min', 'ups_alpha', 'ups_beta', 'induc_scale')}
    return dict(kw, blocks=blocks, Z=np.asarray(dat['Z'], dtype=float), freq=np.asarray(dat['freq'], dtype=float))


SHAPES = ((81, 161), (81, 101), (81, 81), (41, 51), (53, 81), (106, 101))       # (frequencies, basis functions): the headline, the
# package default for an 81-point spectrum (inversion.py:2191-2197) and the other shapes of the reference's stored fits


def shape_problem_kwargs(nf, k, n_spectra, seed=7):
    """A single-DRT problem on ten-points-per-decade grids with nf frequencies and k basis functions (the basis grid contains the
    measurement grid or lies inside it, so that A is exactly Toeplitz as for the reference's default grids) and n_spectra
    synthetic two-ZARC spectra."""
    from bayes_drt_amd import matrices as gm
    f = np.logspace(6, 6 - (nf - 1) / 10.0, nf)
    half = (k - nf) // 2
    top = 6 + half / 10.0
    basis_freq = np.logspace(top, top - (k - 1) / 10.0, k)
    tau = 1 / (2 * np.pi * basis_freq)
    eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(basis_freq, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    rs = np.random.RandomState(seed)
    w = 2 * np.pi * f
    Z = np.empty((n_spectra, 2 * nf))
    for i in range(n_spectra):
        R1, R2 = rs.uniform(0.5, 2, 2)
        t1, t2 = 10 ** rs.uniform(-3, -1), 10 ** rs.uniform(-4, -2)
        n1, n2 = rs.uniform(0.6, 0.95, 2)
        z = 1.0 + R1 / (1 + (1j * w * t1) ** n1) + R2 / (1 + (1j * w * t2) ** n2)
        z = z + 0.0025 * (z.real.max() - z.real.min()) * (rs.normal(size=nf) + 1j * rs.normal(size=nf))
        z = z / (np.std(np.abs(z)) / np.sqrt(nf / 81))
        Z[i] = np.concatenate([z.real, z.imag])
    blk = dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)
    return dict(blocks=[blk], Z=Z, freq=f, sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)


def steady_rate(prob, n_units, ctrl, n_spectra, rounds=500, launches=4, warm_rounds=600):
    """evals/s of n_units real NUTS chains in their warm-up (no chain finishes), after warm_rounds untimed rounds: the short form
    of the headline measurement, for the tables that accompany it (outside the timed region)."""
    from bayes_drt_amd.engine import Sampler
    spec = (np.arange(n_units) % n_spectra).astype(np.int32)
    with Sampler(prob, n_units, 1000000, 1, 1234, ctrl, spec=spec) as smp:
        smp.advance(warm_rounds); smp.sync()
        n0 = smp.total_leapfrogs(); t0 = time.perf_counter()
        for _ in range(launches):
            smp.advance(rounds)
        smp.sync()
        dt = time.perf_counter() - t0
        return (smp.total_leapfrogs() - n0) / dt, smp.kind()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20, help='timed launches of --rounds leapfrog rounds each')
    ap.add_argument('--warmup', type=int, default=5, help='untimed launches before the timed region')
    ap.add_argument('--rounds', type=int, default=ROUNDS_PER_STEP, help='leapfrog rounds per launch (= per step)')
    ap.add_argument('--spectra', type=int, default=N_SPECTRA, help='spectra per GPU')
    ap.add_argument('--chains', type=int, default=CHAINS_PER_SPECTRUM)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=8.0)
    ap.add_argument('--phase-profile', action='store_true', help='print the in-kernel cycle breakdown (perturbs timing)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'RANK' not in os.environ:
        # started bare: become the launcher.  Nothing in this process has touched the GPU yet (device_count() does not
        # initialise HIP on this image), and the ranks are a CHILD process, never an exec.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        # one process per GPU (pystan: one process per chain, inversion.py:1218-1221): a line that says n_gpus = N is
        # printed by N ranks or not at all
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d; refusing to report n_gpus from a different number of '
                         'ranks' % (args.gpus, world))

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU: workers are plain subprocesses
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product has no CPU path')
    # BDRT_BENCH_ONE_DEVICE=1 (test only): every rank on device 0 with a gloo group -- RCCL refuses two ranks on one device --
    # so that a 1-GPU box can run the N > 1 code path (partition, barriers, reductions, the sharded round trip).  The line is
    # marked; its value says nothing about scaling.
    one_device = bool(os.environ.get('BDRT_BENCH_ONE_DEVICE'))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # BDRT_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank too (lets a 1-GPU box exercise it)
    use_dist = world > 1 or (bool(os.environ.get('BDRT_BENCH_FORCE_DIST')) and 'RANK' in os.environ)
    if use_dist:
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    red_dev = 'cpu' if one_device else 'cuda'

    from bayes_drt_amd import _lib, parallel as par
    from bayes_drt_amd._lib import NutsControl, check
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    lib = _lib.require_gpu()
    check(lib.bdrt_set_device(local_rank), 'bdrt_set_device')

    # ---- the job: args.spectra * world spectra, partitioned like parallel.sample_sharded does (whole spectra per rank) ----
    n_global = args.spectra * world
    t_setup0 = time.perf_counter()
    if use_dist:
        kw = par._unpack_problem(par.broadcast_arrays(par._pack_problem(build_problem_kwargs(n_global)) if rank == 0 else None, src=0))
    else:
        kw = build_problem_kwargs(n_global)
    spec, chain = par.make_units(n_global, args.chains)
    u0, u1 = par.partition_units(n_global, args.chains, world)[rank]
    s0, s1 = int(spec[u0]), int(spec[u1 - 1]) + 1
    local_kw = dict(kw); blocks = local_kw.pop('blocks'); Zall = local_kw.pop('Z'); freq = local_kw.pop('freq')
    prob = Problem(blocks, np.atleast_2d(Zall)[s0:s1], freq, **local_kw)
    assert prob.D == 2 * K + 9
    evaluator = prob.evaluator()                          # 4: GEMM operands from the LDS-resident Toeplitz table, 2: streamed fragments
    mfma_flop = MFMA_PER_TILE.get(evaluator, 924) * 2048 / 16.0
    flop_executed = 1.66e5 - (924 - MFMA_PER_TILE.get(evaluator, 924)) * 2048 / 16.0 + (1.3e3 if evaluator == 4 else 0.0)
    n_units = u1 - u0
    ctrl = NutsControl()
    lib.bdrt_nuts_defaults(C.byref(ctrl))
    ctrl.adapt_delta, ctrl.adapt_t0 = 0.9, 10.0                          # inversion.py:1221
    # warm-up long enough that no chain finishes inside the benchmark (adaptation windows at 100, 150, 250, ...)
    smp = Sampler(prob, n_units, 1000000, 1, 1234, ctrl, spec=spec[u0:u1] - s0, chain_ids=chain[u0:u1])
    setup_ms = (time.perf_counter() - t_setup0) * 1e3

    def sync_all():
        smp.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        smp.advance(args.rounds)
    sync_all()
    if args.phase_profile:
        check(lib.bdrt_sampler_phase_profile(smp.handle, 1, None), 'phase_profile')
    smp.kernel_time(reset=True)
    n0 = smp.total_leapfrogs()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        smp.advance(args.rounds)
    smp.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    n1 = smp.total_leapfrogs()
    ms_total, launches = smp.kernel_time()
    evals = float(n1 - n0)
    if args.phase_profile and rank == 0:
        cyc = (C.c_longlong * 32)()
        check(lib.bdrt_sampler_phase_profile(smp.handle, 0, cyc), 'phase_profile')
        if smp.kind() == 1:          # the one-chain-per-workgroup kernel (bdrt_solo.h) has its own slots
            for k, nm in enumerate(['eval E0 constrain', 'eval E1 A.x + prior', 'eval E2 likelihood + L^T w', 'eval E3 A^T g',
                                    'eval E4 chain rule', 'nuts C kick + kinetic', 'nuts S1 scalar logic', 'nuts D tree / transition end',
                                    "nuts A'/E next point"]):
                print('SOLO %-30s %9.0f cycles/round' % (nm, cyc[k] / n_units / (args.steps * args.rounds)), file=sys.stderr)
        names = ['tile:scalars', 'tile:x', 'tile:gemmA', 'tile:Zacc', 'tile:likelihood', 'tile:x2', 'tile:gemmL',
                 'tile:prior', 'tile:gemmBwd', 'tile:epilogue', 'nuts:A kick-drift', 'nuts:C kick+kin', 'nuts:S1',
                 'nuts:D tree', 'nuts:S2', 'nuts:E next', 'nuts:S3']
        tot = float(sum(cyc[:17])) or 1.0
        cpw = min(16, max(1, -(-n_units // 256)))          # chains per workgroup (bdrt_sampler_create)
        n_wg = -(-n_units // cpw)
        rounds_total = args.steps * args.rounds
        for k, nm in enumerate(names):
            print('PHASE %-20s %6.2f %%  %9.0f cycles/round' % (nm, 100 * cyc[k] / tot, cyc[k] / n_wg / rounds_total), file=sys.stderr)
        if os.environ.get('BDRT_BENCH_FINE') == '1':      # a library built with -DBDRT_PROF_FINE=1 (tools/build_variant.sh)
            fn = ['C.a state + level-0 request', 'C.b gradient + kick', 'C.c sums', 'S1 scalar logic', 'D.a proposal copy', 'D.b merge level 0',
                  'D.c merges of levels >= 1', 'D.d checkpoint store', 'D.e subtree end / leaf index', 'D.f transition end', "A' kick + drift",
                  'Z normals ahead', 'E next start point']
            for k, nm in enumerate(fn):
                print('FINE %-32s %7.0f cycles/round (wave average)' % (nm, cyc[k] / n_wg / rounds_total / 8), file=sys.stderr)
            for k, nm in zip((14, 15, 16), ('proposal copy', 'two levels or more', 'subtree end')):
                print('FINE waves per round with %-20s %6.3f of 8' % (nm, cyc[k] / n_wg / rounds_total), file=sys.stderr)
        wnames = ['tile', 'C', 'S1', 'D', "A'", "E+S3+A''", 'end-of-round barrier wait']
        for k, nm in enumerate(wnames):
            print('WAVE-AVG %-26s %9.0f cycles/round' % (nm, cyc[17 + k] / n_wg / rounds_total / 8), file=sys.stderr)
        for k, nm in enumerate(['B1 (X ready)', 'B2 (A x ready)', 'B3 (g ready)', 'B4 (A^T g ready)']):
            print('WAVE-AVG wait at barrier %-17s %6.0f cycles/round' % (nm, cyc[25 + k] / n_wg / rounds_total / 8), file=sys.stderr)
        for k, nm in enumerate(['entry -> B1 (P1, prior chain)', 'B1 -> B2 (forward GEMM)', 'B2 -> B3 (likelihood)']):
            print('WAVE-AVG tile %-30s %6.0f cycles/round' % (nm, cyc[29 + k] / n_wg / rounds_total / 8), file=sys.stderr)
        print('half-waves per round in stage E (new start point): %.2f of 16' % (cyc[24] / n_wg / rounds_total * 2), file=sys.stderr)
    smp.close()

    # ---- strong scaling: BASELINE config 4 AS WRITTEN -- 512 spectra x 8 chains in total, 512 / N spectra per rank -- timed the
    # same way (barrier + synchronize on both sides, max over ranks), beside the weak-scaling value above.  At N = 1 it is the
    # same job as the headline and is not run twice.
    strong = None
    if world > 1:
        ns_total = min(N_SPECTRA, n_global)               # (a test run with fewer spectra than config 4 has: all of them)
        su0, su1 = par.partition_units(ns_total, args.chains, world)[rank]
        sspec, schain = par.make_units(ns_total, args.chains)
        n_strong = su1 - su0
        if n_strong > 0:
            ss0, ss1 = int(sspec[su0]), int(sspec[su1 - 1]) + 1
            # this rank's rows of the global Z: the first 512 spectra are partitioned again, a rank samples those it holds a copy of
            sprob = Problem(blocks, np.atleast_2d(Zall)[ss0:ss1], freq, **local_kw)
            ssmp = Sampler(sprob, n_strong, 1000000, 1, 1234, ctrl, spec=sspec[su0:su1] - ss0, chain_ids=schain[su0:su1])
        steps_s = max(2, min(args.steps, 10))

        def sync_strong():
            if n_strong > 0:
                ssmp.sync()
            torch.cuda.synchronize()
            dist.barrier()
        for _ in range(max(1, min(args.warmup, 3))):
            if n_strong > 0:
                ssmp.advance(args.rounds)
        sync_strong()
        m0 = ssmp.total_leapfrogs() if n_strong > 0 else 0
        sync_strong()
        ts0 = time.perf_counter()
        for _ in range(steps_s):
            if n_strong > 0:
                ssmp.advance(args.rounds)
        sync_strong()
        ts1 = time.perf_counter() - ts0
        sev = float((ssmp.total_leapfrogs() - m0) if n_strong > 0 else 0)
        tt = torch.tensor([ts1], dtype=torch.float64, device=red_dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        te = torch.tensor([sev], dtype=torch.float64, device=red_dev); dist.all_reduce(te, op=dist.ReduceOp.SUM)
        strong = {'workload': 'BASELINE config 4 as written: %d spectra x %d chains IN TOTAL (%d units per GPU)' % (ns_total, args.chains, ns_total * args.chains // world),
                  'value': float(te.item() / tt.item()), 'unit': 'evals/s', 'steps': steps_s, 'ms_per_step': float(tt.item()) * 1e3 / steps_s,
                  'sampler_kind_rank0': ssmp.kind() if n_strong > 0 else None}
        if n_strong > 0:
            ssmp.close(); sprob.close()

    # ---- fewer chains than the tiles want (the regime of 8-GPU runs of config 4, of config 3 x many spectra, of a run's tail) and
    # the other shapes of the reference's fits: steady-state rates of the same kind, outside the timed region (rank 0, N = 1)
    mid, shapes = None, None
    if rank == 0 and args.gpus == 1 and not os.environ.get('BDRT_BENCH_NO_TABLES'):
        mid = []
        for nu in (512, 1024, 2048):
            r, kind = steady_rate(prob, nu, ctrl, prob.n_spectra)
            mid.append({'units': nu, 'evals_per_s': r, 'sampler_kind': kind})
        shapes = []
        for nf_s, k_s in SHAPES:
            try:
                sp_ = Problem(**{**shape_problem_kwargs(nf_s, k_s, 64)})
            except Exception as exc:
                shapes.append({'nf': nf_s, 'K': k_s, 'error': str(exc)[:120]})
                continue
            e_s = 4 * (2 * nf_s * k_s + 3 * k_s * k_s)              # dense-formulation flop per evaluation
            row = {'nf': nf_s, 'K': k_s, 'evaluator': sp_.evaluator(), 'flop_per_eval_algorithmic': e_s}
            for nu in (4096, 2048):
                r, kind = steady_rate(sp_, nu, ctrl, 64)
                row['units_%d' % nu] = {'evals_per_s': r, 'sampler_kind': kind, 'frac': r * e_s / 1e12 / PEAK_F64_MFMA_TFLOPS}
            shapes.append(row)
            sp_.close()

    # ---- raw log-posterior+gradient kernel at B = 1 .. 4096 points, inputs resident in HBM; outside the timed region
    sweep = None
    if rank == 0 and args.gpus == 1:
        sweep = []
        ts = torch.cuda.Stream()
        for B in (1, 4, 32, 512, 4096):
            th = torch.empty(B, prob.D, dtype=torch.float64, device='cuda').uniform_(-2, 2)
            gr = torch.empty_like(th); lpv = torch.empty(B, dtype=torch.float64, device='cuda')
            sp = torch.randint(0, prob.n_spectra, (B,), dtype=torch.int32, device='cuda')
            torch.cuda.synchronize()
            with torch.cuda.stream(ts):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                for it in range(13):
                    if it == 3:
                        e0.record(ts)
                    check(lib.bdrt_logp_grad_dev(prob.handle, th.data_ptr(), sp.data_ptr(), B, 1, lpv.data_ptr(), gr.data_ptr(),
                                                 ts.cuda_stream), 'bdrt_logp_grad_dev')
                e1.record(ts)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            sweep.append({'B': B, 'us_per_launch': us, 'evals_per_s': B / us * 1e6})

    # ---- BASELINE config 3 (the reference's own call shape: 4 chains of one spectrum): latency-bound, reported beside the
    # throughput figure as SURVEY 8(d) asks.  One-chain-per-workgroup kernel (bdrt_solo.h); outside the timed region.
    few = None
    if rank == 0 and args.gpus == 1:
        p1 = Problem(blocks, np.atleast_2d(Zall)[:1], freq, **local_kw)
        s4 = Sampler(p1, 4, 1000000, 1, 1234, ctrl)
        s4.advance(2000); s4.sync()
        m0 = s4.total_leapfrogs(); s4.kernel_time(reset=True)
        tf = time.perf_counter()
        for _ in range(10):
            s4.advance(1000)
        s4.sync()
        tf = time.perf_counter() - tf
        m1 = s4.total_leapfrogs(); kms, kl = s4.kernel_time()
        few = {'workload': 'config3: 1 spectrum x 4 NUTS chains, Series_pos 81x161', 'chains': 4,
               'evals_per_s': (m1 - m0) / tf, 'us_per_leapfrog_round': kms * 1e3 / max(kl, 1) / 1000,
               'sampler_kind': s4.kind(), 'bound': 'latency-bound: 4 of 256 CUs busy, one dependent evaluation per round'}
        s4.close()
        p1.close()

    ranks_seen, devices_seen, backend = 1, [local_rank], 'none (single process)'
    if use_dist:
        # who actually took part: every rank contributes 1 and its device index
        one = torch.ones(1, dtype=torch.float64, device=red_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(one.item()))
        mine = torch.tensor([float(torch.cuda.current_device())], dtype=torch.float64, device=red_dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        devices_seen = [int(g.item()) for g in got]
        backend = dist.get_backend()
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit('bench.py: %d ranks answered, --gpus %d' % (ranks_seen, args.gpus))
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([evals], dtype=torch.float64, device=red_dev)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        elapsed, evals = float(t.item()), float(e.item())

    if rank == 0:
        value = evals / elapsed
        launches = max(launches, 1)
        avg_ms = ms_total / launches
        evals_per_launch = (n1 - n0) / launches                       # this rank's launches
        achieved = evals_per_launch * FLOP_PER_EVAL / (avg_ms * 1e-3) / 1e12
        # HBM bytes per launch of the sampler kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
        # collected separately, gfx950 correction applied: tools/profile_bench.sh), scaled to this run's launch length
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc) and n_units == N_SPECTRA * CHAINS_PER_SPECTRUM:
            try:
                pj = json.load(open(pmc))
                per_round = pj.get('hbm_bytes_per_round_corrected')
                traffic = per_round * args.rounds if per_round else None
                traffic_src = ('profiles/pmc_traffic.json (%s), bytes per round x %d rounds; not collected in this run -- PMC '
                               'passes need rocprofv3 around the process' % (pj.get('source', 'rocprofv3 --pmc passes'), args.rounds))
            except Exception:
                traffic = None
        # measured busy fraction of the fp64 pipe ((MFMA busy cycles + 4 x VALU instructions) / SIMD cycles) from the committed SQ-counter
        # passes of this command (profiles/sq_counters.json, tools/profile_sq.sh + tools/make_sq_json.py): like `traffic`, PMC counters
        # need rocprofv3 around the process, so the figure is that run's, labelled as such
        pipe_busy, pipe_src, mfma_busy = None, None, None
        sqj = os.path.join(ROOT, 'profiles', 'sq_counters.json')
        if os.path.exists(sqj) and n_units == N_SPECTRA * CHAINS_PER_SPECTRUM:
            try:
                sj = json.load(open(sqj))
                pipe_busy, mfma_busy = sj.get('pipe_busy'), sj.get('mfma_busy_frac')
                pipe_src = '%s: launch %.3f ms there, %.3f ms here; not collected in this run' % (sj.get('source'), sj.get('avg_launch_ms', 0.0), avg_ms)
            except Exception:
                pipe_busy = None
        line = {
            'metric': 'HMC log-posterior+grad evals/sec (81 freq x 161 tau-basis)',
            'value': value, 'unit': 'evals/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config4-shard: %d synthetic 2-ZARC spectra x %d NUTS chains per GPU (%d units/GPU; %d '
                                   'spectra over %d GPU(s), whole spectra per rank), Series_pos 81x161 (D=331), real NUTS '
                                   'transitions; step = one launch of %d leapfrog rounds'
                                   % (args.spectra, args.chains, n_units, n_global, world, args.rounds),
                       'ranks_seen': ranks_seen,
                       'rccl': {'backend': backend + (' (= RCCL on ROCm)' if backend == 'nccl' else ''),
                                'device_ids': devices_seen, 'data_path_collectives': 0,
                                'launched_by': 'bench.py itself (child torch.distributed.run)'
                                if os.environ.get('BDRT_BENCH_SELF_LAUNCHED') else
                                ('torch.distributed.run' if 'RANK' in os.environ else 'plain python, one process')},
                       'units_per_gpu': n_units, 'rounds_per_launch': args.rounds, 'evaluator': evaluator,
                       'evals_in_timed_region': evals, 'timed_region_s': elapsed, 'setup_ms': setup_ms},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F64_MFMA_TFLOPS, 'traffic': traffic,
                         # `traffic` is NOT measured by this process: it is bytes-per-round of the committed PMC passes x the rounds
                         # of a launch (PMC counters need rocprofv3 around the process: tools/profile_bench.sh)
                         'traffic_measured': False,
                         'kernel': 'nuts_kernel', 'avg_launch_ms': avg_ms, 'launches': launches,
                         'flop_per_eval_algorithmic': FLOP_PER_EVAL,
                         # the same launch priced on what the structured path executes (band convolutions instead of dense
                         # L products) and on the MFMA pipe alone (A and A^T tiles)
                         'flop_per_eval_executed': flop_executed,
                         'frac_executed': achieved * flop_executed / FLOP_PER_EVAL / PEAK_F64_MFMA_TFLOPS,
                         'mfma_flop_per_eval': mfma_flop,
                         'mfma_pipe_frac': achieved * mfma_flop / FLOP_PER_EVAL / PEAK_F64_MFMA_TFLOPS,
                         'traffic_source': traffic_src,
                         'pipe_busy': pipe_busy, 'mfma_busy': mfma_busy, 'pipe_busy_measured': False, 'pipe_busy_source': pipe_src,
                         'executed_tflops_structured_path': achieved * flop_executed / FLOP_PER_EVAL},
        }
        if one_device:
            line['config']['test_mode'] = 'all %d ranks on ONE device over gloo (BDRT_BENCH_ONE_DEVICE): not a scaling measurement' % world
        if sweep is not None:
            line['config']['raw_logp_grad_kernel_sweep'] = sweep
        if few is not None:
            line['config']['few_chains'] = few
        if strong is not None:
            line['config']['strong_scaling'] = strong
        elif world == 1:
            is_config4 = args.spectra == N_SPECTRA and args.chains == CHAINS_PER_SPECTRUM
            line['config']['strong_scaling'] = {'workload': ('BASELINE config 4 as written: %d spectra x %d chains in total' % (N_SPECTRA, args.chains)) if is_config4
                                                else ('this run\'s own job (%d spectra x %d chains: NOT config 4)' % (args.spectra, args.chains)),
                                                'value': value, 'unit': 'evals/s', 'note': 'the headline job itself at N = 1'}
        if mid is not None:
            line['config']['mid_occupancy'] = mid          # sampler_kind 0: 16 chains per workgroup, 1: one chain per 512-thread workgroup, 3: one chain per wave
        if shapes is not None:
            line['config']['shapes'] = shapes
        if cpu is not None:
            line['cpu_baseline'] = cpu
    else:
        line = None

    # ---- N > 1: one complete sample_sharded call (broadcast of the problem + short run + summary gather over RCCL), timed
    # apart from the rate and AFTER it has been secured: a watchdog prints the line and ends the process should the exercise
    # (or the teardown of the process group) not come back -- the steady-state measurement must not hang on it
    printed = threading.Event()
    finished = threading.Event()
    emit_lock = threading.Lock()

    def emit(extra=None):
        # one critical section for "amend the line + print it": the main thread and the watchdog can both get here
        with emit_lock:
            if rank == 0 and not printed.is_set():
                if extra:
                    line['config'].update(extra)
                printed.set()
                print(json.dumps(line), flush=True)

    if use_dist:
        def watchdog():
            if not finished.wait(float(os.environ.get('BDRT_BENCH_ROUNDTRIP_TIMEOUT', '180'))):
                emit({'dist_roundtrip': {'error': 'timed out; the rate above was measured before it'}})
                # a hung collective / teardown is never a success: the line stands (the rate was measured before the
                # exercise), the exit code says the round trip failed (WATCHDOG_EXIT; self_launch relays both)
                os._exit(WATCHDOG_EXIT)
        threading.Thread(target=watchdog, daemon=True).start()
        small = dict(kw, Z=np.atleast_2d(Zall)[:8 * world]) if rank == 0 else None
        t_rt = time.perf_counter()
        try:
            dist.barrier()
            res = par.sample_sharded(small, 8 * world, args.chains, 6, 4, seed=1234, control={'max_treedepth': 5},
                                     gather='summary')
            torch.cuda.synchronize()
            dist.barrier()
            roundtrip = {'ms': (time.perf_counter() - t_rt) * 1e3, 'spectra': 8 * world, 'chains': args.chains,
                         'warmup': 6, 'draws': 4, 'gather': 'summary', 'finite': bool(np.all(np.isfinite(res['mean'])))}
        except Exception as exc:                  # reported in the line; the steady-state rate stands on its own
            roundtrip = {'error': '%s: %s' % (type(exc).__name__, exc)}
        emit({'dist_roundtrip': roundtrip})
    emit()
    if use_dist:
        dist.destroy_process_group()
    finished.set()


if __name__ == '__main__':
    main()
