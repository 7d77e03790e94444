#!/usr/bin/env python3
"""bench.py -- HMC log-posterior+gradient evaluations per second, 81 frequencies x 161 tau-basis (BASELINE.json).

Workload (config.workload): BASELINE config 4 sharded the weak-scaling way -- every GPU samples the SAME batch of
512 synthetic 2-ZARC spectra (shared frequency / tau grids => one A/L set in HBM) with 8 NUTS chains per spectrum
(4096 units per GPU = 16 chains on each of the 256 CUs; rank r owns chain ids 8r..8r+7, so 4 GPUs = "32 chains
total" per spectrum, 8 GPUs = 64).  `--chains 4` gives the 2048-unit variant (half of the CUs idle).  Model
Series_pos, D = 331, sampling-mode hyper-parameters.  Each unit is a real NUTS chain (device-resident
transitions, bdrt_nuts.hip); a "step" is one leapfrog round: every unit performs one log-posterior+gradient
evaluation inside its current tree.  No data-path collective: units are independent.

Prints ONE JSON line (rank 0).  `value` = gradient evaluations actually executed by all ranks during the timed
region / wall time (max over ranks).  `roofline` is for the dominant kernel (nuts_kernel): achieved = algorithmic
FLOPs per launch (4.27e5 per evaluation) / HIP-event launch duration.  `cpu_baseline` = the CPU
oracle (plain C port of the same log-posterior+gradient) timed on this box's host cores, one process per core.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 4.27e5      # 4*(2Nf*K + 3K^2) + ~12k element-wise, Nf=81, K=161
BYTES_PER_EVAL = 8.39e5     # unique operand bytes when B = 1
# what the structured path really executes per evaluation on a log-uniform grid: A GEMMs on padded 176 x 168 tiles
# (2 * 462 MFMA * 2048 flop / 16 chains = 1.18e5) + six 17-tap convolutions (3.3e4) + ~1.5e4 element-wise
FLOP_PER_EVAL_EXECUTED = 1.66e5
PEAK_F64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak; measured 78.05 by tools/mfma_probe (profiles/)
NF, K = 81, 161
N_SPECTRA, CHAINS_PER_SPECTRUM = 512, 8
ROUNDS_PER_LAUNCH = 50


def synth_spectra(n, seed=20260101):
    """512 two-ZARC spectra, parameters per SURVEY 8(d) config 4; noise models cycled uniform/Orazem/Macdonald 0.25 %."""
    rs = np.random.RandomState(seed)
    f = np.logspace(6, -2, NF)
    w = 2 * np.pi * f
    Z = np.empty((n, 2 * NF))
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
        scale = np.std(np.abs(z)) / np.sqrt(NF / 81)       # Inverter._scale_Z (inversion.py:2437-2441)
        z = z / scale
        Z[s] = np.concatenate([z.real, z.imag])
    return f, Z


def cpu_baseline(seconds=12.0):
    """Oracle log_prob+grad evaluations/s on the host cores: one worker process per core (pystan's own layout)."""
    ncores = min(os.cpu_count() or 1, 16)
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as orc
from bench import synth_spectra, NF, K
f, Z = synth_spectra(1)
tau = 1 / (2 * np.pi * np.logspace(10, -6, K)); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([orc.construct_A(f, 'real', tau=tau, epsilon=eps), orc.construct_A(f, 'imag', tau=tau, epsilon=eps)])
blk = dict(A=A, L0=orc.construct_L(tau, eps, 0), L1=orc.construct_L(tau, eps, 1), L2=0.75 * orc.construct_L(tau, eps, 2), nonneg=True)
m = orc.OracleModel([blk], Z[0], f, ups_alpha=1.0, ups_beta=0.1)
th = np.random.RandomState(int(sys.argv[1])).uniform(-1, 1, m.D)
orc.eval_loop(m, th, 50)
n = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < float(sys.argv[2]):
    orc.eval_loop(m, th, 500); n += 500
print(n, time.perf_counter() - t0)
''' % ROOT
    procs = [subprocess.Popen([sys.executable, '-c', code, str(i), str(seconds)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, cwd=ROOT) for i in range(ncores)]
    total = 0.0
    for p in procs:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('cpu_baseline worker failed: ' + err.decode()[-400:])
        n, t = out.decode().split()
        total += float(n) / float(t)
    return dict(value=total, unit='evals/s', cores=ncores, kind='port',
                sample='%d processes x %.0f s of oracle/bdrt_oracle.c log_prob+grad (Series_pos, 81x161, jacobian on), '
                       'plain C -O2, one process per core' % (ncores, seconds))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20000)
    ap.add_argument('--warmup', type=int, default=2000)
    ap.add_argument('--spectra', type=int, default=N_SPECTRA)
    ap.add_argument('--chains', type=int, default=CHAINS_PER_SPECTRUM)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--phase-profile', action='store_true', help='print the in-kernel cycle breakdown (perturbs timing)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus must equal WORLD_SIZE')

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU: workers are plain subprocesses
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build()
        cpu = cpu_baseline(args.cpu_seconds)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product has no CPU path')
    torch.cuda.set_device(local_rank)
    # BDRT_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank too (lets a 1-GPU box exercise it)
    use_dist = world > 1 or (bool(os.environ.get('BDRT_BENCH_FORCE_DIST')) and 'RANK' in os.environ)
    if use_dist:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from bayes_drt_amd import _lib, matrices as gm
    from bayes_drt_amd._lib import NutsControl, check, ptr
    from bayes_drt_amd.model import Problem
    lib = _lib.require_gpu()
    check(lib.bdrt_set_device(local_rank), 'bdrt_set_device')

    # ---- problem: matrices built on the GPU, 512 spectra resident in HBM ----
    f, Z = synth_spectra(args.spectra)
    basis_freq = np.logspace(10, -6, K)
    tau = 1 / (2 * np.pi * basis_freq)
    eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(basis_freq, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    blk = dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)       # sample-mode scaling (inversion.py:1725-1730)
    prob = Problem([blk], Z, f, sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)
    assert prob.D == 2 * K + 9

    n_units = args.spectra * args.chains
    spec = np.repeat(np.arange(args.spectra, dtype=np.int32), args.chains)
    chain_id = np.tile(np.arange(args.chains, dtype=np.int32), args.spectra) + rank * args.chains
    ctrl = NutsControl()
    lib.bdrt_nuts_defaults(C.byref(ctrl))
    ctrl.adapt_delta, ctrl.adapt_t0 = 0.9, 10.0                          # inversion.py:1221
    # warm-up long enough that no chain finishes inside the benchmark (adaptation windows at 100, 150, 250, ...)
    h = lib.bdrt_sampler_create(prob.handle, n_units, ptr(spec), ptr(chain_id), 1000000, 1, C.c_uint64(1234), None,
                                C.byref(ctrl))
    if not h:
        raise SystemExit('bdrt_sampler_create: ' + lib.bdrt_last_error().decode())

    def advance(rounds):
        left = rounds
        while left > 0:
            r = min(ROUNDS_PER_LAUNCH, left)
            check(lib.bdrt_sampler_advance(h, r, None), 'bdrt_sampler_advance')
            left -= r

    def sync_all():
        check(lib.bdrt_sampler_sync(h), 'bdrt_sampler_sync')
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    advance(args.warmup)
    sync_all()
    if args.phase_profile:
        check(lib.bdrt_sampler_phase_profile(h, 1, None), 'phase_profile')
    ms0 = C.c_double(); nl0 = C.c_int64()
    check(lib.bdrt_sampler_kernel_time(h, C.byref(ms0), C.byref(nl0), 1), 'kernel_time')
    n0 = lib.bdrt_sampler_total_leapfrogs(h)
    sync_all()
    t0 = time.perf_counter()
    advance(args.steps)
    check(lib.bdrt_sampler_sync(h), 'bdrt_sampler_sync')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if use_dist:
        dist.barrier()
    elapsed = t1 - t0
    n1 = lib.bdrt_sampler_total_leapfrogs(h)
    ms = C.c_double(); nl = C.c_int64()
    check(lib.bdrt_sampler_kernel_time(h, C.byref(ms), C.byref(nl), 0), 'kernel_time')
    evals = float(n1 - n0)
    if args.phase_profile and rank == 0:
        cyc = (C.c_longlong * 32)()
        check(lib.bdrt_sampler_phase_profile(h, 0, cyc), 'phase_profile')
        names = ['tile:scalars', 'tile:x', 'tile:gemmA', 'tile:Zacc', 'tile:likelihood', 'tile:x2', 'tile:gemmL',
                 'tile:prior', 'tile:gemmBwd', 'tile:epilogue', 'nuts:A kick-drift', 'nuts:C kick+kin', 'nuts:S1',
                 'nuts:D tree', 'nuts:S2', 'nuts:E next', 'nuts:S3']
        tot = float(sum(cyc[:17])) or 1.0
        n_wg = (n_units + 15) // 16
        for k, nm in enumerate(names):
            print('PHASE %-20s %6.2f %%  %9.0f cycles/round' % (nm, 100 * cyc[k] / tot, cyc[k] / n_wg / args.steps), file=sys.stderr)
        wnames = ['tile', 'C', 'S1', 'D', "A'", "E+S3+A''", 'end-of-round barrier wait']
        for k, nm in enumerate(wnames):
            print('WAVE-AVG %-26s %9.0f cycles/round' % (nm, cyc[17 + k] / n_wg / args.steps / 8), file=sys.stderr)
        print('half-waves per round in stage E (new start point): %.2f of 16' % (cyc[24] / n_wg / args.steps * 2), file=sys.stderr)
    lib.bdrt_sampler_destroy(h)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([evals], dtype=torch.float64, device='cuda')
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        elapsed, evals = float(t.item()), float(e.item())

    if rank == 0:
        value = evals / elapsed
        launches = max(int(nl.value), 1)
        avg_ms = ms.value / launches
        evals_per_launch = (n1 - n0) / launches                       # this rank's launches
        achieved = evals_per_launch * FLOP_PER_EVAL / (avg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get('nuts_kernel_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': 'HMC log-posterior+grad evals/sec (81 freq x 161 tau-basis)',
            'value': value, 'unit': 'evals/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config4-shard: %d synthetic 2-ZARC spectra x %d NUTS chains per GPU (%d units/GPU), '
                                   'Series_pos 81x161 (D=331), real NUTS transitions, step = one leapfrog round'
                                   % (args.spectra, args.chains, n_units),
                       'units_per_gpu': n_units, 'rounds_per_launch': ROUNDS_PER_LAUNCH,
                       'evals_in_timed_region': evals},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F64_MFMA_TFLOPS, 'traffic': traffic,
                         'kernel': 'nuts_kernel', 'avg_launch_ms': avg_ms, 'launches': launches,
                         'hbm_frac_B1_accounting': value / args.gpus * BYTES_PER_EVAL / 8e12,
                         'executed_tflops_structured_path': achieved * FLOP_PER_EVAL_EXECUTED / FLOP_PER_EVAL},
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
