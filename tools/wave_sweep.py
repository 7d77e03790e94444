"""The one-chain-per-wave sampler (bdrt_wave.h) at the benchmark shape (81 x 161) against the other kernels of the headline
family: evals/s at 4 ... 4096 units.  usage: python tools/wave_sweep.py [units ...]"""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
UNITS = (4, 256, 512, 768, 1024, 1536, 2048, 3072, 4096)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from bench import build_problem_kwargs
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    kw = build_problem_kwargs(64)
    blocks, Z, f = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, f, **kw)
    units = [int(a) for a in sys.argv[3:]] or UNITS
    for n in units:
        spec = (np.arange(n) % 64).astype(np.int32)
        s = Sampler(prob, n, 1000000, 1, 7, spec=spec)
        s.advance(600); s.sync()
        n0 = s.total_leapfrogs(); t0 = time.perf_counter()
        for _ in range(4):
            s.advance(500)
        s.sync()
        dt = time.perf_counter() - t0
        print('%-34s units %5d (kernel kind %d): %7.2f M evals/s, %6.2f us per round of all units' % (
            sys.argv[2], n, s.kind(), (s.total_leapfrogs() - n0) / dt / 1e6, dt / 2000 * 1e6), flush=True)
        if os.environ.get('WAVE_PROF'):
            import ctypes as C
            lib = prob._lib
            lib.bdrt_sampler_phase_profile(s.handle, 1, None)
            R = 400
            s.advance(R); s.sync()
            cyc = (C.c_longlong * 32)()
            lib.bdrt_sampler_phase_profile(s.handle, 0, cyc)
            per = [cyc[k] / (R * n) for k in range(32)]
            print('   cycles per chain-round: ' + ' '.join('%d:%.0f' % (k, per[k]) for k in range(32) if per[k] > 0) + '  sum %.0f' % sum(per), flush=True)
        s.close()
else:
    which = os.environ.get('WAVE_SWEEP', 'wave,duo,16').split(',')
    table = {'wave': ('one chain per wave', {'BDRT_WAVE': '1'}),
             'solo': ('one workgroup per CU', {'BDRT_WAVE': '0', 'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '0'}),
             'duo': ('two workgroups per CU', {'BDRT_WAVE': '0', 'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '1'}),
             '16': ('16-chain kernel', {'BDRT_WAVE': '0', 'BDRT_SOLO': '0', 'BDRT_WIDE1': '0'})}
    for k in which:
        label, env = table[k]
        subprocess.run([sys.executable, __file__, 'child', label] + sys.argv[1:], env=dict(os.environ, **env))
