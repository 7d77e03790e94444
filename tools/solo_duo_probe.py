"""The one-chain-per-workgroup kernel at the benchmark shape (81 x 161), one workgroup per CU (BDRT_SOLO_DUO=0) against two
(BDRT_SOLO_DUO=1: 128 VGPRs, 16 of the chain's rows in LDS): evals/s at 256 ... 2048 units, beside the 16-chain kernel."""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from bench import build_problem_kwargs
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    kw = build_problem_kwargs(64)
    blocks, Z, f = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, f, **kw)
    for n in (4, 256, 384, 512, 768, 1024, 1536, 2048):
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
        s.close()
else:
    for label, env in (('one workgroup per CU', {'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '0'}),
                       ('two workgroups per CU', {'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '1'}),
                       ('16-chain kernel', {'BDRT_SOLO': '0', 'BDRT_WIDE1': '0'})):
        subprocess.run([sys.executable, __file__, 'child', label], env=dict(os.environ, **env))
