"""One sampler of n units of the headline problem (81 x 161) advanced for a fixed number of leapfrog rounds: the process that
tools/profile_pmc.sh wraps in rocprofv3.  usage: python3 tools/wave_run.py <units> <kernel: wave|duo|solo|16> [rounds] [launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]); kern = sys.argv[2] if len(sys.argv) > 2 else 'wave'
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 500
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 4
env = {'wave': {'BDRT_WAVE': '1'}, 'solo': {'BDRT_WAVE': '0', 'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '0'},
       'duo': {'BDRT_WAVE': '0', 'BDRT_SOLO': '1', 'BDRT_SOLO_DUO': '1'}, '16': {'BDRT_WAVE': '0', 'BDRT_SOLO': '0', 'BDRT_WIDE1': '0'}}[kern]
os.environ.update(env)
import numpy as np
from bench import build_problem_kwargs
from bayes_drt_amd.engine import Sampler
from bayes_drt_amd.model import Problem
kw = build_problem_kwargs(64)
blocks, Z, f = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
prob = Problem(blocks, Z, f, **kw)
spec = (np.arange(n) % 64).astype(np.int32)
s = Sampler(prob, n, 1000000, 1, 7, spec=spec)
s.advance(600); s.sync()
n0 = s.total_leapfrogs(); t0 = time.perf_counter()
for _ in range(launches):
    s.advance(rounds)
s.sync()
dt = time.perf_counter() - t0
print('%s units %d (kernel kind %d): %.2f M evals/s, %.2f us per round of all units, %d leapfrogs in the timed launches' % (
    kern, n, s.kind(), (s.total_leapfrogs() - n0) / dt / 1e6, dt / (launches * rounds) * 1e6, s.total_leapfrogs() - n0), flush=True)
s.close()
