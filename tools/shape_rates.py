"""Per-shape sampler rate (bench.py's steady_rate: real NUTS chains in warm-up) with the A operands from the LDS generator table
against the streamed fragments (BDRT_STREAM_A=1), 16-chain kernel at 4096 units; the rows of bench.py's config.shapes.
usage: python tools/shape_rates.py [units] [NFxK ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import bench
from bayes_drt_amd import _lib
from bayes_drt_amd.model import Problem

args = sys.argv[1:]
units = int(args.pop(0)) if args and 'x' not in args[0] else 4096
shapes = [tuple(int(v) for v in a.split('x')) for a in args] or list(bench.SHAPES)
lib = _lib.require_gpu()
ctrl = _lib.NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl))
print('%9s %6s %28s %28s' % ('shape', 'units', 'table (evaluator, M evals/s, frac)', 'streamed fragments'))
for nf, k in shapes:
    e_s = 4 * (2 * nf * k + 3 * k * k)
    cells = []
    for env in ({}, {'BDRT_STREAM_A': '1'}):
        for key in ('BDRT_STREAM_A',):
            os.environ.pop(key, None)
        os.environ.update(env)
        p = Problem(**bench.shape_problem_kwargs(nf, k, 64))
        r, kind = bench.steady_rate(p, units, ctrl, 64)
        cells.append('%d %7.1f %.3f (kind %d)' % (p.evaluator(), r / 1e6, r * e_s / 1e12 / bench.PEAK_F64_MFMA_TFLOPS, kind))
        p.close()
    os.environ.pop('BDRT_STREAM_A', None)
    print('%3d x %3d %6d %28s %28s' % (nf, k, units, cells[0], cells[1]), flush=True)
