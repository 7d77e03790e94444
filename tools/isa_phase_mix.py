"""Static instruction mix of the headline sampler kernel PER PHASE of a round: the listing of one nuts_kernel instantiation
(tools/isa_kernel.sh) is cut at the round loop's workgroup barriers -- B1 (X ready), B2 (A x ready), B3 (g ready), B4 (A^T g ready) -- into
  [loop head .. B1)  bookkeeping of the previous leapfrog's tail is NOT here: parameters (P1) + prior chain (P2)
  [B1 .. B2)         forward GEMM (+ the spectrum request)
  [B2 .. B3)         likelihood (P3)
  [B3 .. B4)         state request + backward GEMM
  [B4 .. loop end)   chain rule + the sampler's stages C, S1, D, A', Z, E (all their branches: a STATIC count -- a round executes one path)
and every instruction is classed: fp64 arithmetic, integer / address, moves, compares / selects, cross-lane (DPP, readlane, writelane: the
latter two are SGPR spill traffic), MFMA, scalar ALU, LDS, vector memory, s_waitcnt, s_nop, branches.
Usage: python tools/isa_phase_mix.py <kernel.s> [label]"""
import collections
import re
import sys


def classify(s):
    op = s.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'sgpr_spill'
    if op.startswith('v_'):
        if 'dpp' in op or 'row_' in s or 'quad_perm' in s or op.startswith('v_readfirstlane') or 'permlane' in op: return 'cross_lane'
        if 'f64' in op: return 'fp64'
        if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'mov'
        if op.startswith('v_cmp') or op.startswith('v_cndmask'): return 'cmp_sel'
        return 'int_addr'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'): return 'vmem'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    bars = [i for i, l in enumerate(lines) if l.strip().startswith('s_barrier')]
    # the four barriers of the evaluation: the run of four whose LAST one is followed by the longest stretch (the sampler's stages)
    best = max(range(len(bars) - 4), key=lambda k: bars[k + 4] - bars[k + 3])
    b1, b2, b3, b4, end = bars[best], bars[best + 1], bars[best + 2], bars[best + 3], bars[best + 4]
    # the round loop: the backward branch with the longest span; its header is where a round's first phase begins
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l.strip())] if m}
    span = (0, bars[best - 1], end)
    for i, l in enumerate(lines):
        m = re.match(r'\s*s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > span[0]:
            span = (i - labels[m.group(1)], labels[m.group(1)], i)
    head, end = span[1], span[2]
    regions = [('P1 + P2 (parameters, prior chain)', head, b1), ('forward GEMM', b1, b2), ('P3 (likelihood)', b2, b3),
               ('backward GEMM', b3, b4), ('chain rule + stages C S1 D A\' Z E (all branches)', b4, end)]
    keys = ['fp64', 'int_addr', 'mov', 'cmp_sel', 'cross_lane', 'sgpr_spill', 'mfma', 'salu', 'branch', 'lds', 'vmem', 'waitcnt', 'nop']
    print('== %s' % label)
    print('%-50s %6s ' % ('phase (static instruction counts)', 'VALU') + ' '.join('%10s' % k for k in keys))
    tot = collections.Counter()
    for name, lo, hi in regions:
        c = collections.Counter()
        for l in lines[lo:hi]:
            s = l.strip()
            if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
                continue
            c[classify(s)] += 1
        valu = sum(c[k] for k in ('fp64', 'int_addr', 'mov', 'cmp_sel', 'cross_lane', 'sgpr_spill'))
        print('%-50s %6d ' % (name, valu) + ' '.join('%10d' % c[k] for k in keys))
        tot.update(c)
    valu = sum(tot[k] for k in ('fp64', 'int_addr', 'mov', 'cmp_sel', 'cross_lane', 'sgpr_spill'))
    print('%-50s %6d ' % ('round loop, all paths', valu) + ' '.join('%10d' % tot[k] for k in keys))


if __name__ == '__main__':
    main()
