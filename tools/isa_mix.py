"""Static instruction mix of one kernel in a device-only assembly listing (hipcc --offload-device-only -S).
Usage: python tools/isa_mix.py file.s '<substring of the mangled kernel name>' [--loop]
--loop restricts the count to the body of the largest backward-branch loop (the sampler's round loop)."""
import collections
import re
import sys


def kernel_body(lines, key):
    start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(':') or (key in l and re.match(r'^_Z\S+:', l)))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    return lines[start:end]


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_') and 'f64' in op: return 'valu_f64'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'): return 'vmem'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    body = kernel_body(lines, sys.argv[2])
    if '--loop' in sys.argv:
        labels = {l.strip()[:-1]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            m = re.match(r'\s*s_cbranch\S*\s+(\.LBB\d+_\d+)', l) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
                best = (i - labels[m.group(1)], labels[m.group(1)], i)
        body = body[best[1]:best[2] + 1]
        print('loop body: %d lines' % len(body))
    cnt = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        cnt[l.split()[0]] += 1
    groups = collections.Counter()
    for op, n in cnt.items():
        groups[classify(op)] += n
    print('total', sum(cnt.values()), dict(groups))
    for op, n in cnt.most_common(int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 40):
        print('%6d %s' % (n, op))


if __name__ == '__main__':
    main()
