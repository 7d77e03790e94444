import re, sys, collections
def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'):
        if 'f64' in op: return 'vf64'
        if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'vmov'
        if 'dpp' in op or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane') or 'permlane' in op or 'swizzle' in op: return 'vxl'
        if op.startswith('v_cmp') or op.startswith('v_cndmask'): return 'vcmp'
        return 'vint'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.split('_')[0] in ('global','buffer','scratch','flat'): return 'vmem'
    return 'other'
lines = open(sys.argv[1]).read().split('\n')
blocks = []  # (label, start_line, counts, succ)
cur = dict(label='entry', start=0, cnt=collections.Counter(), succ=[], ops=[])
for i,l in enumerate(lines):
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        blocks.append(cur)
        cur = dict(label=m.group(1), start=i, cnt=collections.Counter(), succ=[], ops=[])
        continue
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
    op = s.split()[0]
    c = classify(op)
    # dpp modifiers
    if c in ('vf64','vint','vmov','vcmp') and ('row_' in s or 'quad_perm' in s or 'wave_' in s or 'bank_mask' in s): c = 'vxl'
    cur['cnt'][c] += 1
    cur['ops'].append(s)
    m = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', s)
    if m: cur['succ'].append(m.group(1))
blocks.append(cur)
keys = ['vf64','vint','vmov','vcmp','vxl','mfma','salu','lds','vmem','wait','nop','bar']
print('%-12s %6s ' % ('label','line') + ' '.join('%5s'%k for k in keys) + '  VALU  succ')
tot = collections.Counter()
for b in blocks:
    n = sum(b['cnt'].values())
    if n == 0: continue
    valu = sum(b['cnt'][k] for k in ('vf64','vint','vmov','vcmp','vxl'))
    print('%-12s %6d ' % (b['label'], b['start']) + ' '.join('%5d'%b['cnt'][k] for k in keys) + ' %5d  %s' % (valu, ','.join(b['succ'])))
    tot.update(b['cnt'])
print('TOTAL', dict(tot))
