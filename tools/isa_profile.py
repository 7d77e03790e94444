"""Static instruction census of one kernel from `hipcc -S -gline-tables-only` output: VALU / MFMA / SALU / LDS / VMEM
instructions per source file and line bucket.  Usage: isa_profile.py file.s <mangled-name-substring> [bucket]"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 20
files = {}
inside = False
cur = (0, 0)
cnt = collections.defaultdict(lambda: collections.Counter())
ops = collections.defaultdict(lambda: collections.Counter())
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
    if m: files[int(m.group(1))] = m.group(2); continue
    if not inside:
        if s.startswith('_Z') and sym in s and ':' in s.split(';')[0]: inside = True
        continue
    if s.startswith('.Lfunc_end'): break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    s = s.split(';')[0].strip()
    if not s or s[0] == '.' or s.endswith(':'): continue
    op = s.split()[0]
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith('s_'): k = 'salu'
    elif op.startswith('ds_'): k = 'lds'
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): k = 'vmem'
    else: k = 'other'
    key = (files.get(cur[0], '?'), cur[1] // bucket * bucket)
    cnt[key][k] += 1
    if k == 'valu': ops[key][op] += 1
tot = collections.Counter()
for key in sorted(cnt):
    c = cnt[key]; tot.update(c)
    if c['valu'] + c['mfma'] + c['lds'] + c['vmem'] < 15: continue
    top = ' '.join('%s:%d' % (o.replace('v_', ''), n) for o, n in ops[key].most_common(5))
    print('%-24s %5d  valu %5d mfma %3d salu %4d lds %4d vmem %3d | %s' % (key[0], key[1], c['valu'], c['mfma'], c['salu'], c['lds'], c['vmem'], top))
print('TOTAL', dict(tot))
