"""Per-kernel register / spill / scratch table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py bayes_drt_amd/csrc/bdrt_wave.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=fast', '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name)}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print('%-60s %5s %5s %6s %6s %8s %4s' % ('kernel', 'VGPR', 'SGPR', 'vspill', 'sspill', 'scratch', 'occ'))
for r in rows:
    print('%-60s %5s %5s %6s %6s %8s %4s' % (r['name'][-60:], r.get('VGPRs'), r.get('TotalSGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'),
                                              r.get('ScratchSize'), r.get('Occupancy')))
