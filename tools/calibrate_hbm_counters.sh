#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts on the sampler's row pattern (tools/ubench/hbm_calib.hip).
# Writes gpurun_out/hbm_calib.txt: per kernel the counter (KB) and the ratio known bytes / (counter * 1024).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/hbm_calib; mkdir -p $OUT
tools/ubench/hbm_calib > $OUT/known.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $OUT/$C -o calib -- tools/ubench/hbm_calib > $OUT/$C.log 2>&1
done
python3 - > gpurun_out/hbm_calib.txt <<'PY'
import glob, sqlite3
known = 256 * 16 * 256 * 352 * 8.0
print(open('gpurun_out/hbm_calib/known.txt').read())
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    db = glob.glob('gpurun_out/hbm_calib/%s/**/*.db' % c, recursive=True)
    if not db:
        print(c, 'no database'); continue
    cur = sqlite3.connect(db[0]).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (c,)).fetchall()
    for name, v, n in rows:
        if 'calib' not in name: continue
        moved = known if (('read' in name or 'copy' in name) and c == 'FETCH_SIZE') or (('write' in name or 'copy' in name) and c == 'WRITE_SIZE') else 0.0
        print('%-12s %-24s counter %.6e KB = %.6e bytes; known %.6e bytes; known / counted = %s' % (
            c, name.split('(')[0][-24:], v, v * 1024, moved, ('%.4f' % (moved / (v * 1024))) if v > 0 and moved > 0 else 'n/a'))
PY
cat gpurun_out/hbm_calib.txt
find $OUT -name '*.db' -delete
