#!/bin/bash
# instruction / scalar cache counters of the sampler kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/icache_${1:-r01}
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH -d $OUT/p -o bench -- python3 bench.py --steps 300 --warmup 200 --no-cpu-baseline > $OUT/p.log 2>&1
DB=$(find $OUT/p -name '*.db' | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%nuts_kernel%' group by counter_name order by 1"):
    print('%-32s dispatches %4d  avg/dispatch %.6g' % r)
PY
find $OUT -name '*.db' -size +20M -delete
