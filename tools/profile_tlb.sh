#!/bin/bash
# Address-translation and L2 counters of the sampler kernel (PMC passes, no trace domains).  Usage: tools/profile_tlb.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tlb_$TAG
mkdir -p $OUT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
i=0
for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET -d $OUT/p$i -o bench -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then
    python3 - "$DB" >> $OUT/summary.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%nuts_kernel%' "
                   "group by counter_name order by 1").fetchall()
for r in rows: print('%-36s dispatches %4d  avg/dispatch %.6g' % r)
PY
  else tail -3 $OUT/p$i.log >> $OUT/summary.txt
  fi
done
cat $OUT/summary.txt
find $OUT -name '*.db' -size +20M -delete
