#!/bin/bash
# Instruction-fetch counters of the sampler kernel (one PMC pass, no trace domains).  Usage: tools/profile_icache.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ic_$TAG
mkdir -p $OUT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
i=0
for SET in "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET -d $OUT/p$i -o bench -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then
    python3 - "$DB" >> $OUT/summary.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%nuts_kernel%' "
                   "group by counter_name order by 1").fetchall()
for r in rows: print('%-32s dispatches %4d  avg/dispatch %.6g' % r)
PY
  else tail -5 $OUT/p$i.log >> $OUT/summary.txt
  fi
done
cat $OUT/summary.txt
find $OUT -name '*.db' -size +20M -delete
