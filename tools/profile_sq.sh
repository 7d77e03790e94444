#!/bin/bash
# SQ counters of the sampler kernel (two PMC passes, no trace domains), summarised per dispatch.  Usage: tools/profile_sq.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
export BDRT_BENCH_NO_TABLES=1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
B="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
i=0
for SET in "$A" "$B"; do
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
  fi
done
cat $OUT/summary.txt
find $OUT -name '*.db' -size +20M -delete
