#!/bin/bash
# SQ counters of one kernel (two PMC passes, no trace domains), summarised per dispatch.
# usage: tools/profile_pmc.sh <tag> <kernel name pattern> <python script and args ...>
set -u
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
i=0
: > $OUT/summary.txt
for SET in "$A" "$B"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET -d $OUT/p$i -o run -- python3 "$@" > $OUT/p$i.log 2>&1
  grep -v amdgpu.ids $OUT/p$i.log | tail -2 >> $OUT/summary.txt
  DB=$(find $OUT/p$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then
    python3 - "$DB" "$PAT" >> $OUT/summary.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value), max(value) from counters_collection where kernel_name like ? "
                   "group by counter_name order by 1", ('%' + sys.argv[2] + '%',)).fetchall()
for r in rows: print('%-32s dispatches %4d  avg/dispatch %.6g  max %.6g' % r)
PY
  fi
done
cat $OUT/summary.txt
find $OUT -name '*.db' -size +20M -delete
