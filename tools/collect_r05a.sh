#!/bin/bash
# Round-5 first GPU batch: validates the split build, baseline bench line + phase profile, and the VALU / SALU instruction
# counts of the headline kernel with one evaluator phase skipped at a time (BDRT_DEBUG_SKIP: results wrong on purpose) --
# the dynamic per-phase instruction mix of profiles/r05/isa_mix_nuts_kernel.txt.  Writes gpurun_out/r05a/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05a
mkdir -p $OUT
PARTS="${*:-tests bench ablate}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_config4.py tests/test_gpu_toep_gen.py tests/test_gpu_wave.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1
  tail -3 $OUT/pytest_subset.txt
fi
export BDRT_BENCH_NO_TABLES=1
if has bench; then
  python bench.py --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench.err
  python bench.py --no-cpu-baseline --phase-profile > /dev/null 2> $OUT/phase_profile.txt
  cat $OUT/bench_line.json | cut -c1-400
fi
if has ablate; then
  SET="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"
  for SK in 0 4 8 1 2 15; do
    BDRT_DEBUG_SKIP=$SK timeout 300 rocprofv3 --pmc $SET -d $OUT/ab$SK -o bench -- python3 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/ab$SK.log 2>&1
    DB=$(find $OUT/ab$SK -name '*.db' | head -1)
    echo "== BDRT_DEBUG_SKIP=$SK" >> $OUT/ablate.txt
    grep '"metric"' $OUT/ab$SK.log | cut -c1-200 >> $OUT/ablate.txt
    if [ -n "$DB" ]; then
      python3 - "$DB" >> $OUT/ablate.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%nuts_kernel%' "
                   "group by counter_name order by 1").fetchall()
for r in rows: print('%-32s dispatches %4d  avg/dispatch %.6g' % r)
PY
    fi
  done
  cat $OUT/ablate.txt
fi
find $OUT -name '*.db' -delete
