#!/bin/bash
# Round-2 second measurement batch (wide-vector path, MAP): writes gpurun_out/r02b/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02b
mkdir -p $OUT
python tools/bench_config5.py 4096 --phase-profile > $OUT/config5.txt 2>&1
python tools/bench_config5.py 4096 --series-outliers >> $OUT/config5.txt 2>&1
python tools/config5_run.py > $OUT/config5_run.txt 2>&1
python tools/map_timing.py > $OUT/map_timing.txt 2>&1
BDRT_NEWTON_PROF=1 python tools/map_timing.py 2>&1 | grep "newton prof" > $OUT/newton_phases.txt
rocprofv3 --kernel-trace --stats -d /tmp/c5trace -o c5 -- python3 tools/bench_config5.py 4096 > $OUT/config5_trace.log 2>&1
DB=$(find /tmp/c5trace -name '*.db' | head -1); [ -n "$DB" ] && python3 tools/rocpd_summary.py "$DB" > $OUT/config5_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d /tmp/c5pmc_$C -o c5 -- python3 tools/bench_config5.py 4096 > $OUT/config5_pmc_$C.log 2>&1
  DB=$(find /tmp/c5pmc_$C -name '*.db' | head -1); [ -n "$DB" ] && python3 tools/rocpd_pmc.py "$DB" $C 2>&1 | grep -i "nuts_kernel\|columns" > $OUT/config5_pmc_$C.txt
done
ls -la $OUT; head -4 $OUT/config5_kernel_stats.txt; cat $OUT/config5_pmc_*.txt | grep nuts
