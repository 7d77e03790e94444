#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as the MI355X guide
# prescribes.  Usage: tools/profile_bench.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/ and summaries.
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# the driver's own command line (bench.py --gpus 1 --steps 20 --warmup 5): 25 launches of 1000 leapfrog rounds
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $*"
# the headline job alone under the profiler: the tables beside it (mid occupancy, other shapes) launch the same kernels on other problems
export BDRT_BENCH_NO_TABLES=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_line.json
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then python3 tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.txt; fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C -d $OUT/pmc_$C -o bench -- python3 bench.py $ARGS > $OUT/pmc_$C.log 2>&1
  DB=$(find $OUT/pmc_$C -name '*.db' | head -1)
  if [ -n "$DB" ]; then python3 tools/rocpd_pmc.py "$DB" $C > $OUT/pmc_$C.txt 2>&1; fi
done
python3 tools/make_traffic_json.py $OUT > $OUT/pmc_traffic.json 2>/dev/null
ls -la $OUT
head -5 $OUT/kernel_stats.txt; for C in FETCH_SIZE WRITE_SIZE; do [ -f $OUT/pmc_$C.txt ] && head -8 $OUT/pmc_$C.txt; done
find $OUT -name '*.db' -size +20M -delete
