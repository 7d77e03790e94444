#!/bin/bash
# kernel timeline of the K = 161 MAP fit (rocprofv3 --kernel-trace): a stretch of rounds; $1 = 1: the random start alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06p
[ "${1:-0}" = 1 ] && export BDRT_MAP_SINGLE_START=1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r06p/p -o run -- python3 tools/dbg/map_once.py > gpurun_out/r06p/log.txt 2>&1
DB=$(find gpurun_out/r06p/p -name '*.db' | head -1)
python3 tools/dbg/rocpd_timeline.py "$DB" 1500 24 > gpurun_out/r06p/timeline.txt 2>&1
cat gpurun_out/r06p/timeline.txt
find gpurun_out/r06p -name '*.db' -delete
