#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06o
for sp in 1 0; do
  export BDRT_NEWTON_SPEC=$sp
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06o/p$sp -o run -- python3 tools/dbg/map_once.py > gpurun_out/r06o/log$sp.txt 2>&1
  DB=$(find gpurun_out/r06o/p$sp -name '*.db' | head -1)
  echo "== BDRT_NEWTON_SPEC=$sp"; [ -n "$DB" ] && python3 tools/rocpd_summary.py "$DB" | head -14 | cut -c1-150
done > gpurun_out/r06o/kernels.txt 2>&1
find gpurun_out/r06o -name '*.db' -delete; cat gpurun_out/r06o/kernels.txt
