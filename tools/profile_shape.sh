#!/bin/bash
# Kernel trace of the sampler at one of the other shapes (tools/shape_rates.py: real NUTS chains in warm-up, 4096 units).
# Usage: tools/profile_shape.sh NFxK; writes gpurun_out/prof_shape_NFxK/kernel_stats.txt
set -u
SH=${1:-81x101}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_shape_$SH
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o shape -- python3 tools/shape_rates.py 4096 $SH > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
if [ -n "$DB" ]; then python3 tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.txt; fi
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL\|amdgpu.ids" $OUT/trace.log | tail -3
head -12 $OUT/kernel_stats.txt
find $OUT -name '*.db' -size +20M -delete
