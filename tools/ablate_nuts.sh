#!/bin/bash
# GPU box: timing ablations of the sampler's bookkeeping (library built with -DBDRT_NUTS_ABLATE=1: tools/build_variant.sh abl ...).
# BDRT_DEBUG_SKIP bits: 16 no merges of levels >= 1, 32 no checkpoint stores, 64 no proposal copies, 128 no momentum store
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BDRT_BENCH_NO_TABLES=1 BDRT_LIBRARY=$PWD/bayes_drt_amd/variants/libbdrt_abl.so
for SK in 0 16 32 64 128 48 240; do
  BDRT_DEBUG_SKIP=$SK python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skip %3d  %8.2f M evals/s  %7.3f ms/step' % ($SK, d['value']/1e6, d['ms_per_step']))"
done | tee gpurun_out/ablate_nuts.txt
