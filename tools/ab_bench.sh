#!/bin/bash
# GPU box: the headline bench line (tables off) for every library under bayes_drt_amd/variants/ and for the default build.
# Usage: tools/ab_bench.sh [--profile] [tag ...]     (no tags: all variants)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BDRT_BENCH_NO_TABLES=1
PROF=0; if [ "${1:-}" = "--profile" ]; then PROF=1; shift; fi
OUT=gpurun_out/ab; mkdir -p $OUT
TAGS="$*"; if [ -z "$TAGS" ]; then TAGS="default $(ls bayes_drt_amd/variants/ 2>/dev/null | sed -n 's/^libbdrt_\(.*\)\.so$/\1/p')"; fi
for T in $TAGS; do
  if [ "$T" = default ]; then unset BDRT_LIBRARY; else export BDRT_LIBRARY=$PWD/bayes_drt_amd/variants/libbdrt_$T.so; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline 2> $OUT/$T.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-24s %8.2f M evals/s  %7.3f ms/step  frac %.4f' % ('$T', d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"
  done
  if [ $PROF = 1 ]; then python bench.py --no-cpu-baseline --phase-profile 2>&1 >/dev/null | grep -E "WAVE-AVG|FINE" | sed "s/^/    [$T] /"; fi
done | tee $OUT/summary.txt
