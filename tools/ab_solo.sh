#!/bin/bash
# GPU box: the one-chain-per-workgroup kernel (4 units: one workgroup per CU; 512 units: two per CU) for every library under
# bayes_drt_amd/variants/ and the default build: us per round and the thread-0 phase profile.  Usage: tools/ab_solo.sh [tag ...]
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_solo; mkdir -p $OUT
TAGS="$*"; if [ -z "$TAGS" ]; then TAGS="default $(ls bayes_drt_amd/variants/ 2>/dev/null | sed -n 's/^libbdrt_\(.*\)\.so$/\1/p')"; fi
for T in $TAGS; do
  if [ "$T" = default ]; then unset BDRT_LIBRARY; else export BDRT_LIBRARY=$PWD/bayes_drt_amd/variants/libbdrt_$T.so; fi
  echo "== $T"
  WAVE_SWEEP=solo python tools/wave_sweep.py 4 2>&1 | grep -v amdgpu.ids
  WAVE_SWEEP=duo python tools/wave_sweep.py 512 2>&1 | grep -v amdgpu.ids
  WAVE_PROF=1 WAVE_SWEEP=solo python tools/wave_sweep.py 4 2>&1 | grep "cycles per"
done | tee $OUT/summary.txt
