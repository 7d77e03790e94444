#!/bin/bash
# hyper-lambda iterations of the ridge starting point (BDRT_RIDGE_START_ITER, default 3): fit time and what the fits find
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06q
for it in 3 2 1; do
  export BDRT_RIDGE_START_ITER=$it
  echo "== BDRT_RIDGE_START_ITER=$it"
  timeout 300 python tools/map_timing.py 2>&1 | grep "^K=\|starts" | cut -c1-175
  timeout 600 python -m tests.fuzz_inverter --first 4000 --count 200 2>&1 | tail -1
  timeout 300 python tools/map_suite_many.py 2>&1 | grep -v amdgpu.ids | tail -2
done > gpurun_out/r06q/ridge_start_iter.txt 2>&1
cat gpurun_out/r06q/ridge_start_iter.txt
