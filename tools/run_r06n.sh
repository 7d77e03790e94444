#!/bin/bash
# the speculative second factorisation of the Newton solve: tests, timing with and without (BDRT_NEWTON_SPEC=0), A/B over random problems
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06n
timeout 900 python -m pytest tests/test_gpu_hessian.py tests/test_gpu_engine.py tests/test_gpu_fit_many.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r06n/pytest.txt
cat gpurun_out/r06n/pytest.txt
grep -q failed gpurun_out/r06n/pytest.txt && exit 1
for sp in 1 0; do
  echo "== BDRT_NEWTON_SPEC=$sp"
  BDRT_NEWTON_SPEC=$sp timeout 600 python tools/map_timing.py 2>&1 | grep "^K=\|starts\|alone"
done > gpurun_out/r06n/map_timing_spec.txt 2>&1
cat gpurun_out/r06n/map_timing_spec.txt | cut -c1-170
