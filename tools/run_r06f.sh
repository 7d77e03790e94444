cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06f
timeout 120 python -m pytest tests/test_gpu_hessian.py -x -q -m gpu -s 2>&1 | tail -5 > gpurun_out/r06f/pytest_hessian.txt; cat gpurun_out/r06f/pytest_hessian.txt
grep -q "passed" gpurun_out/r06f/pytest_hessian.txt && ! grep -q "failed" gpurun_out/r06f/pytest_hessian.txt || exit 1
BDRT_NEWTON_PROF=1 timeout 100 python tools/map_single_trace.py 161 > gpurun_out/r06f/newton_prof_161.txt 2>&1; grep "newton prof\] [Dc]" gpurun_out/r06f/newton_prof_161.txt | tail -2
BDRT_NEWTON_PROF=1 timeout 100 python tools/map_single_trace.py 81 > gpurun_out/r06f/newton_prof_81.txt 2>&1; grep "newton prof\] [Dc]" gpurun_out/r06f/newton_prof_81.txt | tail -2
timeout 200 python tools/map_timing.py 2>&1 | grep -E "fit\(mode" 
BDRT_NEWTON_FD=1 timeout 200 python tools/map_timing.py 2>&1 | grep -E "fit\(mode"
timeout 100 python tools/map_batch_timing.py 2>&1 | grep spectra
timeout 100 python tools/map_suite_many.py 2>&1 | tail -1
