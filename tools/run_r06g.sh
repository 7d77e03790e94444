cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06g
timeout 120 python -m pytest tests/test_gpu_hessian.py -x -q -m gpu -s 2>&1 | tail -5 > gpurun_out/r06g/pytest_hessian.txt; cat gpurun_out/r06g/pytest_hessian.txt
grep -q "passed" gpurun_out/r06g/pytest_hessian.txt && ! grep -q "failed" gpurun_out/r06g/pytest_hessian.txt || exit 1
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fit_many.py tests/test_gpu_inverter.py tests/test_gpu_hmc_reference.py tests/test_gpu_big.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -q -m gpu -k "map or optimize or fit_many or inverter or fuzz or edge or big" > gpurun_out/r06g/pytest_map.txt 2>&1; tail -8 gpurun_out/r06g/pytest_map.txt
timeout 200 python tools/map_timing.py 2>&1 | grep -E "fit\(mode|starts \(random" | tee gpurun_out/r06g/map_timing.txt
BDRT_NEWTON_PROF=1 timeout 100 python tools/map_single_trace.py 161 > gpurun_out/r06g/newton_prof_161.txt 2>&1; grep "newton prof\] [Dc]" gpurun_out/r06g/newton_prof_161.txt | tail -2
