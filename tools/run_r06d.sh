cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_hessian.py -x -q -m gpu -s > gpurun_out/r06d/pytest_hessian.txt 2>&1
tail -8 gpurun_out/r06d/pytest_hessian.txt
echo "--- schur"; python tools/map_timing.py 2>&1 | grep -E "fit\(mode|random start alone" > gpurun_out/r06d/map_timing_schur.txt; cat gpurun_out/r06d/map_timing_schur.txt
echo "--- dense solve"; BDRT_NEWTON_SCHUR=0 python tools/map_timing.py 2>&1 | grep -E "fit\(mode|random start alone" > gpurun_out/r06d/map_timing_dense.txt; cat gpurun_out/r06d/map_timing_dense.txt
echo "--- fd"; BDRT_NEWTON_FD=1 python tools/map_timing.py 2>&1 | grep -E "fit\(mode|random start alone" > gpurun_out/r06d/map_timing_fd.txt; cat gpurun_out/r06d/map_timing_fd.txt
BDRT_NEWTON_PROF=1 python tools/map_single_trace.py 161 > gpurun_out/r06d/newton_prof_161.txt 2>&1; grep "newton prof" gpurun_out/r06d/newton_prof_161.txt | tail -3
BDRT_NEWTON_PROF=1 python tools/map_single_trace.py 81 > gpurun_out/r06d/newton_prof_81.txt 2>&1; grep "newton prof" gpurun_out/r06d/newton_prof_81.txt | tail -3
python -m pytest tests/test_gpu_engine.py tests/test_gpu_fit_many.py tests/test_gpu_inverter.py -x -q -m gpu -k "map or optimize or fit_many or inverter" > gpurun_out/r06d/pytest_map.txt 2>&1; tail -8 gpurun_out/r06d/pytest_map.txt
python tools/map_suite_many.py > gpurun_out/r06d/map_suite_many.txt 2>&1; tail -3 gpurun_out/r06d/map_suite_many.txt
