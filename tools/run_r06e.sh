cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_hessian.py -x -q -m gpu -s > gpurun_out/r06e/pytest_hessian.txt 2>&1
tail -6 gpurun_out/r06e/pytest_hessian.txt
echo "--- schur"; python tools/map_timing.py 2>&1 | grep -E "fit\(mode|random start alone" > gpurun_out/r06e/map_timing_schur.txt; cat gpurun_out/r06e/map_timing_schur.txt
echo "--- dense solve"; BDRT_NEWTON_SCHUR=0 python tools/map_timing.py 2>&1 | grep -E "fit\(mode" > gpurun_out/r06e/map_timing_dense.txt; cat gpurun_out/r06e/map_timing_dense.txt
echo "--- fd"; BDRT_NEWTON_FD=1 python tools/map_timing.py 2>&1 | grep -E "fit\(mode" > gpurun_out/r06e/map_timing_fd.txt; cat gpurun_out/r06e/map_timing_fd.txt
BDRT_NEWTON_PROF=1 python tools/map_single_trace.py 161 > gpurun_out/r06e/newton_prof_161.txt 2>&1; grep "newton prof\] D" gpurun_out/r06e/newton_prof_161.txt | tail -1
BDRT_NEWTON_PROF=1 python tools/map_single_trace.py 81 > gpurun_out/r06e/newton_prof_81.txt 2>&1; grep "newton prof\] D" gpurun_out/r06e/newton_prof_81.txt | tail -1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r06e/trace -o map -- python3 tools/map_single_trace.py 161 > gpurun_out/r06e/trace.log 2>&1
DB=$(find gpurun_out/r06e/trace -name '*.db' | head -1)
python3 tools/rocpd_summary.py "$DB" > gpurun_out/r06e/kernel_stats_map161.txt
head -12 gpurun_out/r06e/kernel_stats_map161.txt | cut -c1-150
find gpurun_out/r06e -name '*.db' -delete
python tools/map_suite_many.py > gpurun_out/r06e/map_suite_many.txt 2>&1; tail -2 gpurun_out/r06e/map_suite_many.txt
python tools/map_batch_timing.py > gpurun_out/r06e/map_batch.txt 2>&1; grep spectra gpurun_out/r06e/map_batch.txt
