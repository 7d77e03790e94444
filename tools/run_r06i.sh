cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06i
timeout 200 python -m pytest tests/test_gpu_hessian.py tests/test_gpu_fit_many.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/map_timing.py 2>&1 | grep -E "fit\(mode"
timeout 1200 python tools/map_scale_ab.py 4000 400 > gpurun_out/r06i/map_scale_ab.txt 2>&1; tail -4 gpurun_out/r06i/map_scale_ab.txt
