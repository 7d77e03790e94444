cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_hessian.py -x -q -m gpu -s > gpurun_out/r06c/pytest_hessian.txt 2>&1
tail -25 gpurun_out/r06c/pytest_hessian.txt
python tools/map_timing.py > gpurun_out/r06c/map_timing.txt 2>&1; tail -14 gpurun_out/r06c/map_timing.txt
python -m pytest tests/test_gpu_big.py -x -q -m gpu -k "beyond" > gpurun_out/r06c/pytest_big.txt 2>&1; tail -5 gpurun_out/r06c/pytest_big.txt
