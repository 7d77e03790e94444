cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_fit_many.py tests/test_gpu_hmc_reference.py -q -m gpu -s > gpurun_out/r06b/pytest_fit_hmc.txt 2>&1
tail -5 gpurun_out/r06b/pytest_fit_hmc.txt
python -m pytest tests/test_gpu_nccl.py tests/test_gpu_bench.py -q -m gpu > gpurun_out/r06b/pytest_ranks.txt 2>&1
tail -15 gpurun_out/r06b/pytest_ranks.txt
