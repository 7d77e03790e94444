cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_fit_many.py tests/test_gpu_nccl.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r06a/pytest_new.txt 2>&1
tail -15 gpurun_out/r06a/pytest_new.txt
python tools/divergence_study.py hip --seeds 1234,1,2,3,4,5,6,7,8,9,10,11 --out gpurun_out/r06a/divergence_hip.npz > gpurun_out/r06a/divergence_hip.log 2>&1
tail -14 gpurun_out/r06a/divergence_hip.log
python tools/divergence_study.py outliers --seeds 1234,1,2,3 --out gpurun_out/r06a/divergence_outliers.npz > gpurun_out/r06a/divergence_outliers.log 2>&1
tail -6 gpurun_out/r06a/divergence_outliers.log
