cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests/test_gpu_wave.py tests/test_gpu_solo.py tests/test_gpu_solo_wide.py tests/test_gpu_config4.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/r06h/pytest_samplers.txt 2>&1; tail -4 gpurun_out/r06h/pytest_samplers.txt
grep -q "passed" gpurun_out/r06h/pytest_samplers.txt && ! grep -q "failed" gpurun_out/r06h/pytest_samplers.txt || exit 1
python tools/wave_sweep.py 512 1024 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h/wave_sweep.txt
for n in 512 768 1024; do python tools/bench_config5.py $n 2>&1 | grep -v amdgpu.ids | head -1; python tools/bench_config5.py $n --series-outliers 2>&1 | grep -v amdgpu.ids | head -1; done | tee gpurun_out/r06h/wave_om.txt
