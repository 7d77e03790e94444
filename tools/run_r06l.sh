#!/bin/bash
# the one-wave-per-SIMD schedule of the one-chain-per-wave kernel (OCC = 1 instantiations) against the default, same library (BDRT_WAVE_OCC forces one)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06l
for occ in "" 1 2; do
  echo "== BDRT_WAVE_OCC=$occ: tests/test_gpu_wave.py"
  BDRT_WAVE_OCC=$occ timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -m gpu 2>&1 | tail -2
done > gpurun_out/r06l/pytest_wave_occ.txt 2>&1
cat gpurun_out/r06l/pytest_wave_occ.txt
for occ in 2 ""; do
  echo "== BDRT_WAVE_OCC=$occ"
  BDRT_WAVE_OCC=$occ WAVE_SWEEP=wave timeout 600 python tools/wave_sweep.py 4 256 512 768 1024 1536 2048 2>&1 | grep -v amdgpu.ids
  BDRT_WAVE_OCC=$occ WAVE_SWEEP=wave WAVE_PROF=1 timeout 300 python tools/wave_sweep.py 4 1024 2>&1 | grep -v amdgpu.ids | grep cycles
done > gpurun_out/r06l/wave_occ.txt 2>&1
cat gpurun_out/r06l/wave_occ.txt
