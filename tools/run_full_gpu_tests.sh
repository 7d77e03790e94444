cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
python -m pytest tests/ -q -m gpu > gpurun_out/r06/pytest_gpu.txt 2>&1
tail -12 gpurun_out/r06/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
