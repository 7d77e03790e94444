cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_map
python3 tools/map_batch_timing.py 2>&1 | grep spectra
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_map/trace -o map -- python3 tools/map_batch_timing.py > gpurun_out/prof_map/trace.log 2>&1
DB=$(find gpurun_out/prof_map/trace -name '*.db' | head -1)
python3 tools/rocpd_summary.py "$DB" > gpurun_out/prof_map/kernel_stats.txt
head -14 gpurun_out/prof_map/kernel_stats.txt | cut -c1-170
find gpurun_out/prof_map -name '*.db' -size +20M -delete
