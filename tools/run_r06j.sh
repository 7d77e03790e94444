#!/bin/bash
# trace of the two starts of the K = 161 / 101 / 81 MAP fits (status records at every host synchronisation)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06j
BDRT_NEWTON_TRACE=1 BDRT_NEWTON_ROUNDS=1 timeout 600 python tools/map_timing.py > gpurun_out/r06j/map_trace.txt 2>&1
tail -5 gpurun_out/r06j/map_trace.txt
