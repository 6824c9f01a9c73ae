#!/bin/bash
# GPU session: GPU tests, then tools/gpu_perf.py (final_scene at 1024 spp with kernel timings, the small scenes).
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03h}
O=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_gputests.log 2>&1 || { tail -30 $O/${T}_gputests.log; exit 1; }
tail -3 $O/${T}_gputests.log
timeout -k 10 300 python tools/gpu_perf.py 1024 0,16 > $O/${T}_gpu_perf.txt 2>&1 || { tail $O/${T}_gpu_perf.txt; exit 1; }
cat $O/${T}_gpu_perf.txt
