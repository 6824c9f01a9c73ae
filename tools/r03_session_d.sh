#!/bin/bash
# GPU session: the GPU test suite, then the shading-wave statistics of the analysis build (variants/shadestats.so swapped in for the run).
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03g}
O=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_gputests.log 2>&1 || { tail -30 $O/${T}_gputests.log; exit 1; }
tail -3 $O/${T}_gputests.log
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/keep.so && cp raytracing_amd/csrc/variants/shadestats.so $lib
timeout -k 10 200 python tools/shade_waves.py final_scene > $O/${T}_shade_waves_final_scene.txt 2>&1; rc=$?
cp /tmp/keep.so $lib
cat $O/${T}_shade_waves_final_scene.txt
exit $rc
