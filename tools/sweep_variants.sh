#!/bin/bash
# Runs tools/gpu_perf.py once per library variant in raytracing_amd/csrc/variants/ (built by hand with -D overrides),
# each in a process of its own, and restores the regular library afterwards.  For parameter sweeps on the GPU box.
set -u
cd "$(dirname "$0")/.."
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
rc=0
for v in raytracing_amd/csrc/variants/*.so; do
    cp "$v" $lib
    line=$(timeout -k 10 200 python tools/gpu_perf.py ${1:-1024} ${2:-0} 2>&1 | grep -m1 flags) || { rc=1; echo "$(basename $v) failed"; break; }
    echo "$(basename $v .so) $line"
done
cp /tmp/librt_hip.keep.so $lib
exit $rc
