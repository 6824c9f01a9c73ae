#!/bin/bash
# tools/small_perf.py once per library variant in raytracing_amd/csrc/variants/, each in a process of its own; restores the regular library.
set -u
cd "$(dirname "$0")/.."
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
rc=0
for v in raytracing_amd/csrc/variants/*.so; do
    cp "$v" $lib
    line=$(timeout -k 10 240 python tools/small_perf.py ${1:-0.25} 2>&1 | tail -1) || { rc=1; echo "$(basename $v) failed: $line"; break; }
    echo "$(basename $v .so) $line"
done
cp /tmp/librt_hip.keep.so $lib
exit $rc
