#!/bin/bash
# tools/converged_seeds.py once per library variant in raytracing_amd/csrc/variants/ (the regular library first).
set -u
cd "$(dirname "$0")/.."
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
for v in $lib raytracing_amd/csrc/variants/*.so; do
  [ -f "$v" ] || continue
  [ "$v" = "$lib" ] || cp "$v" $lib
  echo "== $(basename $v)"
  timeout -k 10 250 python tools/converged_seeds.py "$@" | tail -4 || break
  cp /tmp/librt_hip.keep.so $lib
done
cp /tmp/librt_hip.keep.so $lib
