#!/bin/bash
# bench.py on the three small configurations, once per library variant in raytracing_amd/csrc/variants/ (and the regular
# library first); restores the regular library afterwards.  For A/B runs on the GPU box.
set -u
cd "$(dirname "$0")/.."
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
rc=0
for v in $lib raytracing_amd/csrc/variants/*.so; do
  [ -f "$v" ] || continue
  [ "$v" = "$lib" ] || cp "$v" $lib
  for cfg in ${CONFIGS:-cornell perlin cornell_smoke}; do
    timeout -k 10 300 python bench.py --config $cfg --steps 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$(basename $v)', d['config']['workload'][:30], d['value'], d['ms_per_step'])
" || { rc=1; break 2; }
  done
  cp /tmp/librt_hip.keep.so $lib
done
cp /tmp/librt_hip.keep.so $lib
exit $rc
