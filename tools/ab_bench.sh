#!/bin/bash
# bench.py (no CPU baseline) on the given configurations, once with the regular library and once per variant in
# raytracing_amd/csrc/variants/; restores the regular library afterwards.  For A/B runs on the GPU box.
set -u
cd "$(dirname "$0")/.."
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
rc=0
for v in $lib raytracing_amd/csrc/variants/*.so; do
  [ -f "$v" ] || continue
  [ "$v" = "$lib" ] || cp "$v" $lib
  for cfg in ${CONFIGS:-final_scene cornell perlin cornell_smoke}; do
    timeout -k 10 300 python bench.py --config $cfg --steps $([ $cfg = final_scene ] && echo 1 || echo 3) --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$(basename $v)', d['config']['workload'][:30], d['value'], d['ms_per_step'], d['schedule'])
" || { rc=1; break 2; }
  done
  cp /tmp/librt_hip.keep.so $lib
done
cp /tmp/librt_hip.keep.so $lib
exit $rc
