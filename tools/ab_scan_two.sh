#!/bin/bash
# bench.py (no CPU baseline) on the scanned configurations with the first launch of a pass running one bounce (RT_SCAN_TWO=0)
# or two (2 = every scanned scene, 1 = the library's own choice), twice each, on one box.
set -u
cd "$(dirname "$0")/.."
for round in 1 2; do
  for two in 0 2 1; do
    for cfg in ${CONFIGS:-cornell perlin cornell_smoke}; do
      RT_SCAN_TWO=$two timeout -k 10 300 python bench.py --config $cfg --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('RT_SCAN_TWO=$two', d['config']['workload'][:34], d['value'], d['ms_per_step'], d['schedule'])
" || exit 1
    done
  done
done
