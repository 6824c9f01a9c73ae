#!/bin/bash
# GPU session: bench.py on the four GPU configurations (final_scene once, the others three steps each).
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03i}
O=gpurun_out
for cfg in ${CONFIGS:-final_scene cornell perlin cornell_smoke}; do
  timeout -k 10 400 python bench.py --config $cfg --steps $([ $cfg = final_scene ] && echo 1 || echo 3) > $O/${T}_bench_$cfg.json.log 2> $O/${T}_bench_$cfg.err || { tail -20 $O/${T}_bench_$cfg.err; exit 1; }
  python - $O/${T}_bench_$cfg.json.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(d["config"]["workload"][:34], d["value"], "Msamples/s", d["ms_per_step"], "ms;", r["kernel"], r["bound"], "frac", r["frac"], r.get("valu_frac"), "kernel_ms", r["kernel_ms"], "cpu", d["cpu_baseline"]["value"])
PY
done
