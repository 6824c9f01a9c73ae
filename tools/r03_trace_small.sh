#!/bin/bash
# GPU session: kernel trace of one profiled pass of a scanned scene; prints the per-launch durations in dispatch order.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=${1:-perlin}; spp=${2:-64}
rm -rf gpurun_out/trace_$cfg
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_$cfg -o run --output-format csv -- python3 bench.py --config $cfg --spp $spp --one-stream --no-cpu-baseline > gpurun_out/trace_$cfg.log 2>&1 || { tail gpurun_out/trace_$cfg.log; exit 1; }
python3 - gpurun_out/trace_$cfg <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = []
for r in rows:
    n = r["Kernel_Name"]
    short = "shade" if "k_shade" in n else "traverse" if "k_traverse" in n else "tail" if "k_tail" in n else n.split("(")[0][-24:]
    out.append("%10.1f us  +%8.1f us  %-10s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
print("\n".join(out[-140:]))
PY
rm -rf gpurun_out/trace_$cfg
