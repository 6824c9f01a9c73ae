#!/bin/bash
# GPU session: kernel trace of a short timed (two-stream) final_scene run; prints a summary of the timeline of its last frame:
# busy time per kernel, time covered by 0 / 1 / 2 kernels at once, and the largest gaps.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
spp=${1:-627}
rm -rf gpurun_out/trace_tl
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_tl -o run --output-format csv -- python3 bench.py --spp $spp --no-cpu-baseline --warmup 1 > gpurun_out/trace_tl.log 2>&1 || { tail gpurun_out/trace_tl.log; exit 1; }
python3 - gpurun_out/trace_tl <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return "shade" if "k_shade" in n else "traverse" if "k_traverse" in n else "tail" if "k_tail" in n else "raygen" if "k_raygen" in n else "resolve" if "k_resolve" in n else "finalize" if "k_finalize" in n else "other"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
fins = [i for i, e in enumerate(ev) if e[2] == "finalize"]
# the timed frame: between the second-to-last finalize (warm-up's... the profile render comes last) -- take the frame with the most kernels
frames, prev = [], 0
for i in fins:
    frames.append(ev[prev:i + 1]); prev = i + 1
frames.sort(key=len)
fr = frames[-1]
t0, t1 = fr[0][0], max(e[1] for e in fr)
print("frame: %d kernels, %.3f ms" % (len(fr), (t1 - t0) / 1e6))
busy = {}
for s, e, k in fr: busy[k] = busy.get(k, 0) + (e - s)
for k, v in sorted(busy.items(), key=lambda x: -x[1]): print("  %-9s %9.3f ms summed duration, %d launches" % (k, v / 1e6, sum(1 for x in fr if x[2] == k)))
pts = sorted([(s, 1, k) for s, e, k in fr] + [(e, -1, k) for s, e, k in fr])
cover, depth, last = {}, 0, t0
active = {}
combo = {}
for t, d, k in pts:
    key = "+".join(sorted(x for x, c in active.items() if c > 0)) or "idle"
    combo[key] = combo.get(key, 0) + (t - last)
    active[k] = active.get(k, 0) + d
    last = t
for k, v in sorted(combo.items(), key=lambda x: -x[1])[:12]: print("  %-28s %9.3f ms (%.1f %%)" % (k, v / 1e6, 100.0 * v / (t1 - t0)))
PY
rm -rf gpurun_out/trace_tl
