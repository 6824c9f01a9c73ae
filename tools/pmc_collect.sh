#!/bin/bash
# Collects the rocprofv3 counters of the wavefront kernels on the four GPU configurations, one run per counter group
# (a --pmc run is never combined with a trace), into gpurun_out/pmc/<config>/<group>/.  Run on the GPU box from the repo root:
#     bash tools/pmc_collect.sh [config ...]
# then summarise with tools/pmc_summarise.py gpurun_out/pmc profiles/<tag>_pmc.json profiles/<tag>_pmc_summary.md
set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
OUT=gpurun_out/pmc
CONFIGS=${@:-cornell perlin cornell_smoke final_scene}
declare -A SPP=( [cornell]=32 [perlin]=6 [cornell_smoke]=16 [final_scene]=16 )
for cfg in $CONFIGS; do
  spp=${SPP[$cfg]}
  mkdir -p $OUT/$cfg
  timeout -k 10 120 python3 tools/prof_workload.py $cfg $spp $OUT/$cfg/workload.json || exit 1
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $OUT/$cfg/stats -o run --output-format csv -- python3 tools/prof_workload.py $cfg $spp > $OUT/$cfg/stats.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/$cfg/sq -o run --output-format csv -- python3 tools/prof_workload.py $cfg $spp > $OUT/$cfg/sq.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/$cfg/tcc -o run --output-format csv -- python3 tools/prof_workload.py $cfg $spp > $OUT/$cfg/tcc.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/$cfg/fetch -o run --output-format csv -- python3 tools/prof_workload.py $cfg $spp > $OUT/$cfg/fetch.log 2>&1 || exit 1
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $OUT/$cfg/write -o run --output-format csv -- python3 tools/prof_workload.py $cfg $spp > $OUT/$cfg/write.log 2>&1 || exit 1
  echo "$cfg done"
done
