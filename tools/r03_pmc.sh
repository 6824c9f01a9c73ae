#!/bin/bash
# GPU session: counters (tools/pmc_collect.sh) for the given configurations, summarised into gpurun_out/<tag>_pmc.{json,md}.
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03j}; shift
rm -rf gpurun_out/pmc
bash tools/pmc_collect.sh "$@" > gpurun_out/${T}_pmc_collect.log 2>&1 || { tail -20 gpurun_out/${T}_pmc_collect.log; exit 1; }
python tools/pmc_summarise.py gpurun_out/pmc gpurun_out/${T}_pmc.json gpurun_out/${T}_pmc_summary.md
rm -rf gpurun_out/pmc/*/stats gpurun_out/pmc/*/sq gpurun_out/pmc/*/tcc gpurun_out/pmc/*/fetch gpurun_out/pmc/*/write
