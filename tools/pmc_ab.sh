#!/bin/bash
# SQ counters of one small scene for every library variant: bash tools/pmc_ab.sh <scene> <w> <h> <spp>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
lib=raytracing_amd/csrc/librt_hip.so
cp $lib /tmp/librt_hip.keep.so
for v in raytracing_amd/csrc/variants/*.so; do
    cp "$v" $lib
    n=$(basename $v .so)
    rm -rf gpurun_out/pmc_ab/$n
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_ab/$n -o run --output-format csv -- python3 tools/pmc_one.py "$@" > gpurun_out/pmc_ab/$n.log 2>&1 || { echo "$n failed"; break; }
    echo "== $n"; python3 tools/pmc_sum.py gpurun_out/pmc_ab/$n
done
cp /tmp/librt_hip.keep.so $lib
