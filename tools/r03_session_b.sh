#!/bin/bash
# GPU session: the graph-capture test of the no-allocation entry point, then the two-rank rehearsal with gloo handed the device
# tensor itself (the path that hung), under a watchdog that dumps every thread's Python stack after 60 s and exits.
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03b}
O=gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_graph_capture.py -x -q --timeout 120 --timeout-method thread > $O/${T}_graph_capture.log 2>&1
echo "capture rc=$?"; tail -25 $O/${T}_graph_capture.log
RT_GLOO_DEVICE_TENSORS=1 BENCH_WATCHDOG_S=60 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/bench_debug.py --gpus 2 --rehearse-on-one-gpu --spp 64 --no-cpu-baseline > $O/${T}_gloo_device_tensors.log 2>&1
echo "gloo rc=$?"; tail -60 $O/${T}_gloo_device_tensors.log
