#!/bin/bash
# GPU session: the GPU test suite, bench.py as one RCCL rank under the launcher, a two-rank rehearsal on one GPU (gloo), the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-r03a}
O=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_gputests.log 2>&1 || { tail -30 $O/${T}_gputests.log; exit 1; }
tail -3 $O/${T}_gputests.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --spp 512 > $O/${T}_bench_rccl_one_rank.json.log 2> $O/${T}_bench_rccl_one_rank.err || { tail -30 $O/${T}_bench_rccl_one_rank.err; exit 1; }
tail -c 600 $O/${T}_bench_rccl_one_rank.json.log
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --spp 512 > $O/${T}_rehearsal_two_ranks_one_gpu.json.log 2> $O/${T}_rehearsal_two_ranks_one_gpu.err || { tail -30 $O/${T}_rehearsal_two_ranks_one_gpu.err; exit 1; }
tail -c 600 $O/${T}_rehearsal_two_ranks_one_gpu.json.log
timeout -k 10 400 python bench.py > $O/${T}_bench_final_scene.json.log 2> $O/${T}_bench_final_scene.err || { tail -30 $O/${T}_bench_final_scene.err; exit 1; }
cat $O/${T}_bench_final_scene.json.log
