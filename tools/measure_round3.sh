#!/bin/bash
# GPU session: every measurement that is reported (tests, counters, bench lines, kernel trace, rank rehearsals, split emulation), on one build.  Tag = first argument.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r03o}
O=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_gputests.log 2>&1 || { tail -30 $O/${T}_gputests.log; exit 1; }
tail -2 $O/${T}_gputests.log
for cfg in final_scene cornell perlin cornell_smoke; do
  timeout -k 10 400 python bench.py --config $cfg --steps $([ $cfg = final_scene ] && echo 1 || echo 3) > $O/${T}_bench_$cfg.json.log 2> $O/${T}_bench_$cfg.err || { tail -20 $O/${T}_bench_$cfg.err; exit 1; }
done
PS=$(python -c "print((128 << 20) // 640000)")
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o run --output-format csv -- python3 bench.py --spp $PS --one-stream --no-cpu-baseline > $O/${T}_bench_spp${PS}_one_stream.json.log 2> $O/${T}_prof.err || { tail -20 $O/${T}_prof.err; exit 1; }
find $O/${T}_prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats_bench_spp${PS}_one_stream.csv \;
rm -rf $O/${T}_prof
bash tools/r03_pmc.sh $T || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --spp 1024 > $O/${T}_bench_rccl_one_rank.json.log 2> $O/${T}_bench_rccl_one_rank.err || { tail -30 $O/${T}_bench_rccl_one_rank.err; exit 1; }
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --spp 1024 > $O/${T}_rehearsal_two_ranks_one_gpu.json.log 2> $O/${T}_rehearsal_two_ranks_one_gpu.err || { tail -30 $O/${T}_rehearsal_two_ranks_one_gpu.err; exit 1; }
timeout -k 10 500 python tools/split_emulation.py > $O/${T}_split_emulation.md 2> $O/${T}_split.err || { tail $O/${T}_split.err; exit 1; }
cat $O/${T}_split_emulation.md
echo done
