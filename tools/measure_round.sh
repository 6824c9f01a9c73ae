set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r02j
for cfg in final_scene cornell perlin cornell_smoke; do
  timeout -k 10 400 python bench.py --config $cfg --steps $([ $cfg = final_scene ] && echo 1 || echo 3) > gpurun_out/${T}_bench_$cfg.json.log 2> gpurun_out/${T}_bench_$cfg.err || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o run --output-format csv -- python3 bench.py --spp 104 --one-stream > gpurun_out/${T}_bench_spp104_one_stream.json.log 2> gpurun_out/${T}_prof.err || exit 1
bash tools/pmc_collect.sh > gpurun_out/${T}_pmc_collect.log 2>&1 || exit 1
timeout -k 10 400 python tools/split_emulation.py > gpurun_out/${T}_split_emulation.md 2> gpurun_out/${T}_split.err || exit 1
echo done
