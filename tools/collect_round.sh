#!/bin/bash
# usage (on the GPU box): tools/collect_round.sh <tag>   -- the bench lines, rocprofv3 kernel stats and PMC passes that go under
# profiles/ for a round, written to gpurun_out/<tag>_*
set -e
t=$1
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
python bench.py --conv-algo direct --no-cpu-baseline > gpurun_out/${t}_bench_direct.json 2>> gpurun_out/${t}_bench.err
python bench.py --config 2 --no-cpu-baseline > gpurun_out/${t}_bench_config2.json 2>> gpurun_out/${t}_bench.err
python bench.py --batch 8 --horizon 5 --no-cpu-baseline > gpurun_out/${t}_bench_config0_shape.json 2>> gpurun_out/${t}_bench.err
python bench.py --size 84 --batch 128 --actions 4 --rewards 1 --no-cpu-baseline --steps 10 > gpurun_out/${t}_bench_84.json 2>> gpurun_out/${t}_bench.err
python bench.py --size 128 --channels 4 --actions 4 --horizon 20 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/${t}_bench_128.json 2>> gpurun_out/${t}_bench.err
python bench.py --precision fp16 --no-cpu-baseline > gpurun_out/${t}_bench_fp16.json 2>> gpurun_out/${t}_bench.err
python bench.py --size 128 --channels 4 --actions 4 --horizon 20 --precision fp16 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/${t}_bench_cfg5_fp16.json 2>> gpurun_out/${t}_bench.err
echo "bench lines done"
tools/rocprof_stats.sh ${t}_stats bench.py --profile-mode --steps 8 --warmup 2 > gpurun_out/${t}_stats_summary.txt
echo "kernel stats done"
tools/pmc_collect.sh ${t}_pmc
python3 tools/pmc_parse.py gpurun_out/${t}_pmc > gpurun_out/${t}_pmc.json
echo "pmc done"
