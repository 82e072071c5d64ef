#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box) -> gpurun_out/<tag>/{stats,pmc,traffic}
TAG=$1; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python bench.py --steps 20 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_stats.log 2>&1
tools/experiments/pmc.sh $TAG/pmc > /dev/null 2>&1
tools/traffic.sh $TAG/traffic > /dev/null 2>&1
tools/traffic.sh $TAG/traffic_c3 --config 3 > /dev/null 2>&1
tools/traffic.sh $TAG/traffic_c5 --config 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_c3 -- python bench.py --steps 8 --warmup 1 --no-cpu-baseline --config 3 > gpurun_out/$TAG/bench_stats_c3.log 2>&1
python bench.py --steps 10 --warmup 2 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python bench.py --steps 5 --warmup 2 --config 3 --no-cpu-baseline > gpurun_out/$TAG/bench_c3.json 2>> gpurun_out/$TAG/bench.err
python bench.py --steps 5 --warmup 2 --config 5 --no-cpu-baseline > gpurun_out/$TAG/bench_c5.json 2>> gpurun_out/$TAG/bench.err
python bench.py --steps 2 --warmup 1 --config 5 --passes 16 > gpurun_out/$TAG/bench_c5_progressive.json 2>> gpurun_out/$TAG/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_c5 -- python bench.py --steps 8 --warmup 1 --no-cpu-baseline --config 5 > gpurun_out/$TAG/bench_stats_c5.log 2>&1
ls gpurun_out/$TAG
