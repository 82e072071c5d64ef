#!/bin/bash
# usage: tools/demo_profile.sh <tag>  (GPU box): the reference's own workload (tools/demo_time.py) under the schedule switches,
# then rocprofv3 kernel stats of it -> gpurun_out/<tag>/
TAG=$1; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
python3 tools/demo_time.py 100 > $O/default.txt 2>&1
TDT_TWO_PHASE_MIN_SPP=4 TDT_PROBE_DIV=4 python3 tools/demo_time.py 100 > $O/two_phase4.txt 2>&1
TDT_NO_COST_ORDER=1 python3 tools/demo_time.py 100 > $O/no_cost_order.txt 2>&1
TDT_NO_TABLE_FORM=1 python3 tools/demo_time.py 100 > $O/literal.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/demo_time.py 100 > $O/stats.log 2>&1
cat $O/default.txt $O/two_phase4.txt $O/no_cost_order.txt $O/literal.txt | grep demo
S=$(ls $O/stats/*/*kernel_stats.csv | tail -1); cut -d, -f1-6 $S | head -14
