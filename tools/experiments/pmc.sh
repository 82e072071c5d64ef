#!/bin/bash
# usage: tools/experiments/pmc.sh <outdir-name>  (runs two PMC passes of one bench step; summaries under gpurun_out/<name>)
N=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/$N/a -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$N.a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/$N/b -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$N.b.log 2>&1 &&
python tools/pmc_summary.py gpurun_out/$N/a gpurun_out/$N/b > gpurun_out/$N.summary.txt; tail -3 gpurun_out/$N.summary.txt
