#!/bin/bash
# HBM-side traffic of one bench step (separate PMC passes: FETCH_SIZE uses 3 of 4 TCC slots, WRITE_SIZE 2)
N=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$N/f -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$N.f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$N/w -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$N.w.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$N/h -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$N.h.log 2>&1
python tools/pmc_summary.py gpurun_out/$N/f gpurun_out/$N/w gpurun_out/$N/h > gpurun_out/$N.summary.txt; cat gpurun_out/$N.summary.txt
