#!/bin/bash
N=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/$N/c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/$N.c.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VSKIPPED --kernel-trace --output-format csv -d gpurun_out/$N/d -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/$N.d.log 2>&1
python tools/pmc_summary.py gpurun_out/$N/c gpurun_out/$N/d > gpurun_out/$N.summary2.txt; cat gpurun_out/$N.summary2.txt
