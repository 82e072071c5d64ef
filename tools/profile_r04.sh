#!/bin/bash
# usage: tools/profile_r04.sh <tag>   (on the GPU box) -> gpurun_out/<tag>/: rocprofv3 kernel stats, SQ + TCC counter passes (separate
# --pmc runs, never combined with other trace domains), bench JSON lines.  Copy the summaries into profiles/.
# A second argument selects a part (a gpurun call is limited to 20 minutes): `pmc` = the counter passes, `bench` = kernel stats, bench lines,
# pass statistics; none = both.
TAG=$1; PART=${2:-all}; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
if [ $PART = all ] || [ $PART = pmc ]; then
SQ_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU"
SQ_B="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD"
for C in 2 3 5 0; do
  SPP=64; [ $C = 0 ] && SPP=4
  for M in fresh replay; do
    rocprofv3 --pmc $SQ_A --kernel-trace --output-format csv -d $O/pmc_c${C}_${M}/a -- python3 tools/pmc_frame.py --config $C --mode $M > $O/pmc_c${C}_${M}.a.log 2>&1
    rocprofv3 --pmc $SQ_B --kernel-trace --output-format csv -d $O/pmc_c${C}_${M}/b -- python3 tools/pmc_frame.py --config $C --mode $M > $O/pmc_c${C}_${M}.b.log 2>&1
    python3 tools/pmc_summary.py $O/pmc_c${C}_${M}/a $O/pmc_c${C}_${M}/b --json $O/pmc_summary.json --key config${C}_spp${SPP}_gpus1$([ $M = replay ] && echo _replay) \
        --note "$([ $M = fresh ] && echo 'last launch of a history-free frame (spp >= 16: the main launch, samples [spp/16, spp), of a two-phase frame; else the one launch in image order)' || echo 'replay of an identical frame (one launch, all samples)'); tools/pmc_frame.py --config $C --mode $M" > $O/pmc_c${C}_${M}.summary.txt
    echo "pmc c$C $M done"
  done
  for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    N=$(echo $P | cut -d' ' -f1)
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/tcc_c${C}/$N -- python3 tools/pmc_frame.py --config $C --mode fresh > $O/tcc_c${C}.$N.log 2>&1
  done
  python3 tools/pmc_summary.py $O/tcc_c${C}/FETCH_SIZE $O/tcc_c${C}/WRITE_SIZE $O/tcc_c${C}/TCC_HIT_sum $O/tcc_c${C}/TCP_TCC_READ_REQ_sum > $O/tcc_c${C}.summary.txt
  echo "tcc c$C done"
done
fi
if [ $PART = all ] || [ $PART = bench ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strong --no-single-process --no-target --no-reference-default > $O/bench_stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --steps 6 --warmup 2 --config 3 --no-cpu-baseline --no-strong --no-single-process > $O/bench_stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 bench.py --steps 6 --warmup 2 --config 5 --no-cpu-baseline --no-strong --no-single-process > $O/bench_stats_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c0 -- python3 tools/demo_time.py 100 > $O/bench_stats_c0.log 2>&1
echo "stats done"
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench.err
python3 bench.py --steps 6 --warmup 2 --config 3 --no-cpu-baseline --no-strong --no-single-process > $O/bench_c3.json 2>> $O/bench.err
python3 bench.py --steps 6 --warmup 2 --config 5 --no-cpu-baseline --no-strong --no-single-process > $O/bench_c5.json 2>> $O/bench.err
python3 bench.py --steps 2 --warmup 1 --config 5 --passes 16 --no-strong --no-single-process > $O/bench_c5_progressive.json 2>> $O/bench.err
python3 bench.py --steps 4 --warmup 1 --config 4 --no-cpu-baseline --no-single-process > $O/bench_c4_8k.json 2>> $O/bench.err
python3 tools/demo_time.py 100 > $O/demo_time.txt 2>&1
# pass statistics of the product kernels (a -DTDT_STATS build of the same sources: build_ab/lib_stats.so) for the loss budget
for C in 2 3 5 0; do for M in fresh replay; do
  TDT_LIB=$PWD/build_ab/lib_stats.so TDT_STATS_SKIP_PROBE=1 timeout -k 10 300 python3 tools/loss_budget.py collect --config $C --mode $M > $O/stats_c${C}_${M}.json 2> $O/stats_c${C}_${M}.err
done; done
echo "pass statistics done"
fi
ls $O
