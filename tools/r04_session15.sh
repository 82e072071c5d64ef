#!/bin/bash
# GPU session 15: packed-fp32 rows of the pipe model; the default bench line with the round's committed counter summary
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 200 build_ab/pipe_model > $O/pipe_model.txt 2>&1; head -16 $O/pipe_model.txt
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
