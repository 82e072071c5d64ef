#!/bin/bash
# GPU session 25: kernel stats, bench lines and pass statistics of the final kernels (tools/profile_r04.sh bench part)
bash tools/profile_r04.sh r04y bench > gpurun_out/r04y.bench.log 2>&1; tail -5 gpurun_out/r04y.bench.log
