#!/bin/bash
# GPU session 24: counter passes of the final kernels (tools/profile_r04.sh pmc part)
bash tools/profile_r04.sh r04y pmc > gpurun_out/r04y.pmc.log 2>&1; tail -5 gpurun_out/r04y.pmc.log
