#!/bin/bash
# GPU session 20: two-class hand-out order for thin histories (tiles in which a probe path hit something first): parity subset, then A/B
# against the single class (TDT_NO_HIT_CLASS=1) and with the probe split 1 + 3 samples (TDT_PREPROBE=1)
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -x -q > $O/parity.txt 2>&1; rc=$?; tail -3 $O/parity.txt
[ $rc -eq 0 ] && timeout -k 10 900 python3 tools/ab.py --reps 2 --out $O/ab_class.json "one_class|TDT_NO_HIT_CLASS=1|-" "two_classes||-" "two_classes_preprobe1|TDT_PREPROBE=1|-" "one_class_preprobe1|TDT_NO_HIT_CLASS=1 TDT_PREPROBE=1|-" > $O/ab_class.txt 2>&1; tail -14 $O/ab_class.txt
