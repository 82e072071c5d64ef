#!/bin/bash
# GPU session 32: the event threshold's clamp after the stuck-ray cut (steps per ray fell, the clamp binds more often)
O=gpurun_out/r04ae; mkdir -p $O
timeout -k 10 900 python3 tools/ab.py --reps 2 --out $O/ab_clamp.json "clamp40||-" "clamp32|TDT_EVENT_CLAMP=32|-" "clamp48|TDT_EVENT_CLAMP=48|-" "clamp56|TDT_EVENT_CLAMP=56|-" > $O/ab_clamp.txt 2>&1; tail -14 $O/ab_clamp.txt
