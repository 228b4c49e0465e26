#!/bin/bash
# 3-6 streams: persistent kernel (default) vs multi-launch decode on a CU partition
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for DT in ${DTS:-0 1}; do for B in 3 4 5 6; do for CFG in "" "ar_persistent=0,cu_partition=1,cu_ar=128" "ar_persistent=0,cu_partition=1,cu_ar=96"; do
  SVA_DEBUG=$CFG timeout 300 python bench.py --ar-dtype $DT --streams $B --steps 80 --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ar_dtype $DT streams $B [$CFG] frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done; done
