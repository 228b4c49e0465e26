#!/bin/bash
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for DT in ${DTS:-0}; do for B in ${BS:-8 16 24 32}; do for N in ${NS:-64 96 128}; do
  SVA_DEBUG=cu_partition=1,cu_ar=$N timeout 300 python bench.py --ar-dtype $DT --streams $B --steps 60 --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ar_dtype $DT streams $B AR CUs $N frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done; done
