#!/bin/bash
# A/B of the pipelined mode's CU partition at 7-16 streams, fp32 and fp16 AR
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for DT in ${DTS:-0 1}; do for B in ${BS:-7 8 12 16}; do for P in 0 1; do
  SVA_DEBUG=cu_partition=$P timeout 300 python bench.py --ar-dtype $DT --streams $B --steps 60 --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ar_dtype $DT streams $B partition $P frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done; done
