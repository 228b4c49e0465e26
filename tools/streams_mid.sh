#!/bin/bash
# mid-range of the streams curve, both AR dtypes (default dispatch)
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for DT in 0 1; do for B in 7 8 12 16 24 32 48; do
  timeout 300 python bench.py --ar-dtype $DT --streams $B --steps 60 --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ar_dtype $DT streams $B frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done
