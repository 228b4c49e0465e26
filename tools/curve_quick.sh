#!/bin/bash
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for B in ${BS:-1 2 4 6 8 12 16 24 32 64}; do
  timeout 300 python bench.py --ar-dtype ${DT:-0} --streams $B --steps $((B>=32?30:100)) --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ar_dtype ${DT:-0} streams $B frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done
