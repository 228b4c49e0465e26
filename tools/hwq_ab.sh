#!/bin/bash
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for B in 1 2 8 16 64; do for Q in 2 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --streams $B --steps $((B>=32?30:150)) --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $B GPU_MAX_HW_QUEUES=$Q frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done
