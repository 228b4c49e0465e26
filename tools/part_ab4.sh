#!/bin/bash
# chunk = 4 batches: default rule vs no partition
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for B in 8 16 32; do for CFG in "" "cu_partition=0"; do
  SVA_DEBUG=$CFG timeout 300 python bench.py --streams $B --chunk 4 --steps 30 --warmup 4 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk 4 streams $B [$CFG] frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done
