#!/bin/bash
# Which chains stretch which: the single-stream pipelined period with chains left out (timing only; outputs are garbage)
X="--no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline"
for M in 0 1 2 4 8 3 12 5 10 9 6 7 11 13 14; do
  SVA_DEBUG=pipe_skip=$M timeout 300 python bench.py --streams ${B:-1} --steps 150 --warmup 5 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=$M; print('left out:', ' '.join(n for b,n in ((1,'front'),(2,'side'),(4,'AR'),(8,'vocoder')) if m&b) or 'nothing', '-> ms/step', d['ms_per_step'])"
done
