#!/bin/bash
# Round 5, VERDICT r04 item 6: the 24-48-stream policy after the planes-DMA kernel -- pipelined frames/s and synchronous p50 per CU partition of the AR chain
# (cu_partition=0: none; cu_ar=N: AR chain on N CUs, encoder + vocoder on the rest) and per decode kernel (ar_batch=0: multi-launch, 2: the one-launch batched kernel)
for B in ${BS:-24 32 48}; do
  for CFG in "" "cu_partition=0" "cu_partition=1,cu_ar=32" "cu_partition=1,cu_ar=48" "cu_partition=1,cu_ar=64" "cu_partition=1,cu_ar=96" "ar_batch=2,cu_partition=0" "ar_batch=0,cu_partition=0"; do
    SVA_DEBUG=$CFG timeout 300 python bench.py --streams $B --steps 30 --warmup 5 --no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $B, 'SVA_DEBUG=\"$CFG\"', 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'sync p50', d['sync_latency_ms']['p50'], 'stages', d['stage_ms_last_step'])"
  done
done
