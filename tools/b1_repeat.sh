#!/bin/bash
# headline stability: the single-stream pipelined bench several times at K = 20 (the driver's) and K = 200
mkdir -p gpurun_out/rep
for k in 20 20 20 200 200; do
  python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-batched --no-pmc --no-torch-gpu-baseline --no-offline --no-roofline > gpurun_out/rep/b.json 2> gpurun_out/rep/b.err
  python -c "
import json;d=json.loads(open('gpurun_out/rep/b.json').read().strip().splitlines()[-1]);print('K=$k', d['value'], d['ms_per_step'], d['sync_latency_ms']['p50'], d['stage_ms_last_step'])"
done
