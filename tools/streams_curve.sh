# aggregate frames/s against the number of concurrent streams on one GPU
for B in ${BS:-1 2 3 4 6 8 12 16 32 64 128}; do
  timeout 300 python bench.py ${AR_DTYPE:+--ar-dtype $AR_DTYPE} ${MM_MODE:+--mm-mode $MM_MODE} ${VOC_DTYPE:+--voc-dtype $VOC_DTYPE} --streams $B --steps $((B>=32?20:60)) --warmup 5 --no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $B, 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'rtf/stream', d['rtf'], 'sync p50', d['sync_latency_ms']['p50'], 'stages', d['stage_ms_last_step'])"
done
