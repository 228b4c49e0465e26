for f in 1 0; do
  SVA_XCD_SWIZZLE=$f SVA_GEMM_TABLE=gpurun_out/gt_xcd$f.csv timeout 300 python bench.py --streams 64 --steps 12 --warmup 3 --no-cpu-baseline --no-batched 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd_swizzle=$f', d['value'], d['ms_per_step'], d['stage_ms_last_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
done
python - <<'PY'
import csv
a={tuple(r[k] for k in ('M','N','K','taps','mode')):r for r in csv.DictReader(open('gpurun_out/gt_xcd1.csv'))}
b={tuple(r[k] for k in ('M','N','K','taps','mode')):r for r in csv.DictReader(open('gpurun_out/gt_xcd0.csv'))}
rows=sorted(a, key=lambda k:-float(b[k]['total_us']) if k in b else 0)[:14]
for k in rows:
    if k in b: print(k, 'swz %s us %s TF | plain %s us %s TF'%(a[k]['avg_us'],a[k]['TFLOPs'],b[k]['avg_us'],b[k]['TFLOPs']))
PY
timeout 300 python -m pytest tests -m gpu -x -q -k "gemm or encoder_codes or batched_streams" 2>&1 | tail -2
