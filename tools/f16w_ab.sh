#!/bin/bash
# A/B of the fp16-weight GEMM path (gemm_f16w.hip) on the batched fp16 AR decode: unit timings per layer shape, then bench lines.
set -u
mkdir -p gpurun_out/f16w
OUT=gpurun_out/f16w
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16w or fp16" -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 600 python - > $OUT/shapes.txt 2>&1 <<'PY'
import numpy as np
from streamvoiceanon_amd import engine as E
rng = np.random.default_rng(0)
for M in (16, 32, 64, 128, 256):
    for (N, K, kw, name) in ((2304, 768, dict(rms=True), "wqkv"), (768, 768, dict(res=True), "wo"), (4608, 768, dict(rms=True, swiglu=True), "w13"),
                             (768, 2304, dict(res=True), "w2"), (8200, 768, dict(rms=True), "head")):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        args = {}
        if kw.get("rms"): args["rms_w"] = np.ones(K, np.float32)
        if kw.get("res"): args["res"] = np.zeros((M, N), np.float32)
        if kw.get("swiglu"): args["swiglu"] = True
        _, us = E.test_gemm_f16w(A, W, iters=200, **args)
        gbs = N * K * 2 / us / 1e3
        print(f"M={M:4d} {name:5s} N={N:5d} K={K:5d}  {us:7.2f} us  weight stream {gbs:7.1f} GB/s  {2.0*M*N*K/us/1e6:7.2f} TFLOP/s")
PY
for B in 8 16 64 128; do
  for F in 0 1; do
    SVA_DEBUG=f16_weights=$F timeout 900 python bench.py --streams $B --ar-dtype 1 --steps 100 --warmup 20 --no-torch-gpu-baseline --no-pmc --no-offline --no-cpu-baseline 2>$OUT/b${B}_f$F.err | tail -1 > $OUT/b${B}_f$F.json
    python - <<PY >> $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/b${B}_f$F.json").read())
    print("B=$B f16_weights=$F value", j["value"], "ms_per_step", j["ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("achieved", "frac", "frac_of_own_pipes")})
except Exception as ex:
    print("B=$B f16_weights=$F failed", ex)
PY
  done
done
cat $OUT/summary.txt
tail -5 $OUT/pytest.log
