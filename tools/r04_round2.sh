#!/bin/bash
# Round-4 GPU session after the planes GEMM: full parity tests, default bench (with its PMC passes), streams curves (default modes, fp16 AR +
# fp16 vocoder), steady-step kernel tables at 1 / 64 streams.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.log
tail -4 gpurun_out/r04_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; tail -1 gpurun_out/r04_smoke.log
( time SVA_GEMM_TABLE=gpurun_out/r04_gemm_table_b1.csv python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_b1_k20.json 2> gpurun_out/r04_bench_b1_k20.err ) 2> gpurun_out/r04_bench_default_wall.txt
tail -1 gpurun_out/r04_bench_b1_k20.json | cut -c1-300; tail -1 gpurun_out/r04_bench_b1_k20.json | tail -c 400
SVA_GEMM_TABLE=gpurun_out/r04_gemm_table_b64.csv python bench.py --steps 10 --warmup 3 --streams 64 --no-cpu-baseline --no-batched --no-pmc --no-torch-gpu-baseline --no-offline > gpurun_out/r04_bench_b64.json 2> gpurun_out/r04_bench_b64.err
bash tools/streams_curve.sh > gpurun_out/r04_streams_curve.txt 2>&1
AR_DTYPE=1 VOC_DTYPE=1 bash tools/streams_curve.sh > gpurun_out/r04_streams_curve_fp16.txt 2>&1
cut -c1-220 gpurun_out/r04_streams_curve.txt
for B in 1 64; do bash tools/prof_steady.sh r04_b$B $B $((B>=32?20:100)) > gpurun_out/r04_prof_b$B.txt 2>&1; done
