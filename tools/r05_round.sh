#!/bin/bash
# Round-5 closing GPU session: full parity tests, smoke, default bench (with its PMC passes and GEMM table), 64-stream bench + GEMM table, configs[3] / [4]
# per-GPU shapes, streams curves (default modes; fp16 AR + fp16 vocoder), steady-step kernel tables at 1 / 64 streams, two-build wait-count audit.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.log
tail -4 gpurun_out/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -1 gpurun_out/r05_smoke.log
( time SVA_GEMM_TABLE=gpurun_out/r05_gemm_table_b1.csv python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_b1_k20.json 2> gpurun_out/r05_bench_b1_k20.err ) 2> gpurun_out/r05_bench_default_wall.txt
tail -1 gpurun_out/r05_bench_b1_k20.json | cut -c1-400; tail -1 gpurun_out/r05_bench_b1_k20.json | tail -c 600
SVA_GEMM_TABLE=gpurun_out/r05_gemm_table_b64.csv python bench.py --steps 20 --warmup 5 --streams 64 --no-cpu-baseline --no-batched --no-pmc --no-torch-gpu-baseline --no-offline > gpurun_out/r05_bench_b64.json 2> gpurun_out/r05_bench_b64.err
python bench.py --config 4 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-offline --no-batched --no-roofline > gpurun_out/r05_bench_config4.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-offline --no-batched --no-roofline > gpurun_out/r05_bench_config5.json 2>/dev/null
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-offline --no-batched --no-roofline > gpurun_out/r05_bench_b1_k200.json 2>/dev/null
bash tools/streams_curve.sh > gpurun_out/r05_streams_curve.txt 2>&1
AR_DTYPE=1 VOC_DTYPE=1 bash tools/streams_curve.sh > gpurun_out/r05_streams_curve_fp16.txt 2>&1
cut -c1-220 gpurun_out/r05_streams_curve.txt
for B in 1 64; do bash tools/prof_steady.sh r05_b$B $B $((B>=32?20:100)) > gpurun_out/r05_prof_b$B.txt 2>&1; done
bash tools/waitcnt_audit.sh > gpurun_out/r05_waitcnt_audit.txt 2>&1; head -3 gpurun_out/r05_waitcnt_audit.txt
python tools/prompt_latency_probe.py > gpurun_out/r05_prompt_latency_probe.txt 2>&1
