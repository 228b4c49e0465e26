#!/bin/bash
# Round-3 measurement session (one GPU box): everything under profiles/r03_* that is a measurement comes from this script.
# Outputs -> gpurun_out/r03/ (copy what is to be judged into profiles/).
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export TMPDIR=/tmp
X="--no-cpu-baseline --no-batched --no-pmc --no-torch-gpu-baseline --no-offline"
# 1. HBM traffic / MFMA-pipe counters of the conv-GEMM kernels, steady steps only (separate --pmc passes, kernel-trace only)
bash tools/pmc.sh r03_b1 $X --steps 120 > $O/pmc_b1.log 2>&1
bash tools/pmc.sh r03_b64 $X --streams 64 --steps 30 > $O/pmc_b64.log 2>&1
cp gpurun_out/pmc_r03_b1.json $O/r03_pmc_b1.json; cp gpurun_out/pmc_r03_b64.json $O/r03_pmc_b64.json
# 2. steady-state kernel tables (rocprofv3 --kernel-trace reduced to whole steps)
bash tools/prof_steady.sh r03_b1 1 100 > $O/steady_b1.log 2>&1
bash tools/prof_steady.sh r03_b64 64 30 > $O/steady_b64.log 2>&1
cp gpurun_out/r03_b1_steady_kernel_stats.csv gpurun_out/r03_b1_steady_summary.json gpurun_out/r03_b64_steady_kernel_stats.csv gpurun_out/r03_b64_steady_summary.json $O/
# 3. the bench line exactly as the driver runs it (default flags; K = 20 as the driver's BENCH_rNN) + per-shape GEMM tables
SVA_GEMM_TABLE=$O/r03_gemm_table_b1.csv python bench.py --steps 20 --warmup 5 > $O/r03_bench_b1_k20.json 2> $O/bench_b1_k20.err
mv $O/r03_gemm_table_b1.csv.b64 $O/r03_gemm_table_b64.csv
python bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-offline --no-pmc > $O/r03_bench_b1_k200.json 2> $O/bench_b1_k200.err
python bench.py --ar-dtype 1 $X > $O/r03_bench_b1_fp16ar.json 2> $O/bench_b1_fp16.err
# 4. persistent AR kernel phase timeline, fp32 and fp16
python tools/ar_timing.py > $O/r03_ar_timing_fp32.log 2>&1
AR_DTYPE=1 python tools/ar_timing.py > $O/r03_ar_timing_fp16.log 2>&1
# 5. re-prefill burst latency inside a stream
python tools/reprefill_probe.py > $O/r03_reprefill_probe.log 2>&1
# 6. pipelined stage spans (when each chain could start / ended)
SVA_DEBUG=pipe_trace=230 python bench.py $X --no-roofline --steps 200 2>&1 >/dev/null | grep "pipe trace" | sed -n 1p\;100,130p > $O/r03_pipe_trace_b1.txt
# 7. streams-per-GPU curve
bash tools/streams_curve.sh > $O/r03_streams_curve.txt 2>&1
ls -la $O
