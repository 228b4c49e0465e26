#!/bin/bash
# Round-3 closing measurement session (one GPU box): the driver-shaped bench line, steady-state kernel tables, streams curves for
# both AR dtypes, the GPU test log.  Outputs -> gpurun_out/r03f/ (copy what is to be judged into profiles/).
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
export TMPDIR=/tmp
X="--no-cpu-baseline --no-batched --no-pmc --no-torch-gpu-baseline --no-offline"
timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$" > $O/r03_pytest_gpu.log
# the bench line exactly as the driver runs it (default flags; K = 20 as the driver's BENCH_rNN), then K = 200
T0=$(date +%s); SVA_GEMM_TABLE=$O/r03_gemm_table_b1.csv timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r03_bench_b1_k20.json 2> $O/bench_b1_k20.err
echo "default bench.py wall seconds: $(( $(date +%s) - T0 ))" > $O/bench_default_wall.txt; mv $O/r03_gemm_table_b1.csv.b64 $O/r03_gemm_table_b64.csv 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-offline --no-pmc > $O/r03_bench_b1_k200.json 2> $O/bench_b1_k200.err
# steady-state kernel tables (rocprofv3 --kernel-trace reduced to whole steps)
bash tools/prof_steady.sh r03_b1 1 100 > $O/steady_b1.log 2>&1
bash tools/prof_steady.sh r03_b64 64 30 > $O/steady_b64.log 2>&1
cp gpurun_out/r03_b1_steady_kernel_stats.csv gpurun_out/r03_b1_steady_summary.json gpurun_out/r03_b64_steady_kernel_stats.csv gpurun_out/r03_b64_steady_summary.json $O/
# streams-per-GPU curves, fp32 AR and fp16 AR (batched fp16 decode on the f16 pipes above 6 streams)
bash tools/streams_curve.sh > $O/r03_streams_curve.txt 2>&1
AR_DTYPE=1 bash tools/streams_curve.sh > $O/r03_streams_curve_fp16.txt 2>&1
# re-prefill burst latency inside a stream (MFMA prefill attention)
python tools/reprefill_probe.py > $O/r03_reprefill_probe.log 2>&1
ls -la $O
