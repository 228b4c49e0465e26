#!/bin/bash
# Round-2 measurement session (one GPU box): everything under profiles/r02_* comes from this script.  Outputs -> gpurun_out/r02/.
mkdir -p gpurun_out/r02
O=gpurun_out/r02
export TMPDIR=/tmp
# 1. HBM traffic / MFMA-pipe counters of the conv-GEMM kernels, steady steps only (separate --pmc passes, kernel-trace only)
bash tools/pmc.sh r02_b1 --no-batched --steps 120 > $O/pmc_b1.log 2>&1
bash tools/pmc.sh r02_b64 --no-batched --streams 64 --steps 30 > $O/pmc_b64.log 2>&1
cp gpurun_out/pmc_r02_b1.json $O/r02_pmc_b1.json; cp gpurun_out/pmc_r02_b64.json $O/r02_pmc_b64.json
cp $O/r02_pmc_b1.json profiles/r02_pmc_b1.json; cp $O/r02_pmc_b64.json profiles/r02_pmc_b64.json     # bench.py reads the newest committed pass
# 2. steady-state kernel tables (rocprofv3 --kernel-trace reduced to whole steps)
bash tools/prof_steady.sh r02_b1 1 100 > $O/steady_b1.log 2>&1
bash tools/prof_steady.sh r02_b64 64 30 > $O/steady_b64.log 2>&1
cp gpurun_out/r02_b1_steady_kernel_stats.csv gpurun_out/r02_b1_steady_summary.json gpurun_out/r02_b64_steady_kernel_stats.csv gpurun_out/r02_b64_steady_summary.json $O/
# 3. the bench line (default flags, as the driver runs it) + per-shape GEMM tables + the torch-ROCm second baseline
SVA_GEMM_TABLE=$O/r02_gemm_table_b1.csv python bench.py --torch-gpu-baseline > $O/r02_bench_b1.json 2> $O/bench_b1.err
mv $O/r02_gemm_table_b1.csv.b64 $O/r02_gemm_table_b64.csv
python bench.py --ar-dtype 1 --no-cpu-baseline --no-batched > $O/r02_bench_b1_fp16ar.json 2> $O/bench_b1_fp16.err
# 4. persistent AR kernel phase timeline, fp32 and fp16
python tools/ar_timing.py > $O/r02_ar_timing_fp32.log 2>&1
AR_DTYPE=1 python tools/ar_timing.py > $O/r02_ar_timing_fp16.log 2>&1
# 5. re-prefill burst latency inside a stream
python tools/reprefill_probe.py > $O/r02_reprefill_probe.log 2>&1
# 6. pipelined stage spans (when each chain could start / ended)
SVA_PIPE_TRACE=230 python bench.py --no-cpu-baseline --no-batched --no-roofline --steps 200 2>&1 >/dev/null | grep "pipe trace" | sed -n 1p\;100,130p > $O/r02_pipe_trace_b1.txt
ls -la $O
