#!/bin/bash
# Round-4 GPU session: parity tests, smoke, default bench, streams curves (fp32 / fp16 AR), batched-decode phase table, steady-step
# kernel tables at 1 / 12 / 64 streams.  Outputs under gpurun_out/ (copied into profiles/ by hand).
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.log
tail -4 gpurun_out/r04_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; tail -1 gpurun_out/r04_smoke.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_b1_k20.json 2> gpurun_out/r04_bench_b1_k20.err ) 2> gpurun_out/r04_bench_default_wall.txt
tail -1 gpurun_out/r04_bench_b1_k20.json | cut -c1-600
bash tools/streams_curve.sh > gpurun_out/r04_streams_curve.txt 2>&1
AR_DTYPE=1 bash tools/streams_curve.sh > gpurun_out/r04_streams_curve_fp16.txt 2>&1
cut -c1-220 gpurun_out/r04_streams_curve.txt
STEPS=10 timeout 600 python tools/ar_batch_check.py 8 12 32 64 > gpurun_out/r04_abatch_check_fp32.txt 2>&1
tail -40 gpurun_out/r04_abatch_check_fp32.txt
for B in 1 12 64; do bash tools/prof_steady.sh r04_b$B $B $((B>=32?20:100)) > gpurun_out/r04_prof_b$B.txt 2>&1; done
