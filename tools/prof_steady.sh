#!/bin/bash
# rocprofv3 kernel trace of bench.py reduced to ONE STEADY STEP (tools/steady_stats.py):  tools/prof_steady.sh TAG STREAMS STEPS
TAG=$1; B=$2; K=${3:-100}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${TAG} -o p -- python bench.py --streams $B --steps $K --warmup 5 --no-pin \
    --no-cpu-baseline --no-batched --no-roofline --no-pmc --no-torch-gpu-baseline --no-offline > gpurun_out/prof_${TAG}.log 2>&1
tail -1 gpurun_out/prof_${TAG}.log | cut -c1-300
# 2 delay chunks + 5 warm-up steps precede the timed region
python tools/steady_stats.py $(ls gpurun_out/prof_${TAG}/*kernel_trace.csv | head -1) 7 $K gpurun_out/${TAG}_steady_kernel_stats.csv | tee gpurun_out/${TAG}_steady_summary.json
head -12 gpurun_out/${TAG}_steady_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_${TAG}
