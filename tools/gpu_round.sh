#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
TAG=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
SVA_GEMM_TABLE=gpurun_out/gemm_table_b1.csv python bench.py > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err; tail -1 gpurun_out/bench_b1.json
SVA_GEMM_TABLE=gpurun_out/gemm_table_b64.csv python bench.py --steps 10 --warmup 3 --streams 64 --no-cpu-baseline --no-batched > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err; tail -1 gpurun_out/bench_b64.json
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o b1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batched > gpurun_out/prof_${TAG}_b1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o b64 -- python bench.py --steps 6 --warmup 2 --streams 64 --no-cpu-baseline --no-batched > gpurun_out/prof_${TAG}_b64.log 2>&1
ls gpurun_out/prof_${TAG}
rm -f gpurun_out/prof_${TAG}/*kernel_trace.csv   # per-dispatch rows are large; the stats summary is what gets committed
