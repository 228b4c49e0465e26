#!/bin/bash
# ONE closing GPU session (replaces the per-round tools/r0N_round*.sh):   bash tools/closing_session.sh TAG [retune]
#   retune: first re-derive every measured policy from this library -- GEMM dispatch table (tools/make_tune_table.py collect) and the pipelined-mode
#           decode / CU-partition table (tools/make_policy.py collect); their dumps land in gpurun_out/ for `merge` + rebuild on the build host
# then, always: full parity tests, smoke, the driver's default bench (with its PMC passes and GEMM table), 64-stream bench + GEMM table, configs[3] / [4]
# per-GPU shapes, K = 200, streams curves (default modes; fp16 AR + fp16 vocoder), steady-step kernel tables at 1 / 64 streams (rocprofv3 --kernel-trace),
# PMC aggregates at 1 / 64 streams (tools/pmc.sh: fabric bytes, MFMA busy, per stage), two-build wait-count audit, prompt latency.
# Every file is written as gpurun_out/${TAG}_*; copy the ones to keep into profiles/.
TAG=${1:?tag}; MODE=$2
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$MODE" = retune ]; then
  python tools/make_tune_table.py collect > gpurun_out/${TAG}_tune_collect.log 2>&1; tail -1 gpurun_out/${TAG}_tune_collect.log
  python tools/make_policy.py collect > gpurun_out/${TAG}_policy_collect.log 2>&1; tail -1 gpurun_out/${TAG}_policy_collect.log
fi
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
( time SVA_GEMM_TABLE=gpurun_out/${TAG}_gemm_table_b1.csv python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_b1_k20.json 2> gpurun_out/${TAG}_bench_b1_k20.err ) 2> gpurun_out/${TAG}_bench_default_wall.txt
[ -f gpurun_out/${TAG}_gemm_table_b1.csv.b64 ] && mv gpurun_out/${TAG}_gemm_table_b1.csv.b64 gpurun_out/${TAG}_gemm_table_b64_of_default_run.csv
tail -1 gpurun_out/${TAG}_bench_b1_k20.json | cut -c1-400; tail -1 gpurun_out/${TAG}_bench_b1_k20.json | tail -c 600
X="--no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-offline --no-batched"
SVA_GEMM_TABLE=gpurun_out/${TAG}_gemm_table_b64.csv python bench.py --steps 20 --warmup 5 --streams 64 $X > gpurun_out/${TAG}_bench_b64.json 2> gpurun_out/${TAG}_bench_b64.err
python bench.py --config 4 $X --no-roofline > gpurun_out/${TAG}_bench_config4.json 2>/dev/null
python bench.py --config 5 $X --no-roofline > gpurun_out/${TAG}_bench_config5.json 2>/dev/null
# the multi-rank code path (RCCL group, per-rank clocks, gather + its assertion) on the one GPU there is: a 1-rank process group
SVA_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --config 4 --steps 20 $X --no-roofline > gpurun_out/${TAG}_bench_config4_rccl_world1.json 2> gpurun_out/${TAG}_bench_config4_rccl_world1.err
python bench.py --steps 200 --warmup 5 $X --no-roofline > gpurun_out/${TAG}_bench_b1_k200.json 2>/dev/null
bash tools/streams_curve.sh > gpurun_out/${TAG}_streams_curve.txt 2>&1
AR_DTYPE=1 VOC_DTYPE=1 bash tools/streams_curve.sh > gpurun_out/${TAG}_streams_curve_fp16.txt 2>&1
cut -c1-220 gpurun_out/${TAG}_streams_curve.txt
for B in 1 64; do bash tools/prof_steady.sh ${TAG}_b$B $B $((B>=32?20:100)) > gpurun_out/${TAG}_prof_b$B.txt 2>&1; done
bash tools/pmc.sh ${TAG}_b1 --steps 110 --warmup 3 --no-batched --no-pmc --no-torch-gpu-baseline --no-offline > gpurun_out/${TAG}_pmc_b1.log 2>&1; tail -1 gpurun_out/${TAG}_pmc_b1.log | cut -c1-300
bash tools/pmc.sh ${TAG}_b64 --streams 64 --steps 110 --warmup 3 --no-batched --no-pmc --no-torch-gpu-baseline --no-offline > gpurun_out/${TAG}_pmc_b64.log 2>&1; tail -1 gpurun_out/${TAG}_pmc_b64.log | cut -c1-300
bash tools/waitcnt_audit.sh > gpurun_out/${TAG}_waitcnt_audit.txt 2>&1; head -3 gpurun_out/${TAG}_waitcnt_audit.txt
python tools/prompt_latency_probe.py > gpurun_out/${TAG}_prompt_latency_probe.txt 2>&1
