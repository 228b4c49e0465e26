#!/bin/bash
# HBM traffic of the conv-GEMM kernels from PMC counters (separate passes, kernel-trace only -- see
# /opt/skills/guides/MI355X_MICROARCH.md "HBM").  rocprofv3's derived FETCH_SIZE crashes the tool on this image, so
# the read side is taken from its definition: FETCH_SIZE = 64 B x TCC_EA0_RDREQ (32 B for the _32B requests), and the
# guide's gfx950 correction (a wide coalesced stream is tallied at half its bytes) doubles the non-32B part.
# Single-stream engine (SVA_DEBUG=concurrency=0): counter collection serialises dispatches.
#   tools/pmc.sh TAG [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
run() {  # name, counters...
  local NAME=$1; shift
  rm -rf gpurun_out/pmc_${TAG}_${NAME}
  SVA_DEBUG=concurrency=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_${NAME} -o p -- \
      python bench.py --no-cpu-baseline --no-roofline --no-pipeline "${BENCH_ARGS[@]}" > gpurun_out/pmc_${TAG}_${NAME}.log 2>&1
  echo "pass ${NAME} rc=$?"
}
BENCH_ARGS=("$@")
run RD TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run WR WRITE_SIZE
# MFMA pipe occupancy (SURVEY.md 8d reporting): busy cycles of the matrix pipes (summed over SIMDs) against the GPU-active
# cycles of the same dispatch; separate passes so that a counter this image lacks only loses its own pass
run MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run MOPS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16
python tools/pmc_agg.py ${TAG}
rm -rf gpurun_out/pmc_${TAG}_RD gpurun_out/pmc_${TAG}_WR gpurun_out/pmc_${TAG}_MFMA gpurun_out/pmc_${TAG}_MOPS   # per-dispatch rows are large; the aggregate JSON is what gets committed
