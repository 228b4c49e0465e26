#!/bin/bash
# Counters of ONE planes-GEMM shape (sva_test_gemm_planes), separate rocprofv3 --pmc passes (kernel-trace only).
#   tools/planes_pmc.sh TAG M N K mode variant a_planes
TAG=$1; M=$2; N=$3; K=$4; MODE=$5; VAR=$6; AP=${7:-0}
export TMPDIR=/tmp
OUT=gpurun_out/planes_pmc_$TAG
mkdir -p $OUT
pass() {
  local NAME=$1; shift
  rm -rf $OUT/$NAME
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$NAME -o p -- \
      python -c "
import numpy as np
from streamvoiceanon_amd import engine as E
rng = np.random.default_rng(1)
A = rng.standard_normal(($M, $K)).astype(np.float32); W = (rng.standard_normal(($N, $K)) * 0.05).astype(np.float32)
print(E.test_gemm_planes(A, W, mode=$MODE, variant=$VAR, a_planes=bool($AP), iters=10)[1])" > $OUT/$NAME.log 2>&1
  echo "pass $NAME rc=$? $(tail -1 $OUT/$NAME.log)"
}
pass P1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass P2 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass P3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass P4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
pass P5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass P6 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/P*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "planes_gemm" not in r["Kernel_Name"] and "planes_dma" not in r["Kernel_Name"]: continue
        a = per[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in per.items():
    print(k[:120])
    for c, v in sorted(cs.items()): print(f"   {c:34s} {v[1] / max(v[0], 1):16.1f}")
PY
rm -rf $OUT/P?
