#!/bin/bash
# PMC look at the split-bf16 GEMM kernel on one large shape:  tools/split_pmc.sh VARIANT
V=${1:-4}
export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import sys; sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E
print(E.bench_gemm(64, 128, 512, 1536, 1, 1, 2, iters=20))
PY
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  rm -rf gpurun_out/spmc
  SVA_TUNE_TABLE=0 SVA_SPLIT_VARIANT=$V timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/spmc -o p -- python /tmp/one_gemm.py > gpurun_out/spmc.log 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/spmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "split_" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items(): print(f"{k}: {v / max(n,1):.4g} per launch ({n} launches)")
PY
done
rm -rf gpurun_out/spmc
