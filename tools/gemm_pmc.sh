#!/bin/bash
# Issue / stall / LDS counters of ONE conv-GEMM shape (sva_bench_gemm), separate rocprofv3 --pmc passes (kernel-trace only).
#   tools/gemm_pmc.sh TAG B T N Cin taps dil mode
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/gemm_pmc_$TAG
mkdir -p $OUT
ARGS="$*"
pass() {
  local NAME=$1; shift
  rm -rf $OUT/$NAME
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$NAME -o p -- \
      python -c "from streamvoiceanon_amd import engine as E; print(E.bench_gemm(*[int(x) for x in '$ARGS'.split()], iters=20))" > $OUT/$NAME.log 2>&1
  echo "pass $NAME rc=$?"
}
pass P1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass P2 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass P3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass P4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass P5 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
pass P6 SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16
python - <<PY
import csv, glob, collections, json
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/P*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"] and "split_ws" not in r["Kernel_Name"]: continue
        a = per[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / max(v[0], 1) for c, v in cs.items()} for k, cs in per.items()}
for k, cs in out.items():
    print(k[:100])
    for c, v in sorted(cs.items()): print(f"   {c:34s} {v:16.1f}")
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
rm -rf $OUT/P?
