#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py:  tools/prof.sh TAG [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o p -- python bench.py --no-cpu-baseline --no-pmc --no-torch-gpu-baseline --no-offline "$@" > gpurun_out/prof_${TAG}.log 2>&1
tail -1 gpurun_out/prof_${TAG}.log | cut -c1-400
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_${TAG}/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f"%(tot/1e6))
for r in rows[:22]:
    print("%-80s n=%6s avg=%9.2fus tot=%8.2fms %5.1f%%"%(r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
rm -f gpurun_out/prof_${TAG}/*kernel_trace.csv
