export TMPDIR=/tmp
rm -rf gpurun_out/pp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp -o p -- python tools/reprefill_probe.py > gpurun_out/pp.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/pp/p_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("ar_attention","rope_kvwrite","build_prompt","build_delayfill")) or float(r["Percentage"])>2.5:
        print("%-70s n=%6s avg=%9.2fus tot=%8.2fms %5.1f%%"%(n[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
rm -rf gpurun_out/pp
