#!/bin/bash
# Timeline of ONE steady-state B=1 step from a rocprofv3 kernel trace: per-kernel duration and the gap to the previous
# kernel on the same queue, aggregated per kernel name.  tools/trace_step.sh [bench args]
export TMPDIR=/tmp
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -o t -- python bench.py --no-cpu-baseline --no-pmc --no-torch-gpu-baseline --no-offline --no-batched --no-roofline --steps 12 --warmup 3 "$@" > gpurun_out/trace_step.log 2>&1
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open("gpurun_out/trace_step/t_kernel_trace.csv")))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: take the last 4 steps' worth by locating stft_mag_kernel dispatches with the small (streaming) grid
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "shift_history_kernel" in n]
print("n dispatches", len(rows))
# the step starts with ring_write; find them
rw = [i for i, n in enumerate(names) if "ring_write" in n]
print("ring_write dispatches:", len(rw))
lo, hi = rw[-3], rw[-2]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print("step kernels %d wall %.1f us" % (len(step), (t1 - t0) / 1e3))
byq = collections.defaultdict(list)
for r in step: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("queue", q, "kernels", len(rs), "busy %.1f us" % (busy / 1e3))
agg = collections.OrderedDict()
prev_end = {}
out = open("gpurun_out/trace_step/timeline.txt", "w")
for r in step:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - prev_end[q]) if q in prev_end else 0
    prev_end[q] = e
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sva::", "").replace("sva::", "")
    a = agg.setdefault(n, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += gap
    out.write("%9.2f q%s %-50s dur %7.2f gap %7.2f grid %s wg %s\n" % ((s - t0) / 1e3, q, n[:50], (e - s) / 1e3, gap / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
print("%-52s %5s %9s %9s %9s" % ("kernel", "n", "dur us", "avg", "gap avg"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %5d %9.1f %9.2f %9.2f" % (n[:52], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / a[0] / 1e3))
PY
rm -f gpurun_out/trace_step/t_kernel_trace.csv
