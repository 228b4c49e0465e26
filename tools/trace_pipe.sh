#!/bin/bash
# Per-queue busy / gap accounting of the pipelined single-stream run (rocprofv3 kernel trace): is a chain stretched by longer
# kernels (sharing CUs) or by gaps between its kernels (dispatch / dependencies)?
export TMPDIR=/tmp
rm -rf gpurun_out/trace_pipe
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_pipe -o t -- python bench.py --no-cpu-baseline --no-pmc --no-torch-gpu-baseline --no-offline --no-batched --no-roofline --steps 40 --warmup 6 "$@" > gpurun_out/trace_pipe.log 2>&1
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open("gpurun_out/trace_pipe/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
rw = [i for i, n in enumerate(names) if "ring_write" in n]
# the timed region = steps 8..48 roughly: take ring_write #12 .. #40 (pipelined steady state)
lo, hi = rw[12], rw[40]
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
nsteps = 28
print("window %.1f us for %d steps -> %.1f us/step" % ((t1 - t0) / 1e3, nsteps, (t1 - t0) / 1e3 / nsteps))
byq = collections.defaultdict(list)
for r in sel: byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    kinds = collections.Counter(re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void sva::", "").replace("sva::", "") for r in rs).most_common(3)
    print("queue %s: %5d kernels (%.0f/step)  busy %.0f us/step  avg dur %.2f us  idle-gap %.0f us/step (median gap %.2f us)  top: %s" % (
        q, len(rs), len(rs) / nsteps, busy / 1e3 / nsteps, busy / 1e3 / len(rs), sum(pos) / 1e3 / nsteps, sorted(gaps)[len(gaps) // 2] / 1e3, kinds))
PY
rm -f gpurun_out/trace_pipe/t_kernel_trace.csv
