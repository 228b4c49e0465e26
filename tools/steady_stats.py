"""Per-STEP kernel statistics of the timed (steady-state) region of a rocprofv3 --kernel-trace of bench.py.

    python tools/steady_stats.py <kernel_trace.csv> <skip_steps> <steps> <out.csv>

bench.py consumes one `ring_write_kernel` dispatch per chunk-step, so the timed region is delimited by dispatch number: it starts
at the (skip + 1)-th ring_write and ends at the (skip + steps + 1)-th (prompt prefill, stream begin, delay / warm-up steps and
the latency / roofline samples that follow fall outside).  Output: per kernel name, calls per step, average duration, summed
duration per step -- i.e. one steady step -- plus a JSON summary on stdout."""
import collections
import csv
import json
import sys

path, skip, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = list(csv.DictReader(open(path)))
key_s = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
key_e = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
rows.sort(key=lambda r: int(r[key_s]))
ring = [int(r[key_s]) for r in rows if "ring_write_kernel" in r["Kernel_Name"]]
assert len(ring) > skip + steps, (len(ring), skip, steps)
t0, t1 = ring[skip], ring[skip + steps]
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r[key_s])
    if t0 <= s < t1:
        a = per[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r[key_e]) - s) / 1e3
tot = sum(v[1] for v in per.values())
with open(out, "w") as f:
    f.write("Name,CallsPerStep,AverageUs,TotalUsPerStep,Percentage\n")
    for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%.3f,%.3f,%.3f,%.2f\n' % (k, n / steps, us / n, us / steps, 100.0 * us / tot))
gemm = {k: v for k, v in per.items() if any(t in k for t in ("gemm_kernel", "split_ws_kernel", "planes_dma_kernel", "voc_conv_kernel"))}
ar = {k: v for k, v in per.items() if "ar_decode_kernel" in k}
print(json.dumps({
    "steps": steps, "wall_us_per_step_under_tracer": round((t1 - t0) / 1e3 / steps, 2),
    "kernel_us_per_step": round(tot / steps, 2), "launches_per_step": round(sum(v[0] for v in per.values()) / steps, 2),
    "gemm_launches_per_step": round(sum(v[0] for v in gemm.values()) / steps, 2),
    "gemm_us_per_step": round(sum(v[1] for v in gemm.values()) / steps, 2),
    "gemm_avg_launch_us": round(sum(v[1] for v in gemm.values()) / max(1, sum(v[0] for v in gemm.values())), 3),
    "ar_decode_kernel_us_per_step": round(sum(v[1] for v in ar.values()) / steps, 2)}))
