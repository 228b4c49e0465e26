"""Aggregate the rocprofv3 --pmc passes of tools/pmc.sh per kernel -> gpurun_out/pmc_<TAG>.json"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name in ("RD", "WR"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{name}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            a = per[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
out = {}
for k, cs in per.items():
    d = {"calls": max(v[0] for v in cs.values())}
    rd = cs.get("TCC_EA0_RDREQ_sum", [1, 0.0])
    rd32 = cs.get("TCC_EA0_RDREQ_32B_sum", [1, 0.0])
    wr = cs.get("WRITE_SIZE", [1, 0.0])
    rdreq, rdreq32 = rd[1] / max(rd[0], 1), rd32[1] / max(rd32[0], 1)
    d["rdreq_avg"] = rdreq
    d["rdreq32_avg"] = rdreq32
    # FETCH_SIZE definition (64 B / request, 32 B for _32B) with the gfx950 x2 on the wide part
    d["read_bytes_avg"] = rdreq32 * 32 + (rdreq - rdreq32) * 64 * 2
    d["write_bytes_avg"] = wr[1] / max(wr[0], 1) * 1024
    out[k] = d
gemm = {k: v for k, v in out.items() if "gemm_kernel" in k}
n = sum(v["calls"] for v in gemm.values()) or 1
rd = sum(v["read_bytes_avg"] * v["calls"] for v in gemm.values()) / n
wr = sum(v["write_bytes_avg"] * v["calls"] for v in gemm.values()) / n
summary = {
    "tag": tag,
    "note": "per-launch averages over every conv-GEMM dispatch (tiled + skinny kernels) of the profiled bench run incl. prompt prefill; "
            "reads = 32 B x RDREQ_32B + 2 x 64 B x (RDREQ - RDREQ_32B) (FETCH_SIZE definition + gfx950 x2 correction of "
            "MI355X_MICROARCH.md); writes = WRITE_SIZE KiB x 1024 (uncalibrated)",
    "gemm_launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "kernels": dict(sorted(out.items(), key=lambda kv: -(kv[1]["read_bytes_avg"] + kv[1]["write_bytes_avg"]) * kv[1]["calls"])[:20]),
}
json.dump(summary, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("gemm_launches", "read_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}))
