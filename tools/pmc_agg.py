"""Aggregate the rocprofv3 --pmc passes of tools/pmc.sh per kernel -> gpurun_out/pmc_<TAG>.json"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
# per STAGE of the (non-pipelined, single-queue) step, by position in the dispatch sequence: ring_write .. append_content = encoder,
# .. fsq_decode = AR, .. next ring_write = vocoder (SURVEY.md 8d "Reporting": achieved HBM GB/s and MFMA utilisation per stage)
stage = collections.defaultdict(lambda: collections.defaultdict(float))


def stage_of(rows):
    """Dispatch_Id -> stage name, from the kernel sequence of one pass"""
    seq = sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in rows})
    cur, m = None, {}
    for did, kn in seq:
        if "ring_write" in kn:
            cur = "encoder"
        elif "append_content" in kn and cur == "encoder":
            m[did] = cur
            cur = "ar"
            continue
        elif "fsq_decode" in kn and cur == "ar":
            cur = "vocoder"
        if cur:
            m[did] = cur
    return m

# steady state only: the dispatches of steps [SKIP, SKIP + STEPS) -- every (non-pipelined) step starts with one ring_write_kernel;
# prompt prefill, delay fill, warm-up and the trailing latency-sample steps of bench.py stay outside the window
SKIP, STEPS = 10, 100
window_note = None
for name in ("RD", "WR", "MFMA", "MOPS"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{name}/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if "ring_write" in r["Kernel_Name"]})
        lo, hi = None, None
        if len(ids) >= SKIP + 2:
            lo, hi = ids[SKIP], ids[min(SKIP + STEPS, len(ids) - 1)]
            window_note = f"dispatches of {min(SKIP + STEPS, len(ids) - 1) - SKIP} steady steps (after {SKIP} start-up steps)"
        st_of = stage_of(rows)
        for row in rows:
            if lo is not None and not (lo <= int(row["Dispatch_Id"]) < hi):
                continue
            a = per[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            sg = st_of.get(int(row["Dispatch_Id"]))
            if sg:
                stage[sg][row["Counter_Name"]] += float(row["Counter_Value"])
        if name == "RD":        # kernel durations of the same pass (the counters serialise dispatches: durations are per kernel, not overlapped)
            for tf in glob.glob(f"gpurun_out/pmc_{tag}_{name}/*kernel_trace.csv"):
                for row in csv.DictReader(open(tf)):
                    did = int(row["Dispatch_Id"])
                    if lo is not None and not (lo <= did < hi):
                        continue
                    sg = st_of.get(did)
                    if sg:
                        stage[sg]["kernel_ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                        stage[sg]["launches"] += 1
            stage["_"]["steps"] = (min(SKIP + STEPS, len(ids) - 1) - SKIP) if lo is not None else 0
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
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"][1] > 0:
        # matrix-pipe occupancy: busy cycles summed over the 1024 SIMDs / (GPU-active cycles x 1024); rocprofv3 reports
        # GRBM_GUI_ACTIVE summed over the 8 XCDs (a 15 us dispatch shows ~8 x 36 k cycles), hence x 1024 / 8 = x 128
        d["mfma_busy_cycles_avg"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(cs["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)
        d["gui_active_cycles_avg"] = cs["GRBM_GUI_ACTIVE"][1] / max(cs["GRBM_GUI_ACTIVE"][0], 1)
        d["mfma_util"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (cs["GRBM_GUI_ACTIVE"][1] * 128.0)
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in cs:
        d["mfma_mops_bf16_avg"] = cs["SQ_INSTS_VALU_MFMA_MOPS_BF16"][1] / max(cs["SQ_INSTS_VALU_MFMA_MOPS_BF16"][0], 1)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in cs:
        d["mfma_mops_f32_avg"] = cs["SQ_INSTS_VALU_MFMA_MOPS_F32"][1] / max(cs["SQ_INSTS_VALU_MFMA_MOPS_F32"][0], 1)
    out[k] = d
gemm = {k: v for k, v in out.items() if any(t in k for t in ("gemm_kernel", "split_ws_kernel", "planes_dma_kernel", "voc_conv_kernel"))}
n = sum(v["calls"] for v in gemm.values()) or 1
rd = sum(v["read_bytes_avg"] * v["calls"] for v in gemm.values()) / n
wr = sum(v["write_bytes_avg"] * v["calls"] for v in gemm.values()) / n
busy = sum(per[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] for k in gemm if "SQ_VALU_MFMA_BUSY_CYCLES" in per[k])
act = sum(per[k]["GRBM_GUI_ACTIVE"][1] for k in gemm if "GRBM_GUI_ACTIVE" in per[k])
summary = {
    "tag": tag,
    "window": window_note or "whole run (no step marker found)",
    "note": "per-launch averages over every conv-GEMM dispatch (pipe / tiled / skinny kernels) inside the window; "
            "reads = 32 B x RDREQ_32B + 2 x 64 B x (RDREQ - RDREQ_32B) (FETCH_SIZE definition + gfx950 x2 correction of "
            "MI355X_MICROARCH.md); writes = WRITE_SIZE KiB x 1024 (uncalibrated)",
    "gemm_mfma_util": (busy / (act * 128.0)) if act > 0 else None,
    "gemm_mfma_util_note": "sum SQ_VALU_MFMA_BUSY_CYCLES (all 1024 SIMDs) / (sum GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024) over the conv-GEMM dispatches (gfx94x-style MfmaUtil; "
                           "rocprofv3 ships no gfx950 derived-metric section)",
    "gemm_launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "stages": {},
    "kernels": dict(sorted(out.items(), key=lambda kv: -(kv[1]["read_bytes_avg"] + kv[1]["write_bytes_avg"]) * kv[1]["calls"])[:20]),
}
nsteps = max(stage["_"].get("steps", 0), 1)
for sg in ("encoder", "ar", "vocoder"):
    c = stage.get(sg)
    if not c or not c.get("kernel_ns"):
        continue
    rdb = c.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32 + (c.get("TCC_EA0_RDREQ_sum", 0.0) - c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) * 64 * 2
    wrb = c.get("WRITE_SIZE", 0.0) * 1024
    summary["stages"][sg] = {
        "launches_per_step": c["launches"] / nsteps, "kernel_ms_per_step": c["kernel_ns"] / nsteps * 1e-6,
        "hbm_read_MB_per_step": rdb / nsteps / 1e6, "hbm_write_MB_per_step": wrb / nsteps / 1e6,
        "achieved_hbm_GBs": (rdb + wrb) / c["kernel_ns"], "frac_hbm_8TBs": (rdb + wrb) / c["kernel_ns"] / 8000.0,
        "mfma_util": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)) if c.get("GRBM_GUI_ACTIVE") else None}
summary["stages_note"] = ("per stage of a serial step (dispatch order: ring_write .. append_content = encoder, .. fsq_decode = AR, rest = vocoder): fabric-side bytes "
                          "(same formula as above) over the summed kernel durations of the stage, and matrix-pipe busy cycles over GPU-active cycles x 1024 SIMDs")
json.dump(summary, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("gemm_mfma_util", "gemm_launches", "read_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch", "stages")}))
