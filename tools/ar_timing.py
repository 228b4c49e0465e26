"""Phase timeline of the persistent AR decode kernel (SVA_DEBUG=ar_timing=1): workgroup 0 stamps wall_clock64() (100 MHz) when a
phase's input has been gathered ("in") and when its outputs are computed ("out").  Prints the mean over a few frames."""
import os
import sys

os.environ["SVA_DEBUG"] = "ar_timing=1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

ar_dtype = int(os.environ.get("AR_DTYPE", "0"))
W = {k: sw.generate(0, k, shp) for k, shp in specs.all_specs().items()}
W = {k: v for k, v in W.items() if v is not None}
eng = E.Engine(W, ar_dtype=ar_dtype)
b = E.Batch(eng, n_streams=1)
ac, cc, style, timbre = synth_prompt(2000, 107)
b.prefill_prompt(0, cc, ac, style, timbre, noise_seed=1000)
b.begin()
src = synth_utterance(1000, 2048 * 40)
labels = []
v2 = False      # label set of the 192-workgroup experiment (commit c4feb26, profiles/r03_ar2_*): sampler fused into the next FA
for l in range(12):
    for ph in ("A", "B1", "B1m", "B2", "C", "D"):
        labels += [f"s{l}.{ph}.in", f"s{l}.{ph}.out"]
if not v2:
    labels.append("hidden.in")
for cb in range(8):
    for l in range(4):
        for ph in ("FA", "FB", "FC", "FD"):
            ph2 = "FSA" if (v2 and ph == "FA" and l == 0 and cb > 0) else ph
            labels += [f"f{cb}.{l}.{ph2}.in", f"f{cb}.{l}.{ph2}.out"]
    labels += [f"f{cb}.FH.in", f"f{cb}.FH.out"] + ([] if v2 else [f"f{cb}.FS.in"])
if v2:
    labels += ["f8.FS.in", "f8.FS.out"]
acc = None
n = 0
for i in range(30):
    b.step(src[i * 2048:(i + 1) * 2048][None])
    if i >= 10:
        t = b.tap("ar_timing", (1024,), np.int64)[:len(labels)].astype(np.float64) * 0.01     # us
        d = np.diff(t)
        acc = d if acc is None else acc + d
        n += 1
        total = t[-1] - t[0]
acc /= n
kinds = {}
for k in range(len(acc)):
    a, bb = labels[k].split(".")[-2:], labels[k + 1].split(".")[-2:]
    key = f"{a[0]}.{a[1]} -> {bb[0]}.{bb[1]}"
    kinds.setdefault(key, []).append(acc[k])
print(f"ar_dtype={ar_dtype}  frame span (first mark -> last mark): {acc.sum():.1f} us   fail={b.tap('ar_fail', (1,), np.int32)[0]}")
for key, v in kinds.items():
    print(f"  {key:24s} n={len(v):3d} mean {np.mean(v):6.2f} us  min {np.min(v):6.2f}  max {np.max(v):6.2f}  sum {np.sum(v):7.1f}")
if not v2:
    ns = b.tap("ar_timing", (1024,), np.int64)[900:908].astype(np.float64)
    print("sampler raw marks (us from entry: softmax done, [search done], ties done, final softmax done, argmax done, back in kernel):",
          [round((x - ns[0]) * 0.01, 2) for x in ns[1:7]], "probes", int(ns[7]))
print("timings:", b.timings())
if not v2:
    # Critical path of the frame: every hand-off edge with the weight bytes the consuming phase requests in front of its
    # polls (a wave's polls return behind its own weight loads: the edge costs max(hand-off, weight stream) + a round trip), the phases'
    # own arithmetic, the samplers; the rows sum to the frame span.  fp32 weights: 4 bytes, fp16: 2.
    wb = 2.0 if ar_dtype else 4.0
    MB = {"wqkv": 2304 * 768 * wb / 1e6, "wo": 768 * 768 * wb / 1e6, "w13": 4608 * 768 * wb / 1e6, "w2": 768 * 2304 * wb / 1e6, "head": 1000 * 768 * wb / 1e6}
    edge_w = {"D.out -> A.in": "wqkv", "A.out -> B1.in": None, "B1.out -> B1m.in": None, "B1m.out -> B2.in": "wo", "B2.out -> C.in": "w13", "C.out -> D.in": "w2",
              "FD.out -> FA.in": "wqkv", "FA.out -> FB.in": "wo", "FB.out -> FC.in": "w13", "FC.out -> FD.in": "w2", "FD.out -> FH.in": "head", "FH.out -> FS.in": None,
              "D.out -> hidden.in": None}
    rows, tot = [], 0.0
    for key, v in kinds.items():
        a, bb = key.split(" -> ")
        n_, mean, sm = len(v), float(np.mean(v)), float(np.sum(v))
        tot += sm
        if key in edge_w:
            w_ = edge_w[key]
            note = ("edge; the consumer streams %s = %.2f MB over 96 CUs in front of its polls -> %.2f TB/s if the edge were only the stream" %
                    (w_, MB[w_], MB[w_] / mean)) if w_ else "edge; no weights in front of the polls (KV prefetch / partial merge / logits)"
            rows.append(("edge", key, n_, mean, sm, note))
        elif a.startswith("FS") or a.startswith("hidden"):
            rows.append(("sampler" if a.startswith("FS") else "semantic head", key, n_, mean, sm, "nucleus sampler (threshold search ~6 us) + embedding of the token" if a.startswith("FS") else "semantic head rows + fast-AR input"))
        else:
            rows.append(("phase", key, n_, mean, sm, "products + RoPE / attention / epilogue of the phase"))
    print("critical path (kind, span, count, mean us, sum us):")
    for kind in ("edge", "phase", "sampler", "semantic head"):
        sub = [r for r in rows if r[0] == kind]
        if sub:
            print("  %-13s %4d spans  %7.1f us  (%.0f %% of the frame)" % (kind, sum(r[2] for r in sub), sum(r[4] for r in sub), 100 * sum(r[4] for r in sub) / tot))
            for r in sub:
                print("      %-24s x%3d  %5.2f us  = %6.1f us   %s" % (r[1], r[2], r[3], r[4], r[5]))
    slow_mb = 12 * (MB["wqkv"] + MB["wo"] + MB["w13"] + MB["w2"])
    fast_mb = 32 * (MB["wqkv"] + MB["wo"] + MB["w13"] + MB["w2"]) + 8 * MB["head"]
    print("  sum %.1f us; weight bytes per frame %.0f MB (slow %.0f once, fast %.0f = 8 passes over 4 layers + heads) -> %.2f TB/s over the frame; "
          "at the ~4.8 TB/s the 96 CUs pull (50 GB/s each) the stream alone is %.0f us of the span" % (tot, slow_mb + fast_mb, slow_mb, fast_mb, (slow_mb + fast_mb) / tot, (slow_mb + fast_mb) / 4.8))
