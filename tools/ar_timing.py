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
