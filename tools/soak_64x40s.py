"""SURVEY 8d config 3 at full length: 64 streams x 40 s of audio (862 chunk-steps, default max_seq_frames = 768, so every stream re-prefills
once, all on the same step), one-pass re-prefill against the per-slot round-3 form: codes must be identical; step latencies reported.
   python tools/soak_64x40s.py        AR_DTYPE=0|1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

B, n = 64, 862
eng = E.Engine(sw.generate_all(0, specs.all_specs()), ar_dtype=int(os.environ.get("AR_DTYPE", "0")))
lib = E.load_library()
base = [synth_utterance(1000 + k, 2048 * n) for k in range(4)]
src = np.stack([base[s % 4] for s in range(B)])
ac, cc, style, timbre = synth_prompt(2000, 107)
res = {}
for mode in (1, 0):
    lib.sva_debug_configure(f"reprefill={mode}".encode())
    b = E.Batch(eng, n_streams=B, pipeline=True)
    for s in range(B):
        b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1 + s % 4)
    b.begin()
    lat, pos = [], []
    t0 = time.perf_counter()
    for i in range(n):
        t1 = time.perf_counter()
        b.step(src[:, i * 2048:(i + 1) * 2048])
        b.sync()
        lat.append((time.perf_counter() - t1) * 1e3)
        pos.append(int(b.tap("last_pos", (B,), np.int32)[0]))
    wall = time.perf_counter() - t0
    codes = np.stack([b.pred_codes(s) for s in range(B)])          # every decoded frame (860 per stream)
    b.close()
    re = [i for i in range(1, n) if pos[i] < pos[i - 1]]
    lat = np.array(lat)
    res[mode] = codes
    print(f"{'one pass, cached prefix' if mode else 'per slot, whole prompt '}: {n} synchronous steps of {B} streams in {wall:.2f} s = {B * n / wall:.0f} frames/s; "
          f"step ms p50 {np.median(lat):.2f} p99 {np.percentile(lat, 99):.2f} max {lat.max():.2f}; re-prefill steps {re}: "
          + ", ".join(f"{lat[i]:.1f} ms" for i in re), flush=True)
    assert all(np.array_equal(codes[s], codes[s % 4]) for s in range(B)), "streams with equal inputs diverged"
lib.sva_debug_configure(b"reprefill=1")
print("codes of the two re-prefill forms identical:", bool(np.array_equal(res[1], res[0])), "; differing entries:", int((res[1] != res[0]).sum()), "of", res[1].size)
