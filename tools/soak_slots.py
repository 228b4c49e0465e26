"""Slot-consistency soak under load: in every configuration the upper half of the batch repeats the lower half's utterances / prompts / seeds, so
slot s and slot s + B/2 must agree bit for bit (PCM and codes) whatever the batch is doing around them -- the property that exposed the paired decode
attention's first build (wrong only in a loaded launch, never in isolation).  Covers the decode paths the policy table selects:
batched persistent kernel, multi-launch chain with and without a CU partition, fp32 / fp16 AR, chunk 1 / 4, skip_semantic on / off.

    python tools/soak_slots.py [chunks] [ragged]   (on the GPU box; default 120 chunks per configuration; "ragged": prompt lengths 60 .. 195 frames across the pairs, so
                                                    the streams of a batch sit at different KV positions and re-prefill on different steps)
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw          # noqa: E402
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance       # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"
W = {k: sw.generate(0, k, shp) for k, shp in specs.all_specs().items()}
W = {k: v for k, v in W.items() if v is not None}
R = 107
bad = 0
for ar_dtype, voc_dtype in ((0, 0), (1, 1)):
    eng = E.Engine(W, ar_dtype=ar_dtype, voc_dtype=voc_dtype)
    for B, chunk, skip in ((64, 1, True), (128, 1, True), (32, 1, False), (24, 1, True), (12, 1, True), (6, 1, True), (32, 4, True), (48, 4, True)):
        half = B // 2
        steps = max(8, n_chunks // chunk)
        utts = [synth_utterance(1000 + u, 2048 * chunk * steps) for u in range(half)]
        prompts = [synth_prompt(2000 + u, 60 + 9 * (u % 16) if ragged else R) for u in range(half)]
        b = E.Batch(eng, n_streams=B, chunk_frames=chunk, pipeline=True, skip_semantic=skip)
        for s in range(B):
            ac, cc, style, timbre = prompts[s % half]
            b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=9000 + s % half)
        b.begin()
        x = np.stack([utts[s % half] for s in range(B)])
        t0 = time.time()
        out = b.stream_chunks(x)
        dt = time.time() - t0
        b.close()
        d = (out[:half] != out[half:]).any(axis=1)
        ok = not d.any() and np.isfinite(out).all() and np.abs(out).max() > 0.01
        bad += 0 if ok else 1
        print(f"ar_dtype {ar_dtype} voc_dtype {voc_dtype} streams {B:3d} chunk {chunk} skip_semantic {int(skip)} steps {steps:4d}: "
              f"{'identical' if ok else 'MISMATCH in %d slot pairs' % int(d.sum())}  ({B * chunk * steps / dt:.0f} frames/s incl. host copies)", flush=True)
    eng.close()
print("soak:", "all slot pairs identical" if bad == 0 else f"{bad} configuration(s) FAILED")
sys.exit(1 if bad else 0)
