"""Step latency around a whole-batch re-prefill (every stream due on the same step: equal prompts), one-pass form against the per-slot
round-3 path.   python tools/reprefill_batch_probe.py [B ...]     AR_DTYPE=0|1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

eng = E.Engine(sw.generate_all(0, specs.all_specs()), ar_dtype=int(os.environ.get("AR_DTYPE", "0")))
lib = E.load_library()
R, msf, n = 107, 160, 70
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 64]:
    for mode in (1, 0):
        lib.sva_debug_configure(f"reprefill={mode}".encode())
        b = E.Batch(eng, n_streams=B, max_seq_frames=msf, buffer_frames=32, pipeline=True)
        ac, cc, style, timbre = synth_prompt(2000, R)
        for s in range(B):
            b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1 + s)
        b.begin()
        src = np.stack([synth_utterance(1000 + s % 5, 2048 * n) for s in range(B)])
        lat, pos = [], []
        for i in range(n):
            t1 = time.perf_counter()
            b.step(src[:, i * 2048:(i + 1) * 2048])
            b.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
            pos.append(int(b.tap("last_pos", (B,), np.int32)[0]))
        re = [i for i in range(1, n) if pos[i] < pos[i - 1]]
        base = float(np.median(lat[10:]))
        print(f"B={B} {'one pass, cached prefix' if mode else 'per slot, whole prompt '}: steady synchronous step {base:.2f} ms; re-prefill steps {re}: "
              + ", ".join(f"{lat[i]:.2f} ms ({lat[i] / base:.2f}x)" for i in re), flush=True)
        b.close()
lib.sva_debug_configure(b"reprefill=1")
