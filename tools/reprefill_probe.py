"""Latency of the re-prefill burst inside a stream (evaluations/infer_arvc.py:547-564: when the slow-AR position reaches
2 * max_seq_frames the KV cache is rebuilt from the prompt + the newest buffer_frames frames, M = 33 + 2 (R + buffer) rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

eng = E.Engine(sw.generate_all(0, specs.all_specs()), ar_dtype=int(os.environ.get("AR_DTYPE", "0")))
for R, msf in ((107, 160), (256, 330)):
    b = E.Batch(eng, n_streams=1, max_seq_frames=msf, buffer_frames=32)
    ac, cc, style, timbre = synth_prompt(2000, R)
    t0 = time.perf_counter()
    b.prefill_prompt(0, cc, ac, style, timbre, noise_seed=1)
    t_prefill = (time.perf_counter() - t0) * 1e3
    b.begin()
    src = synth_utterance(1000, 2048 * 120)
    lat, pos = [], []
    for i in range(120):
        t1 = time.perf_counter()
        b.step(src[i * 2048:(i + 1) * 2048][None])
        lat.append((time.perf_counter() - t1) * 1e3)
        pos.append(int(b.tap("last_pos", (1,), np.int32)[0]))
    re = [i for i in range(1, 120) if pos[i] < pos[i - 1]]
    base = float(np.median(lat[10:]))
    print(f"R={R} max_seq_frames={msf}: initial prefill M={33 + 2 * R} rows {t_prefill:.2f} ms (host-synchronous call); steady step {base:.2f} ms; "
          f"re-prefill steps {re} (M = {33 + 2 * (R + 32)} + 3 rows): " + ", ".join(f"{lat[i]:.2f} ms" for i in re))
    b.close()
