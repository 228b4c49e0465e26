"""Where calculate_prompt's latency goes (evaluations/infer_arvc.py:382-441): wall time of each seam, best of 3, R = 107 and 256."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from streamvoiceanon_amd import audio_io  # noqa: E402
from streamvoiceanon_amd.synth_audio import synth_utterance  # noqa: E402

w = bench._wrapper_with_prompt_path()


def best(fn, n=5):
    v = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        v.append((time.perf_counter() - t0) * 1e3)
    return min(v), r


for R in (107, 256):
    wav = synth_utterance(7300 + R, 2048 * R + 100)
    t_rs, ref16 = best(lambda: audio_io.resample(wav, w.sr, w.RESAMPLE_FREQ))
    t_st, _ = best(lambda: w.calculate_style_vec(ref16))
    t_tm, _ = best(lambda: w.calculate_timbre_latent(ref16))
    t_ac, _ = best(lambda: w.wav2target_fn(wav))
    t_cc, _ = best(lambda: w.encode_content(wav))
    t_all, _ = best(lambda: w.calculate_prompt(wav[None], alpha=1.0))
    print(f"R={R}: resample {t_rs:.2f} ms | style (fbank + CAM++) {t_st:.2f} | timbre (mel + ECAPA + perceiver + FSQ) {t_tm:.2f} | "
          f"firefly.encode {t_ac:.2f} | content encode {t_cc:.2f} | calculate_prompt {t_all:.2f}")
