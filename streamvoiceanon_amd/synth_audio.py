"""Deterministic synthetic speech-like audio at the model rate (44.1 kHz).

SURVEY.md §8(d) config 3: "sum of 5 harmonics of f0 in U[90,260] Hz with 4 Hz vibrato +
pink-ish noise at -30 dB, peak 0.5".  On top of that the signal is cut into random
syllables (own f0 target, harmonic tilt and level, occasional noise bursts and short
pauses) so that a random-weight content encoder emits varied codes -- a stationary tone
collapses to one BSQ code and would make index-parity tests vacuous.

Only numpy's legacy MT19937 RandomState and elementary float64 math are used, so the
waveform is reproducible on the GPU box.
"""
from __future__ import annotations

import numpy as np

SR = 44100
SAMPLES_PER_FRAME = 2048      # evaluations/infer_arvc.py:28


def synth_utterance(seed: int, n_samples: int) -> np.ndarray:
    rng = np.random.RandomState(seed)
    n = int(n_samples)
    t = np.arange(n, dtype=np.float64) / SR
    f0 = np.empty(n)
    amp = np.empty(n)
    tilt = np.empty(n)
    noise_lvl = np.empty(n)
    pos = 0
    base_f0 = rng.uniform(90.0, 260.0)
    while pos < n:
        dur = int(rng.uniform(0.08, 0.30) * SR)
        end = min(n, pos + dur)
        kind = rng.rand()
        f0[pos:end] = base_f0 * rng.uniform(0.8, 1.25)
        tilt[pos:end] = rng.uniform(0.35, 0.85)
        if kind < 0.12:                       # pause
            amp[pos:end] = 0.02
            noise_lvl[pos:end] = 0.01
        elif kind < 0.30:                     # fricative-like burst
            amp[pos:end] = 0.15
            noise_lvl[pos:end] = rng.uniform(0.2, 0.5)
        else:                                 # voiced
            amp[pos:end] = rng.uniform(0.5, 1.0)
            noise_lvl[pos:end] = 0.0316       # -30 dB
        pos = end
    # smooth the piecewise-constant controls (5 ms box filter) to avoid clicks
    k = int(0.005 * SR)
    box = np.ones(k) / k
    f0 = np.convolve(f0, box, mode="same")
    amp = np.convolve(amp, box, mode="same")
    tilt = np.convolve(tilt, box, mode="same")
    noise_lvl = np.convolve(noise_lvl, box, mode="same")
    vib = 1.0 + 0.02 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
    phase = 2 * np.pi * np.cumsum(f0 * vib) / SR
    x = np.zeros(n)
    for h in range(5):
        x += (tilt ** h) * np.sin((h + 1) * phase + rng.uniform(0, 2 * np.pi))
    white = rng.randn(n)
    # one-pole "pink-ish" colouring  y[i] = a*y[i-1] + (1-a)*w[i], evaluated block-wise in
    # float64 (closed form inside a block, carried state across blocks)
    a = 0.85
    w = (1 - a) * white
    pink = np.empty(n)
    block = 4096
    pw = a ** np.arange(1, block + 1)
    kern = np.concatenate(([1.0], pw[:-1]))
    y = 0.0
    for s in range(0, n, block):
        seg = w[s:s + block]
        m = len(seg)
        out = pw[:m] * y + np.convolve(seg, kern[:m])[:m]
        pink[s:s + m] = out
        y = out[-1]
    x = amp * x + noise_lvl * pink * 4.0
    peak = np.abs(x).max()
    if peak > 0:
        x = 0.5 * x / peak
    return x.astype(np.float32)


def pad_to_chunks(wav: np.ndarray, chunk_frames: int = 1) -> np.ndarray:
    """Left-pad rule of stream_infer (evaluations/infer_arvc.py:648-649): pads a *full*
    extra chunk when the length is already a multiple."""
    c = SAMPLES_PER_FRAME * chunk_frames
    pad = c - (wav.shape[-1] % c)
    return np.concatenate([np.zeros(pad, dtype=wav.dtype), wav])


def synth_prompt(seed: int, n_frames: int = 107, codebook_size: int = 1000, vocab: int = 8192,
                 num_codebooks: int = 8, style_dim: int = 192, timbre_tokens: int = 32, timbre_dim: int = 128):
    """Synthetic speaker prompt: reference audio codes [8, R],
    reference content codes [R], style vector [192] ~ N(0,1), timbre latents [32, 128] ~ N(0,1).
    Stands in for the wav -> prompt path (CAM++, SparkTTS encoder, firefly.encode), which is
    row N1 of SURVEY.md §8f."""
    from . import synth_weights as sw

    a = sw.uniform01(seed, "prompt.audio", num_codebooks * n_frames).astype(np.float64)
    c = sw.uniform01(seed, "prompt.content", n_frames).astype(np.float64)
    audio_codes = np.floor(a * codebook_size).astype(np.int32).reshape(num_codebooks, n_frames)
    content_codes = np.floor(c * vocab).astype(np.int64)

    def gauss(tag, n):
        u1 = np.maximum(sw.uniform01(seed, tag + ".u1", n).astype(np.float64), 2.0 ** -25)
        u2 = sw.uniform01(seed, tag + ".u2", n).astype(np.float64)
        return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)).astype(np.float32)

    style = gauss("prompt.style", style_dim)
    timbre = gauss("prompt.timbre", timbre_tokens * timbre_dim).reshape(timbre_tokens, timbre_dim)
    return audio_codes, content_codes, style, timbre


def frame_noise(seed: int, frame: int, vocab: int = 8192, codebook_size: int = 1000, num_codebooks: int = 8):
    """Exp(1) sampler noise for decoded frame `frame` of utterance `seed`: (slow [vocab],
    fast [8, codebook_size]) -- the order in which decode_one_token_ar consumes its RNG
    (modules/dual_ar_stream.py:1183-1216).  Keyed by utterance and frame, never by rank/slot."""
    from . import synth_weights as sw

    slow = sw.exp1_noise(seed, frame, 0, vocab)
    fast = sw.exp1_noise(seed, frame, 1, num_codebooks * codebook_size).reshape(num_codebooks, codebook_size)
    return slow, fast
