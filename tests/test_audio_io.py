"""Host-side audio I/O: RIFF/WAVE codec round trips and the restated torchaudio polyphase resampler.
torchaudio is not installed in the build container, so the resampler is checked (a) against an independent torch conv1d
evaluation of the same published algorithm and (b) through the properties that algorithm guarantees."""
import math
import wave

import numpy as np
import pytest
import torch

from streamvoiceanon_amd import audio_io as A


def _torch_resample(x, orig, new, lpw=6, rolloff=0.99):
    """torchaudio.functional.resample, v2.4.0, evaluated with torch ops the way the library does (conv1d, stride = orig)."""
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    width = math.ceil(lpw * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lpw, lpw)
    window = torch.cos(t * math.pi / lpw / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)
    k = k.to(torch.float32)
    w = torch.nn.functional.pad(torch.from_numpy(x)[None, None], (width, width + orig))
    y = torch.nn.functional.conv1d(w, k, stride=orig)           # [1, new, L]
    y = y.transpose(1, 2).reshape(1, -1)
    return y[0, :math.ceil(new * x.shape[0] / orig)].numpy()


@pytest.mark.parametrize("orig,new", [(44100, 16000), (24000, 44100), (48000, 44100), (16000, 44100)])
def test_resample_matches_conv1d_evaluation(orig, new):
    rng = np.random.default_rng(orig + new)
    x = rng.standard_normal(9000).astype(np.float32)
    y = A.resample(x, orig, new)
    ref = _torch_resample(x, orig, new)
    assert y.shape == ref.shape == (math.ceil(new * 9000 / orig),)
    np.testing.assert_allclose(y, ref, atol=2e-6)


def test_resample_properties():
    sr, n = 44100, 44100
    t = np.arange(n) / sr
    tone = np.sin(2 * np.pi * 1000 * t).astype(np.float32)
    y = A.resample(tone, sr, 16000)
    assert y.shape == (16000,)
    mid = y[2000:14000]
    assert abs(np.abs(mid).max() - 1.0) < 0.01                                   # pass band preserved
    f = np.abs(np.fft.rfft(mid * np.hanning(mid.size)))
    assert abs(np.argmax(f) * 16000 / mid.size - 1000) < 3                        # still 1 kHz
    alias = A.resample(np.sin(2 * np.pi * 12000 * t).astype(np.float32), sr, 16000)
    assert np.abs(alias[2000:14000]).max() < 0.01                                 # 12 kHz is beyond the 16 kHz Nyquist: > 40 dB down
    assert np.array_equal(A.resample(tone, 44100, 44100), tone)
    stereo = np.stack([tone, -tone])
    ys = A.resample(stereo, sr, 16000)
    np.testing.assert_allclose(ys[1], -ys[0], atol=1e-6)                          # linear, applied on the last axis


def test_wav_round_trips(tmp_path):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 1000)) * 0.3).astype(np.float32)
    p = str(tmp_path / "f32.wav")
    A.write_wav(p, x, 44100)
    y, sr = A.read_wav(p)
    assert sr == 44100 and np.array_equal(x, y)
    mono, sr2 = A.load(p, sr=44100)
    np.testing.assert_allclose(mono, x.mean(0), atol=1e-7)
    # 16-bit PCM written by the standard library
    p16 = str(tmp_path / "pcm16.wav")
    q = np.clip(np.round(x[0] * 32768), -32768, 32767).astype("<i2")
    with wave.open(p16, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes(q.tobytes())
    y16, sr16 = A.read_wav(p16)
    assert sr16 == 24000 and np.array_equal(y16[0], q.astype(np.float32) / 32768.0)
    up, sr3 = A.load(p16, sr=44100)
    assert sr3 == 44100 and up.shape == (math.ceil(147 * 1000 / 80),)
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.wav"), "wb").write(b"not a wav file at all")
        A.read_wav(str(tmp_path / "bad.wav"))


def test_realtime_session_lazy_reprefill_logic():
    """streamvoiceanon_amd.realtime (real-time-gui.py:32-49): host logic only -- prompt and caches are rebuilt iff the reference
    name or the block size changed; the block goes through process_one_chunk unchanged in kind and length."""
    from streamvoiceanon_amd.realtime import RealtimeSession

    calls = []

    class Stub:
        def prefill_prompt(self, ref, max_prompt_frames, delay, alpha):
            calls.append(("prefill", ref.shape, max_prompt_frames, delay, alpha))

        def setup_stream_caches(self, **kw):
            calls.append(("setup", kw["encode_window_frames"], kw["decode_window_frames"], kw["max_seq_frames"], kw["buffer_frames"],
                          kw["decode_chunk_frames"]))

        def process_one_chunk(self, block):
            calls.append(("chunk", block.shape))
            return block * 0.5

    s, m = RealtimeSession(), Stub()
    ref = np.zeros(5000, np.float32)
    x = np.ones(2048, np.float32)
    y = s.custom_infer(m, ref, "a", x, n_frame_delay=3, alpha=0.7)
    assert y.shape == (2048,) and np.allclose(y, 0.5)
    assert calls == [("prefill", (5000,), 64, 3, 0.7), ("setup", 64, 64, 768, 32, 1), ("chunk", (1, 2048))]
    s.custom_infer(m, ref, "a", x)
    assert [c[0] for c in calls] == ["prefill", "setup", "chunk", "chunk"] and s.prefills == 1
    s.custom_infer(m, ref, "b", x)
    s.custom_infer(m, ref, "b", np.ones(4096, np.float32))
    assert s.prefills == 3 and calls[-2] == ("setup", 64, 64, 768, 32, 2)
    with pytest.raises(AssertionError):
        s.custom_infer(m, ref, "b", np.ones(1000, np.float32))


def test_gui_presets_schema_and_apply(tmp_path):
    """configs/presets.json schema of the reference GUI (real-time-gui.py:629-662): name -> {description, alpha, block_frame,
    n_frame_delay}; unknown names and "Custom" leave the settings alone, missing keys keep their value, unreadable file -> {}."""
    import json

    from streamvoiceanon_amd.realtime import GuiSettings, apply_preset, load_presets

    assert load_presets(str(tmp_path / "missing.json")) == {}
    f = tmp_path / "presets.json"
    f.write_text(json.dumps({"Max Quality": {"description": "d", "alpha": 1.0, "block_frame": 1, "n_frame_delay": 4},
                             "Only Alpha": {"alpha": 0.25}}))
    presets = load_presets(str(f))
    s0 = GuiSettings()
    assert (s0.alpha, s0.block_frame, s0.n_frame_delay) == (0.7, 1, 2)
    assert apply_preset(s0, "Custom", presets) == s0 and apply_preset(s0, "nope", presets) == s0
    s1 = apply_preset(s0, "Max Quality", presets)
    assert (s1.alpha, s1.block_frame, s1.n_frame_delay) == (1.0, 1, 4) and s0.n_frame_delay == 2
    s2 = apply_preset(s1, "Only Alpha", presets)
    assert (s2.alpha, s2.block_frame, s2.n_frame_delay) == (0.25, 1, 4)
