"""Pins for the third-party front-ends the reference calls but that are not installable here (torchaudio, librosa):
the build-owned restatements (oracle/sva_oracle.py, oracle/prompt_oracle.py, streamvoiceanon_amd/audio_io.py) are compared
with INDEPENDENT implementations of the same published algorithms that are present in this image --
transformers.audio_utils (HTK / Slaney / Kaldi mel filter banks, Kaldi-style and librosa-style spectrograms) and
scipy.signal.resample_poly.  The fixtures tests/golden/* feed the reference networks the restated features; these tests say the
restated features are the library's.

reference call sites: modules/vqgan/spectrogram.py:89-101 (melscale_fbanks), evaluations/infer_arvc.py:179-197 (kaldi.fbank),
modules/bicodec_speaker_encoder/speaker_encoder.py / infer_arvc.py:199-218 (MelSpectrogram), infer_arvc.py:254-278 (librosa.load /
torchaudio resample)."""
import math

import numpy as np
import pytest
import torch

au = pytest.importorskip("transformers.audio_utils")

from oracle import prompt_oracle as PO
from oracle import sva_oracle as O
from streamvoiceanon_amd import audio_io as A


def _tone_mix(n, sr, seed, fmax):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    x = np.zeros(n)
    for _ in range(12):
        x += rng.uniform(0.05, 0.3) * np.sin(2 * np.pi * rng.uniform(60.0, fmax) * t + rng.uniform(0, 6.28))
    return x


def test_slaney_filterbank_matches_transformers():
    """torchaudio.functional.melscale_fbanks(1025, 0, 22050, 160, 44100, norm='slaney', mel_scale='slaney')"""
    ours = O.slaney_mel_fb().numpy()                       # [1025, 160]
    ref = au.mel_filter_bank(num_frequency_bins=1025, num_mel_filters=160, min_frequency=0.0, max_frequency=22050.0,
                             sampling_rate=44100, norm="slaney", mel_scale="slaney")
    assert ours.shape == ref.shape == (1025, 160)
    assert np.abs(ours - ref).max() <= 5e-7, np.abs(ours - ref).max()
    # the 16 kHz bank of the timbre encoder's MelSpectrogram
    ours16 = O.slaney_mel_fb(n_freqs=513, f_min=10.0, f_max=8000.0, n_mels=128, sample_rate=16000).numpy()
    ref16 = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=10.0, max_frequency=8000.0,
                               sampling_rate=16000, norm="slaney", mel_scale="slaney")
    assert np.abs(ours16 - ref16).max() <= 5e-7, np.abs(ours16 - ref16).max()


def test_kaldi_fbank_matches_transformers_kaldi_spectrogram():
    """torchaudio.compliance.kaldi.fbank(wave, num_mel_bins=80, dither=0, sample_frequency=16000) vs transformers' Kaldi-compatible
    spectrogram (povey window, pre-emphasis 0.97, DC removal, Kaldi mel scale, triangles in mel space)."""
    x = (_tone_mix(16000 * 2, 16000, 3, 7000.0) * 0.3 + 0.05 * np.random.default_rng(1).standard_normal(32000)).astype(np.float32)   # tones over a noise floor
    ours = PO.kaldi_fbank(torch.from_numpy(x)).numpy()                    # [frames, 80]
    fb = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20.0, max_frequency=8000.0, sampling_rate=16000,
                            norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    ref = au.spectrogram(x.astype(np.float64), au.window_function(400, "povey", periodic=False), frame_length=400, hop_length=160, fft_length=512,
                         power=2.0, center=False, preemphasis=0.97, mel_filters=fb, log_mel="log", mel_floor=float(np.finfo(np.float32).eps),
                         remove_dc_offset=True).T
    assert ours.shape == ref.shape, (ours.shape, ref.shape)
    d = np.abs(ours - ref).max()
    print(f"kaldi fbank: max |delta| {d:.2e} on log-energies in [{ref.min():.1f}, {ref.max():.1f}]")
    assert d <= 1e-3, d               # fp32 (restatement, as torchaudio computes) vs fp64 arithmetic


def test_mel_spectrogram_16k_matches_transformers():
    """torchaudio.transforms.MelSpectrogram(16000, n_fft=1024, win_length=640, hop_length=320, f_min=10, n_mels=128, power=1,
    norm='slaney', mel_scale='slaney'): centered, reflect-padded STFT magnitude through the Slaney bank."""
    x = (_tone_mix(16000, 16000, 5, 7000.0) * 0.3).astype(np.float32)
    ours = PO.mel_spectrogram_16k(torch.from_numpy(x)).numpy()            # [128, frames]
    fb = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=10.0, max_frequency=8000.0, sampling_rate=16000,
                            norm="slaney", mel_scale="slaney")
    win = np.zeros(1024)
    win[192:832] = au.window_function(640, "hann", periodic=True)           # torch.stft centres a short window inside n_fft
    ref = au.spectrogram(x.astype(np.float64), win, frame_length=1024, hop_length=320, fft_length=1024, power=1.0, center=True, pad_mode="reflect",
                         mel_filters=fb, mel_floor=0.0)
    assert ours.shape == ref.shape, (ours.shape, ref.shape)
    scale = np.abs(ref).max()
    assert np.abs(ours - ref).max() <= 2e-5 * scale, (np.abs(ours - ref).max(), scale)


@pytest.mark.parametrize("orig,new", [(24000, 44100), (44100, 16000), (48000, 44100)])
def test_resampler_vs_scipy_polyphase_on_band_limited_input(orig, new):
    """audio_io.resample (torchaudio's sinc_interp_hann kernel, what audio_io.load uses in place of librosa's soxr_hq) against
    scipy.signal.resample_poly (Kaiser-windowed polyphase FIR) on a signal band-limited well below both Nyquist rates: two
    different anti-aliasing filters agree in their common pass band.  The bound is the stated deviation of `load`: <= -50 dB of the
    signal's RMS (measured -60 .. -66 dB); the short Hann-windowed sinc (width 6) rolls off earlier than soxr_hq near Nyquist, which
    is where the two loaders differ on real audio."""
    sig = pytest.importorskip("scipy.signal")
    n = orig // 2
    x = _tone_mix(n, orig, orig + new, 0.35 * min(orig, new) / 2).astype(np.float32)
    y = A.resample(x, orig, new)
    g = math.gcd(orig, new)
    ref = sig.resample_poly(x.astype(np.float64), new // g, orig // g)
    m = min(len(y), len(ref))
    lo, hi = m // 10, m - m // 10                         # away from the edges (different edge conventions)
    err = y[lo:hi] - ref[lo:hi]
    db = 10 * np.log10(np.mean(err ** 2) / np.mean(ref[lo:hi] ** 2))
    print(f"resample {orig}->{new}: deviation {db:.1f} dB re signal RMS")
    assert db <= -50.0, db
