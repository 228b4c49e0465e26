"""Host-side audio I/O of the drop-in: wav read / write and the polyphase resampler.

The reference loads audio with ``librosa.load(path, sr=44100)`` (evaluations/infer_arvc.py:254, 274, 615, 624),
resamples the prompt to 16 kHz with ``torchaudio.functional.resample`` (:276, 291, 314, 395, 416) and writes results
with ``torchaudio.save`` (:378, 687).  Neither library is vendored in the reference nor installed here, so:

* ``resample`` restates the published algorithm of torchaudio==2.4.0 (requirements.txt:7) ``functional.resample`` with
  its defaults (``sinc_interp_hann``, lowpass_filter_width 6, rolloff 0.99) -- parity unpinned (no torchaudio in the
  build container to generate vectors); tests check the properties the algorithm guarantees.
* ``load`` resamples with that same kernel; librosa's default ``soxr_hq`` is a different (closed-form unavailable)
  filter, so a file loaded here differs from librosa's at the resampler's stop-band level.  File I/O sits outside
  the parity contract of the hot path (tests feed arrays).

Plain numpy; nothing here touches the GPU.
"""
import functools
import math
import struct

import numpy as np


@functools.lru_cache(maxsize=16)
def _sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.functional._get_sinc_resample_kernel (v2.4.0), sinc_interp_hann; orig/new already reduced
    by their gcd.  Returns (kernels [new, 2*width + orig] float64, width)."""
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base_freq, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t)
    return kern * window * scale, width


def resample(x, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample(waveform, orig_freq, new_freq) on the last axis: zero-pad (width, width + orig),
    strided correlation with the `new` phase kernels, interleave, crop to ceil(new * length / orig)."""
    x = np.asarray(x, dtype=np.float32)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kern, width = _sinc_resample_kernel(orig, new, lowpass_filter_width, rolloff)        # (cached per rate pair)
    kern = kern.astype(np.float32)
    lead = x.shape[:-1]
    w = x.reshape(-1, x.shape[-1])
    length = w.shape[1]
    padded = np.pad(w, ((0, 0), (width, width + orig)))
    klen = kern.shape[1]
    n_out = (padded.shape[1] - klen) // orig + 1
    # frames [rows, n_out, klen] as a strided view, then one matmul against the phase kernels
    s0, s1 = padded.strides
    frames = np.lib.stride_tricks.as_strided(padded, shape=(w.shape[0], n_out, klen), strides=(s0, s1 * orig, s1), writeable=False)
    # materialised once (a few hundred KB) and multiplied on ONE BLAS thread: the product is 75 MFLOP for a 5 s prompt -- 0.8 ms on one
    # core, 15-100 ms when OpenBLAS fans it out over a many-core host (that was 40 % of calculate_prompt's latency)
    fc, kt = np.ascontiguousarray(frames), np.ascontiguousarray(kern.T)
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                    # plain numpy: same result, the library's own threading
        out = fc @ kt
    else:
        with threadpool_limits(limits=1, user_api="blas"):
            out = fc @ kt                                  # [rows, n_out, new]
    out = out.reshape(w.shape[0], n_out * new)
    target = int(math.ceil(new * length / orig))
    return out[:, :target].reshape(*lead, target).astype(np.float32)


# ---- RIFF / WAVE ------------------------------------------------------------------------------------------------
def read_wav(path):
    """-> (float32 array [channels, samples] in [-1, 1), sample_rate).  PCM 8/16/24/32-bit and IEEE float 32/64."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and size >= 26:               # WAVE_FORMAT_EXTENSIBLE: real tag = first 2 bytes of the sub-format GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 3:
        a = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif tag == 1:
        if bits == 8:
            a = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            a = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            a = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            a = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = a.shape[0] // ch
    return a[:n * ch].reshape(n, ch).T.copy(), sr


def write_wav(path, x, sample_rate: int):
    """float32 array [samples] or [channels, samples] -> 32-bit IEEE-float WAVE (what torchaudio.save writes for a
    float tensor, infer_arvc.py:378, 687)."""
    a = np.asarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    ch, n = a.shape
    body = a.T.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, int(sample_rate), int(sample_rate) * ch * 4, ch * 4, 32)
    fact = struct.pack("<I", n)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact + b"data" + struct.pack("<I", len(body)) + body
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def load(path, sr: int = 44100):
    """librosa.load(path, sr=sr) stand-in: mono (channel mean), float32, resampled to `sr` -> (array [samples], sr)."""
    a, file_sr = read_wav(path)
    mono = a.mean(axis=0).astype(np.float32)
    if file_sr != sr:
        mono = resample(mono, file_sr, sr)
    return mono, sr
