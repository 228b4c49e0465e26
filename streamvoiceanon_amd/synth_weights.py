"""Platform-independent synthetic weights.

No pretrained checkpoint can exist in this repository (SURVEY.md §7 hard part 6), so every
parity fixture is defined on weights that both the oracle container and the GPU box
regenerate bit-identically from ``(seed, state-dict key, flat index)``.  Only integer
arithmetic (a splitmix64-style hash on uint64 numpy arrays) and one exact int->float
conversion are used, so the result does not depend on the BLAS/libm of the host.

State-dict keys are the reference's on-disk names (SURVEY.md §8a "DualAR state-dict keys");
they are prefixed here with the network they belong to: ``arvc.``, ``tok.``, ``voc.``.
"""
from __future__ import annotations

import math
import re

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


_CH = 1 << 18
_BASE = np.arange(1, _CH + 1, dtype=np.uint64) * _GOLD


def _affine_u24(seed: int, name: str, n: int, lo: float, hi: float) -> np.ndarray:
    """float32 array  lo + (hi-lo) * k/2^24  with k the top 24 bits of a splitmix64-style
    hash of (seed, name, index).  Chunked so the working set stays in cache."""
    key = np.uint64((_fnv1a64(name) ^ ((int(seed) + 1) * 0xD6E8FEB86659FD93)) & 0xFFFFFFFFFFFFFFFF)
    out = np.empty(n, dtype=np.float32)
    z = np.empty(_CH, np.uint64)
    t = np.empty(_CH, np.uint64)
    f = np.empty(_CH, np.float64)
    scale = (hi - lo) / 16777216.0
    with np.errstate(over="ignore"):
        for s in range(0, n, _CH):
            m = min(_CH, n - s)
            zz, tt, ff = z[:m], t[:m], f[:m]
            np.add(_BASE[:m], key + np.uint64(s) * _GOLD, out=zz)
            np.right_shift(zz, np.uint64(30), out=tt); np.bitwise_xor(zz, tt, out=zz); np.multiply(zz, _M1, out=zz)
            np.right_shift(zz, np.uint64(27), out=tt); np.bitwise_xor(zz, tt, out=zz); np.multiply(zz, _M2, out=zz)
            np.right_shift(zz, np.uint64(31), out=tt); np.bitwise_xor(zz, tt, out=zz)
            np.right_shift(zz, np.uint64(40), out=zz)
            ff[:] = zz
            ff *= scale
            ff += lo
            out[s:s + m] = ff
    return out


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """n deterministic uniforms in [0, 1) with 24 random bits each (exact in float32)."""
    return _affine_u24(seed, name, n, 0.0, 1.0)


def _sym(seed, name, shape, amp):
    n = int(np.prod(shape)) if len(shape) else 1
    return _affine_u24(seed, name, n, -amp, amp).reshape(shape)


def _rng(seed, name, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    return _affine_u24(seed, name, n, lo, hi).reshape(shape)


# tensors that exist in the reference state dicts but are never touched by the hot path
# (tokenizer `head` 73 M and `post_module` 27 M are dead at inference)
_SKIP = (
    re.compile(r"^tok\.head\."),
    re.compile(r"^tok\.quantizer\.post_module\."),
    re.compile(r"\.(freqs_cis|fast_freqs_cis|causal_mask|mask|zero|codebook|implicit_codebook|scales|_levels|_basis)$"),
    re.compile(r"^tok\.spec_transform\.|^voc\.spec_transform\."),
)

# per-tensor gain overrides (regex on the full name) so activations stay O(1) through
# 18+ ConvNeXt blocks, 12+4 transformer layers and the HiFiGAN stack, and the tanh
# output is not saturated (SURVEY.md §8c "Weights")
_GAINS = (
    (re.compile(r"\.(output|fast_output)\.weight$"), 2.5),
    (re.compile(r"^voc\.head\.resblocks\..*\.convs2\."), 0.7),
    (re.compile(r"^voc\.head\.resblocks\..*\.convs1\."), 1.5),
    (re.compile(r"^voc\.head\.ups\."), 1.4),
    (re.compile(r"^voc\.head\.conv_pre\."), 1.5),
    (re.compile(r"^voc\.head\.conv_post\."), 1.0),
    (re.compile(r"\.pwconv2\.weight$"), 1.0),
    (re.compile(r"residual_bsq\.rvqs\.\d+\.project_in\.weight$"), 1.0),
)


def is_hot(name: str) -> bool:
    return not any(p.search(name) for p in _SKIP)


def generate(seed: int, name: str, shape) -> np.ndarray | None:
    """Synthetic value of tensor `name` (prefixed state-dict key) or None if not generated."""
    shape = tuple(int(s) for s in shape)
    if not is_hot(name):
        return None
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":                 # BatchNorm statistics of the speaker encoders (prompt path)
        return _rng(seed, name, shape, 0.6, 1.4)
    if leaf == "running_mean":
        return _sym(seed, name, shape, 0.1)
    if leaf == "latents":                     # PerceiverResampler latents (init std 0.02; O(1) here so that they matter)
        return _sym(seed, name, shape, 0.5)
    if leaf == "gamma" and name.startswith("timbre."):
        return _rng(seed, name, shape, 0.8, 1.2)
    if leaf == "gamma":                       # ConvNeXt / LayerScale gammas
        return _rng(seed, name, shape, 0.1, 0.5)
    if "embedding" in name and len(shape) == 2:
        return _sym(seed, name, shape, 0.8)
    if len(shape) <= 1:
        if leaf == "weight":                  # LayerNorm / RMSNorm scales
            return _rng(seed, name, shape, 0.8, 1.2)
        return _sym(seed, name, shape, 0.1)   # biases
    # linear / conv kernels
    gain = 1.0
    for pat, g in _GAINS:
        if pat.search(name):
            gain = g
            break
    if re.search(r"\.ups\.\d+\.conv\.weight$", name):
        fan_in = shape[0] * 2                 # ConvTranspose1d [Cin, Cout, k], k = 2*stride
    elif re.search(r"quantizer\.upsample\.\d+\.0\.conv\.weight$", name):
        fan_in = shape[0]                     # ConvTranspose1d k = stride
    else:
        fan_in = int(np.prod(shape[1:]))
    amp = gain * math.sqrt(3.0 / fan_in)
    return _sym(seed, name, shape, amp)


def _mix64(z: int) -> int:
    z &= 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def noise_key(utt_seed: int, frame: int, kind: int) -> int:
    """64-bit key of the sampler-noise stream for (utterance, decoded frame, kind) with
    kind 0 = semantic/slow head, 1 = fast codebooks.  Pure integer arithmetic: the HIP
    engine's on-device generator (csrc/kernels.hip: sampler noise) evaluates the same formula."""
    z = ((int(utt_seed) + 1) * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
    z ^= ((int(frame) + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z ^= ((int(kind) + 1) * 0xC2B2AE3D27D4EB4F) & 0xFFFFFFFFFFFFFFFF
    return _mix64(z)


def u24_from_key(key: int, n: int) -> np.ndarray:
    """top 24 bits of mix64(key + (i+1)*GOLD), i < n, as uint32."""
    out = np.empty(n, dtype=np.uint32)
    k = np.uint64(key)
    with np.errstate(over="ignore"):
        z = k + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    out[:] = (z >> np.uint64(40)).astype(np.uint32)
    return out


def exp1_noise(utt_seed: int, frame: int, kind: int, n: int) -> np.ndarray:
    """Deterministic Exp(1) noise for the sampler (the reference draws it from torch's global
    generator, modules/dual_ar_stream.py:1095; parity is defined given the same noise).
    u = max(k, 0.5) / 2^24 with k the 24-bit hash; noise = -log(u) (float64 log, cast to f32)."""
    k = u24_from_key(noise_key(utt_seed, frame, kind), n).astype(np.float64)
    u = np.maximum(k, 0.5) * (1.0 / 16777216.0)
    return (-np.log(u)).astype(np.float32)


def generate_all(seed: int, specs: dict) -> dict:
    """name -> float32 array for every (prefixed state-dict name, shape) of `specs` (tensors the hot path never touches are skipped)."""
    out = {}
    for name, shape in specs.items():
        arr = generate(seed, name, shape)
        if arr is not None:
            out[name] = arr
    return out
