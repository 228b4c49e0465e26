"""CPU restatement of the two speaker-embedding encoders of the prompt path (SURVEY.md 8f N1 iii / iv).

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/make_golden.py).  PyTorch-CPU fp32, functional, flat weight dicts keyed
by the reference's state-dict names with a network prefix ("style." = CAM++, "timbre." = SparkTTS SpeakerEncoder).

  style  : evaluations/infer_arvc.py:179-211 calculate_style_vec = Kaldi fbank (80 bins, 16 kHz) - mean -> CAMPPlus
           (modules/campplus/DTDNN.py:50-137, layers.py) -> [192]
  timbre : evaluations/infer_arvc.py:213-223 calculate_timbre_latent = SpeakerEncoder.tokenize_wav
           (modules/bicodec_speaker_encoder/speaker_encoder.py:136-144): MelSpectrogram -> ECAPA-TDNN latent
           (ecapa_tdnn.py:150-213) -> PerceiverResampler (perceiver_encoder.py:287-341) -> ResidualFSQ [4]^6
           (fsq/residual_fsq.py, finite_scalar_quantization.py:126-162) -> zq.mT [32, 128]

Third-party front-ends that are NOT in /root/reference (torchaudio==2.4.0, requirements.txt:7) are restated from their published
algorithms -- parity unpinned for these two functions (no torchaudio in this image): `kaldi_fbank`
(torchaudio.compliance.kaldi.fbank with the reference's arguments) and `mel_spectrogram_16k`
(torchaudio.transforms.MelSpectrogram as configured in configs/hydra_arcs/sv/sparktts_speaker_encoder.yaml).  The networks
behind them are pinned by fixtures captured from the reference modules (tools/make_golden.py prompt_encoders).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .sva_oracle import slaney_mel_fb

BN_EPS = 1e-5


# =========================================================================================
# front-ends (third-party restatements)
# =========================================================================================
def kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0) -> torch.Tensor:
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp = 1): triangular filters on the Kaldi mel scale
    1127 ln(1 + f / 700), [num_bins, padded / 2 + 1] (last column zero)."""
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    nfb = padded // 2
    bw = sr / padded

    def mel(f):
        return 1127.0 * math.log(1.0 + f / 700.0)

    mlo, mhi = mel(low), mel(high)
    delta = (mhi - mlo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mlo + b * delta, mlo + (b + 1.0) * delta, mlo + (b + 2.0) * delta
    m = (1127.0 * (1.0 + bw * torch.arange(nfb) / 700.0).log()).unsqueeze(0)
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    banks = torch.max(torch.zeros(1), torch.min(up, down))
    return F.pad(banks, (0, 1))


def kaldi_fbank(wave: torch.Tensor, num_mel_bins=80, sample_frequency=16000.0) -> torch.Tensor:
    """kaldi.fbank(waveform[1, N], num_mel_bins=80, dither=0, sample_frequency=16000) with torchaudio's defaults (25 ms povey
    window, 10 ms shift, snip_edges, remove_dc_offset, preemphasis 0.97, round_to_power_of_two, power spectrum, log) ->
    [frames, 80]."""
    x = wave.reshape(-1).float()
    ws, sh, pad = int(sample_frequency * 0.025), int(sample_frequency * 0.010), 512
    m = 1 + (x.shape[0] - ws) // sh if x.shape[0] >= ws else 0
    fr = x.unfold(0, ws, sh)[:m]                                   # [m, 400]
    fr = fr - fr.mean(dim=1, keepdim=True)                         # remove_dc_offset
    prev = F.pad(fr.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)[:, :-1]
    fr = fr - 0.97 * prev                                          # preemphasis
    win = torch.hann_window(ws, periodic=False).pow(0.85)          # povey
    fr = F.pad(fr * win, (0, pad - ws))
    spec = torch.fft.rfft(fr).abs().pow(2.0)                       # [m, 257]
    mel = spec @ kaldi_mel_banks(num_mel_bins, pad, sample_frequency).T
    return torch.max(mel, torch.tensor(torch.finfo(torch.float32).eps)).log()


def mel_spectrogram_16k(wave: torch.Tensor) -> torch.Tensor:
    """torchaudio.transforms.MelSpectrogram(16000, n_fft=1024, win_length=640, hop_length=320, f_min=10, n_mels=128, power=1,
    norm='slaney', mel_scale='slaney') with its defaults (centered reflect padding, periodic Hann window zero-padded to n_fft):
    [N] -> [128, 1 + N // 320]."""
    x = wave.reshape(1, -1).float()
    win = torch.hann_window(640)
    spec = torch.stft(x, 1024, hop_length=320, win_length=640, window=win, center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True).abs()[0]          # [513, frames]
    fb = slaney_mel_fb(n_freqs=513, f_min=10.0, f_max=8000.0, n_mels=128, sample_rate=16000)       # [513, 128]
    return fb.T @ spec


# =========================================================================================
# CAM++ (modules/campplus)
# =========================================================================================
def _bn(x, W, p, affine=True, dim=1):
    """eval-mode BatchNorm over channel dim `dim`"""
    shape = [1] * x.dim()
    shape[dim] = -1
    y = (x - W[p + "running_mean"].view(shape)) / torch.sqrt(W[p + "running_var"].view(shape) + BN_EPS)
    if affine:
        y = y * W[p + "weight"].view(shape) + W[p + "bias"].view(shape)
    return y


def _res_block(x, W, p, stride):
    """BasicResBlock, layers.py:227-266 (stride on the frequency axis only)"""
    out = F.relu(_bn(F.conv2d(x, W[p + "conv1.weight"], stride=(stride, 1), padding=1), W, p + "bn1."))
    out = _bn(F.conv2d(out, W[p + "conv2.weight"], padding=1), W, p + "bn2.")
    sc = x
    if (p + "shortcut.0.weight") in W:
        sc = _bn(F.conv2d(x, W[p + "shortcut.0.weight"], stride=(stride, 1)), W, p + "shortcut.1.")
    return F.relu(out + sc)


def campplus(feat: torch.Tensor, W: dict, p: str = "style.") -> torch.Tensor:
    """CAMPPlus.forward (DTDNN.py:130-137), batch 1: feat [T, 80] -> [192]; the statistics pooling runs over the first T // 2
    frames, the x_lens calculate_style_vec passes (evaluations/infer_arvc.py:196-201 -> masked_statistics_pooling,
    layers.py:33-43)."""
    x = feat.T[None, None]                                         # [1, 1, F, T]
    h = p + "head."
    x = F.relu(_bn(F.conv2d(x, W[h + "conv1.weight"], padding=1), W, h + "bn1."))
    for li, name in enumerate(("layer1.", "layer2.")):
        for bi in range(2):
            x = _res_block(x, W, h + name + f"{bi}.", 2 if bi == 0 else 1)
    x = F.relu(_bn(F.conv2d(x, W[h + "conv2.weight"], stride=(2, 1), padding=1), W, h + "bn2."))
    x = x.reshape(1, x.shape[1] * x.shape[2], x.shape[3])          # [1, 320, T]
    xv = p + "xvector."
    x = F.conv1d(x, W[xv + "tdnn.linear.weight"], stride=2, padding=2)
    x = F.relu(_bn(x, W, xv + "tdnn.nonlinear.batchnorm."))
    for bi, (nl, dil) in enumerate(((12, 1), (24, 2), (16, 2))):
        for li in range(nl):
            q = xv + f"block{bi + 1}.tdnnd{li + 1}."
            y = F.conv1d(F.relu(_bn(x, W, q + "nonlinear1.batchnorm.")), W[q + "linear1.weight"])
            y = F.relu(_bn(y, W, q + "nonlinear2.batchnorm."))
            # CAMLayer.forward (layers.py:103-119)
            loc = F.conv1d(y, W[q + "cam_layer.linear_local.weight"], padding=dil, dilation=dil)
            seg = F.avg_pool1d(y, kernel_size=100, stride=100, ceil_mode=True)
            seg = seg.unsqueeze(-1).expand(*seg.shape, 100).reshape(*seg.shape[:-1], -1)[..., :y.shape[-1]]
            ctx = y.mean(-1, keepdim=True) + seg
            ctx = F.relu(F.conv1d(ctx, W[q + "cam_layer.linear1.weight"], W[q + "cam_layer.linear1.bias"]))
            mgate = torch.sigmoid(F.conv1d(ctx, W[q + "cam_layer.linear2.weight"], W[q + "cam_layer.linear2.bias"]))
            x = torch.cat([x, loc * mgate], dim=1)
        q = xv + f"transit{bi + 1}."
        x = F.conv1d(F.relu(_bn(x, W, q + "nonlinear.batchnorm.")), W[q + "linear.weight"])
    x = F.relu(_bn(x, W, xv + "out_nonlinear.batchnorm."))[..., :feat.shape[0] // 2]
    stats = torch.cat([x.mean(-1), x.std(-1, unbiased=True)], dim=-1)              # masked_statistics_pooling (layers.py:33-43)
    y = F.conv1d(stats.unsqueeze(-1), W[p + "dense.linear.weight"]).squeeze(-1)
    return _bn(y, W, p + "dense.nonlinear.batchnorm.", affine=False)[0]


def style_vector(wave16k: torch.Tensor, W: dict) -> torch.Tensor:
    """calculate_style_vec (evaluations/infer_arvc.py:179-211) for one utterance -> [1, 192]"""
    feat = kaldi_fbank(wave16k)
    feat = feat - feat.mean(dim=0, keepdim=True)
    return campplus(feat, W)[None]


# =========================================================================================
# SparkTTS speaker encoder (modules/bicodec_speaker_encoder)
# =========================================================================================
def _conv_relu_bn(x, W, p, **kw):
    """Conv1dReluBn (ecapa_tdnn.py:68-85): bn(relu(conv(x)))"""
    return _bn(F.relu(F.conv1d(x, W[p + "conv.weight"], W[p + "conv.bias"], **kw)), W, p + "bn.")


def _se_res2block(x, W, p, dil):
    """SE_Res2Block (ecapa_tdnn.py:111-147): scale 8, k 3, padding = dilation"""
    y = _conv_relu_bn(x, W, p + "0.")
    spx = torch.split(y, 64, 1)
    outs, sp = [], spx[0]
    for i in range(7):
        if i >= 1:
            sp = sp + spx[i]
        sp = F.conv1d(sp, W[p + f"1.convs.{i}.weight"], W[p + f"1.convs.{i}.bias"], padding=dil, dilation=dil)
        sp = _bn(F.relu(sp), W, p + f"1.bns.{i}.")
        outs.append(sp)
    outs.append(spx[7])
    y = _conv_relu_bn(torch.cat(outs, dim=1), W, p + "2.")
    s = y.mean(dim=2)
    s = torch.sigmoid(F.linear(F.relu(F.linear(s, W[p + "3.linear1.weight"], W[p + "3.linear1.bias"])), W[p + "3.linear2.weight"],
                               W[p + "3.linear2.bias"]))
    return x + y * s.unsqueeze(2)


def ecapa_latent(mel: torch.Tensor, W: dict, p: str = "timbre.speaker_encoder.") -> torch.Tensor:
    """ECAPA_TDNN.forward(..., return_latent=True)[1] (ecapa_tdnn.py:196-213): mel [1, 128, T] -> latent [1, 1536, T]"""
    o1 = _conv_relu_bn(mel, W, p + "layer1.", padding=2)
    o2 = _se_res2block(o1, W, p + "layer2.se_res2block.", 2)
    o3 = _se_res2block(o2, W, p + "layer3.se_res2block.", 3)
    o4 = _se_res2block(o3, W, p + "layer4.se_res2block.", 4)
    return F.relu(F.conv1d(torch.cat([o2, o3, o4], dim=1), W[p + "conv.weight"], W[p + "conv.bias"]))


def perceiver(x: torch.Tensor, n_valid: int, W: dict, p: str = "timbre.perceiver_sampler.") -> torch.Tensor:
    """PerceiverResampler.forward (perceiver_encoder.py:327-341) with the key mask of tokenize_wav (the 32 latents always, then
    the first n_valid context frames): x [1, T, 1536] -> [1, 32, 128]"""
    ctx = F.linear(x, W[p + "proj_context.weight"], W[p + "proj_context.bias"])
    lat = W[p + "latents"][None]
    T = ctx.shape[1]
    mask = torch.arange(32 + T) < (32 + n_valid)
    for l in range(2):
        q = p + f"layers.{l}."
        kv_in = torch.cat([lat, ctx], dim=1)                       # cross_attn_include_queries
        qq = F.linear(lat, W[q + "0.to_q.weight"]).view(1, 32, 8, 64).transpose(1, 2)
        kv = F.linear(kv_in, W[q + "0.to_kv.weight"])
        k, v = kv.chunk(2, dim=-1)
        k = k.view(1, -1, 8, 64).transpose(1, 2)
        v = v.view(1, -1, 8, 64).transpose(1, 2)
        sim = torch.einsum("bhid,bhjd->bhij", qq, k) * (64 ** -0.5)
        sim = sim.masked_fill(~mask[None, None, None, :], -torch.finfo(sim.dtype).max)
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v).transpose(1, 2).reshape(1, 32, 512)
        lat = F.linear(out, W[q + "0.to_out.weight"]) + lat
        h = F.linear(lat, W[q + "1.0.weight"], W[q + "1.0.bias"])
        a, gate = h.chunk(2, dim=-1)
        lat = F.linear(F.gelu(gate) * a, W[q + "1.2.weight"], W[q + "1.2.bias"]) + lat
    return F.normalize(lat, dim=-1) * (128 ** 0.5) * W[p + "norm.gamma"]


def fsq4_quantize(x: torch.Tensor, W: dict, p: str = "timbre.quantizer.") -> torch.Tensor:
    """ResidualFSQ(levels=[4]*6, num_quantizers=1) forward (residual_fsq.py:160-230 -> FSQ.quantize,
    finite_scalar_quantization.py:126-139): x [1, 32, 128] (channel-last) -> quantized_out [1, 32, 128]"""
    z = F.linear(x, W[p + "project_in.weight"], W[p + "project_in.bias"])
    half_l = (4 - 1) * (1 + 1e-3) / 2
    shift = math.atanh(0.5 / half_l)
    q = torch.round(torch.tanh(z + shift) * half_l - 0.5) / 2.0
    return F.linear(q, W[p + "project_out.weight"], W[p + "project_out.bias"])


def timbre_latents(wave16k: torch.Tensor, W: dict) -> torch.Tensor:
    """calculate_timbre_latent (evaluations/infer_arvc.py:213-223) for one utterance -> [1, 32, 128]"""
    mel = mel_spectrogram_16k(wave16k)[None]                       # [1, 128, frames]
    lat = ecapa_latent(mel, W)
    n_valid = wave16k.reshape(-1).shape[0] // 320                  # mel_lens = wav_lens // hop_length
    x = perceiver(lat.transpose(1, 2), n_valid, W)
    return fsq4_quantize(x, W)                                     # zq [1, 128, 32].mT == channel-last [1, 32, 128]
