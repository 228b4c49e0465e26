"""Container-only harness that imports the *reference* (read-only, /root/reference)
on CPU so that golden vectors can be captured from it.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (streamvoiceanon_amd/) imports this
file, and it cannot run on the GPU box (the reference does not travel).  It exists so
that tools/make_golden.py can generate tests/golden/*.npz.

What it does:
  * injects stub modules for the reference's missing third-party imports
    (torchaudio, librosa, hydra, omegaconf, einx, vector_quantize_pytorch);
  * replaces torch.cuda.Event / synchronize with host no-ops (the reference calls them
    unconditionally, evaluations/infer_arvc.py:498-512);
  * builds an InferenceWrapper via __new__ with fp32 KV caches (reference creates fp16
    caches which cannot run on CPU, evaluations/infer_arvc.py:55-59);
  * loads deterministic synthetic weights produced by streamvoiceanon_amd.synth_weights.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import time
import types

import numpy as np
import torch
import yaml

REF_ROOT = os.environ.get("SVA_REFERENCE_ROOT", "/root/reference")


# ----------------------------------------------------------------------------------------
# third-party restatements needed to import the reference
# ----------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if f >= min_log_hz:
        return min_log_mel + math.log(f / min_log_hz) / logstep
    return f / f_sp


def _mel_to_hz_slaney(m: torch.Tensor) -> torch.Tensor:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    f = f_sp * m
    log_t = m >= min_log_mel
    f[log_t] = min_log_hz * torch.exp(logstep * (m[log_t] - min_log_mel))
    return f


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """Published algorithm of torchaudio.functional.melscale_fbanks (torchaudio==2.4.0,
    requirements.txt:7; call site modules/vqgan/spectrogram.py:93-101).  torchaudio is not
    installed in this image, so the algorithm is restated here (slaney scale only)."""
    assert mel_scale == "slaney" and norm == "slaney"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_slaney(f_min)
    m_max = _hz_to_mel_slaney(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(zero, torch.min(down, up))
    enorm = 2.0 / (f_pts[2: n_mels + 2] - f_pts[:n_mels])
    fb = fb * enorm.unsqueeze(0)
    return fb


def _einx_get_at(pattern, codebooks, indices):
    # only pattern used: "q [c] d, b n q -> q b n d"  (residual_fsq.py:136)
    assert pattern.replace(" ", "") == "q[c]d,bnq->qbnd"
    q = codebooks.shape[0]
    outs = [codebooks[i][indices[..., i]] for i in range(q)]
    return torch.stack(outs, dim=0)


def _instantiate(cfg, **kwargs):
    """Minimal hydra.utils.instantiate: recursive `_target_` construction."""
    if isinstance(cfg, dict):
        if "_target_" in cfg:
            target = cfg["_target_"]
            mod_name, cls_name = target.rsplit(".", 1)
            cls = getattr(importlib.import_module(mod_name), cls_name)
            args = {k: _instantiate(v) for k, v in cfg.items() if k != "_target_"}
            args.update(kwargs)
            return cls(**args)
        return {k: _instantiate(v) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [_instantiate(v) for v in cfg]
    return cfg


class _HostEvent:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")
    taf.melscale_fbanks = melscale_fbanks

    def _no_resample(x, orig_freq=None, new_freq=None):
        raise RuntimeError("resample is on the prompt path (N1), stubbed out in the harness")

    taf.resample = _no_resample
    tat = types.ModuleType("torchaudio.transforms")

    class MelScale(torch.nn.Module):
        pass

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            self.hop_length = kw.get("hop_length", 320)

    tat.MelScale = MelScale
    tat.MelSpectrogram = MelSpectrogram
    tac = types.ModuleType("torchaudio.compliance")
    tack = types.ModuleType("torchaudio.compliance.kaldi")
    ta.functional, ta.transforms, ta.compliance = taf, tat, tac
    tac.kaldi = tack
    ta.save = lambda *a, **k: None
    for name, mod in [("torchaudio", ta), ("torchaudio.functional", taf), ("torchaudio.transforms", tat),
                      ("torchaudio.compliance", tac), ("torchaudio.compliance.kaldi", tack)]:
        sys.modules[name] = mod

    sys.modules["librosa"] = types.ModuleType("librosa")

    hydra = types.ModuleType("hydra")
    hutils = types.ModuleType("hydra.utils")
    hutils.instantiate = _instantiate
    hydra.utils = hutils
    sys.modules["hydra"] = hydra
    sys.modules["hydra.utils"] = hutils

    oc = types.ModuleType("omegaconf")

    class OmegaConf:
        @staticmethod
        def load(path):
            return yaml.safe_load(open(path))

    oc.OmegaConf = OmegaConf
    oc.DictConfig = lambda d: d
    sys.modules["omegaconf"] = oc

    einx = types.ModuleType("einx")
    einx.get_at = _einx_get_at
    sys.modules["einx"] = einx

    # vector_quantize_pytorch==1.14.24 (requirements.txt:26) is not installed; the reference
    # vendors the same lucidrains code under modules/bicodec_speaker_encoder/fsq/.
    rf = importlib.import_module("modules.bicodec_speaker_encoder.fsq.residual_fsq")
    rf.ceil = math.ceil
    vqp = types.ModuleType("vector_quantize_pytorch")
    vqp.GroupedResidualFSQ = rf.GroupedResidualFSQ
    sys.modules["vector_quantize_pytorch"] = vqp

    torch.cuda.Event = _HostEvent
    torch.cuda.synchronize = lambda *a, **k: None


def load_yaml(rel):
    return yaml.safe_load(open(os.path.join(REF_ROOT, rel)))


def build_models():
    """Instantiate the three hot-path networks from the reference's own YAMLs."""
    install_stubs()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        top = load_yaml("configs/config_firefly_arvcasr_8192_delay0_8.yaml")
        torch.manual_seed(0)
        model = _instantiate(load_yaml(top["model_params"]["config_path"]))
        tok = _instantiate(load_yaml(top["speech_tokenizer"]["config_path"]))
        voc = _instantiate(load_yaml(top["firefly"]["config_path"]))
    finally:
        os.chdir(cwd)
    return model.eval(), tok.eval(), voc.eval()


def load_synth(model, tok, voc, seed):
    from streamvoiceanon_amd import synth_weights as sw

    voc.remove_parametrizations()   # evaluations/infer_arvc.py:94 (weights become plain tensors)
    for name, net, prefix in (("arvc", model, "arvc"), ("tokenizer", tok, "tok"), ("vocoder", voc, "voc")):
        sd = net.state_dict()
        new = {}
        for k, v in sd.items():
            arr = sw.generate(seed, prefix + "." + k, tuple(v.shape))
            if arr is None:       # not a hot-path tensor: leave reference init
                continue
            new[k] = torch.from_numpy(arr).to(v.dtype)
        missing, unexpected = net.load_state_dict(new, strict=False)
        assert not unexpected, unexpected


def build_wrapper(seed=0, style=None, timbre=None):
    """InferenceWrapper via __new__ (the reference __init__ needs checkpoints + CUDA)."""
    install_stubs()
    from evaluations.infer_arvc import InferenceWrapper

    model, tok, voc = build_models()
    load_synth(model, tok, voc, seed)
    model.setup_caches(max_batch_size=1, max_seq_len=2048, dtype=torch.float32)
    w = InferenceWrapper.__new__(InferenceWrapper)
    w.device = torch.device("cpu")
    w.sr = 44100
    w.model, w.speech_tokenizer, w.firefly = model, tok, voc
    w.compiled_speech_tokenizer_encode = tok.encode
    g = torch.Generator().manual_seed(1234 + seed)
    w._style = style if style is not None else torch.randn(1, 192, generator=g)
    w._timbre = timbre if timbre is not None else torch.randn(1, 32, 128, generator=g)
    w.calculate_style_vec = lambda a, l: w._style.clone()
    w.calculate_timbre_latent = lambda a, l: w._timbre.clone()
    # resample stub for calculate_prompt (16 kHz audio only feeds the stubbed encoders)
    sys.modules["torchaudio"].functional.resample = lambda x, orig_freq=None, new_freq=None: x[..., ::3]
    return w
