"""Host-side mirror of the reference's ``ARVCWrapper`` (modules/arvc_wrapper.py:7-126) on top of the
HIP engine.  Same method names, argument meaning and return conventions, so code written against the
reference's wrapper (evaluations/infer_arvc.py:484-489, 523, 535-537) keeps working; the arithmetic
runs in libsva_hip.so.  torch tensors in / out (CPU or CUDA), batch 1 like the reference.
"""
from __future__ import annotations

import numpy as np

from . import engine as E


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x.astype(dtype) if dtype is not None else x


class ARVCWrapper:
    def __init__(self, engine: E.Engine, delay: int = 2, max_seq_frames: int = 768, buffer_frames: int = 32,
                 temperature: float = 0.7, top_p: float = 0.7, noise_seed: int = 0, **batch_kwargs):
        self.engine = engine
        self.delay = int(delay)                # DualARWrapper.delay
        self.original_delay = list(range(engine.cfg.max_delay + 1))    # dynamic-delay model (config delay list)
        self._kw = dict(max_seq_frames=max_seq_frames, buffer_frames=buffer_frames, temperature=temperature, top_p=top_p,
                        **batch_kwargs)
        self.noise_seed = noise_seed
        self.batch = None
        self.compiled_fn = None

    # modules/arvc_wrapper.py:25-47 ------------------------------------------------------------------------
    def compile_ar_decode_fn(self):
        """The reference wraps decode_one_token_ar in torch.compile(mode="reduce-overhead"); the engine's
        analogue is the captured hipGraph of the steady-state step."""
        self._kw["use_graph"] = True

    def setup_caches(self, max_batch_size: int = 1, max_seq_len: int = 2048, dtype=None):
        assert max_batch_size == 1 and max_seq_len <= self.engine.cfg.max_seq_len
        # KV caches are allocated with the batch (fp32 in parity mode); nothing else to do

    def set_delay(self, delay: int):
        self.delay = int(delay)
        print(f"Setting delay to {self.delay} frames")

    # modules/arvc_wrapper.py:82-98 ------------------------------------------------------------------------
    def generate(self, ref_content_codes, ref_audio_codes, src_content_codes, style_vectors, timbre_latents, noise=None,
                 **sampling_kwargs):
        """Offline conversion -> int32 codes [1, 8, S] (the reference returns pred_audio_codes.transpose(0, 1))."""
        import torch

        kw = dict(self._kw)
        from .infer_arvc import check_sampling_kwargs

        kw.update(check_sampling_kwargs(sampling_kwargs))
        edits = kw.pop("edits", None)
        batch = E.Batch(self.engine, n_streams=1, delay=self.delay, **kw)
        try:
            if edits:
                batch.set_sampler_edits(**edits)       # applied from the second frame on, like the reference's loop (dual_ar_stream.py:722 vs :749-754)
            codes = batch.generate(_np(ref_content_codes, np.int64), _np(ref_audio_codes, np.int32), _np(src_content_codes, np.int64),
                                   _np(style_vectors, np.float32), _np(timbre_latents, np.float32).reshape(32, -1),
                                   noise_seed=self.noise_seed, noise=noise)
        finally:
            batch.close()
        return torch.from_numpy(codes)[None]

    # modules/arvc_wrapper.py:100-126 ----------------------------------------------------------------------
    def prefill_prompt(self, ref_content_codes, ref_audio_codes, style_vectors, timbre_latents):
        """ref_content_codes [1, R] int64, ref_audio_codes [1, 8, R] int, style [1, 192], timbre [1, 32, 128]."""
        if self.batch is not None:
            self.batch.close()
        self.batch = E.Batch(self.engine, n_streams=1, delay=self.delay, **self._kw)
        cc = _np(ref_content_codes, np.int64).reshape(-1)
        ac = _np(ref_audio_codes, np.int32).reshape(8, -1)
        self.batch.prefill_prompt(0, cc, ac, _np(style_vectors, np.float32).reshape(-1), _np(timbre_latents, np.float32).reshape(32, -1),
                                  noise_seed=self.noise_seed)
        self.batch.begin()

    def prefill_src_condition4delay(self, src_content_codes):
        """src_content_codes [1, delay] int64 (dual_ar_stream.py:798-815)."""
        self.batch.ar_delay_fill(_np(src_content_codes, np.int64).reshape(1, -1))

    def decode_one(self, src_content_codes, noise=None):
        """src_content_codes [1, 1] -> (codes int32 [8, 1], kv_pos[-1]) (dual_ar_stream.py:817-837).
        `noise` optionally supplies the Exp(1) draws [8192 + 8*1000]; default = on-device counter RNG."""
        import torch

        codes, pos = self.batch.ar_decode_one(_np(src_content_codes, np.int64).reshape(1), noise=noise)
        return torch.from_numpy(codes[0].reshape(8, 1).copy()), int(pos[0])
