"""Host-side mirror of ``InferenceWrapper`` (evaluations/infer_arvc.py:26-689) on top of the HIP engine:
``infer`` / ``stream_infer`` / ``prefill_prompt`` / ``setup_stream_caches`` / ``process_one_chunk`` / ``calculate_prompt`` with the
reference's names, defaults and quirks, the module seams the hot loop crosses (``.speech_tokenizer.encode``,
``.model.decode_one``, ``.firefly.quantizer.decode``, ``.firefly.head``, :506-508, 535-537, 175) as attributes, and the
reference's command line (``python -m streamvoiceanon_amd.infer_arvc --src_path ... --ref_path ... [--simulate_streaming]``,
:691-743) -- so callers such as the CLI or the GUI's ``custom_infer`` (real-time-gui.py:32-49) switch by changing one import.

The prompt's two code streams (``firefly.encode`` audio codes, speech-tokenizer content codes) are computed on the device.  The
two speaker-embedding encoders (CAM++ style vector, SparkTTS timbre latents; SURVEY.md 8f N1 iii/iv) run on the device when
their weights are loaded (``style.*`` / ``timbre.*`` tensors); otherwise ``calculate_prompt`` takes ``style_vectors=`` /
``timbre_latents=`` (or installable callables) and raises NotImplementedError naming the row.
"""
from __future__ import annotations

import os

import numpy as np

from . import engine as E


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x.astype(dtype) if dtype is not None else x


def _like(ref, arr):
    """numpy result -> torch tensor on ref's device when the caller passed torch (the reference's seams are torch in / torch out)"""
    if hasattr(ref, "detach"):
        import torch

        return torch.from_numpy(np.ascontiguousarray(arr)).to(ref.device)
    return arr


def check_sampling_kwargs(sampling_kwargs: dict) -> dict:
    """`decode_one_token_ar(..., previous_tokens=None, suppress_tokens=None, **sampling_kwargs)` -> `sample` -> `logits_to_probs(
    logits, previous_tokens, suppress_tokens, temperature=0.7, top_p=0.7, repetition_penalty=1.5)` (modules/dual_ar_stream.py:
    1081-1132, 1175-1213).  temperature / top_p become batch parameters; previous_tokens / suppress_tokens / repetition_penalty
    are returned under "edits" (-> `Batch.set_sampler_edits`; the penalty only acts through previous_tokens, as in the reference).
    Unknown names raise TypeError like the reference's signature would."""
    out, edits = {}, {}
    for k, v in sampling_kwargs.items():
        if k in ("temperature", "top_p"):
            out[k] = float(v)
        elif k == "repetition_penalty":
            edits[k] = float(v)
        elif k in ("previous_tokens", "suppress_tokens"):
            if v is not None:
                edits[k] = v
        else:
            raise TypeError(f"logits_to_probs() got an unexpected keyword argument '{k}'")
    if "previous_tokens" in edits or "suppress_tokens" in edits:
        out["edits"] = edits
    return out


class _SpeechTokenizerSeam:
    """`speech_tokenizer.encode(audios, audio_lengths)` (modules/vqgan/modules/firefly_encoder.py:553-566):
    float [B, N] (+ lengths [B]) -> (codes int64 [1, B, N // 2048], lengths // 2048)."""

    def __init__(self, wrapper):
        self._w = wrapper

    def encode(self, audios, audio_lengths=None):
        x = _np(audios, np.float32)
        x = x.reshape(1, -1) if x.ndim == 1 else x.reshape(x.shape[0], -1)
        B, N = x.shape
        T = N // 2048
        lens = np.full(B, N, np.int64) if audio_lengths is None else _np(audio_lengths, np.int64).reshape(B)
        codes = np.zeros((1, B, T), np.int64)
        for i in range(B):
            # the reference masks the mel frames beyond a row's length (:558-559) = encoding the zero-padded row (causal net)
            row = x[i].copy()
            row[int(lens[i]):] = 0.0
            codes[0, i] = self._w.encode_content(row[:T * 2048])
        return _like(audios, codes), _like(audios, lens // 2048)

    __call__ = encode


class _QuantizerSeam:
    def __init__(self, wrapper):
        self._w = wrapper

    def decode(self, codes):
        """firefly.quantizer.decode (modules/vqgan/modules/fsq.py:112-116): int [B, 8, T] -> float32 [B, 512, 4T]."""
        c = _np(codes, np.int32)
        c = c.reshape(-1, 8, c.shape[-1])
        b = E.Batch(self._w.engine, n_streams=c.shape[0], voc_max_frames=c.shape[2])
        try:
            return _like(codes, b.quantizer_decode(c))
        finally:
            b.close()


class _FireflySeam:
    """`firefly.quantizer.decode` / `firefly.head` (code2wav_fn, evaluations/infer_arvc.py:173-176) and `firefly.encode`
    (wav2target_fn, :168-171)."""

    def __init__(self, wrapper):
        self._w = wrapper
        self.quantizer = _QuantizerSeam(wrapper)

    def head(self, z):
        """HiFiGANGenerator.forward (modules/vqgan/modules/firefly.py:280-293): float [B, 512, 4T] -> [B, 1, 2048 T]."""
        zz = _np(z, np.float32)
        zz = zz.reshape(-1, zz.shape[-2], zz.shape[-1])
        b = E.Batch(self._w.engine, n_streams=zz.shape[0], voc_max_frames=zz.shape[2] // 4)
        try:
            return _like(z, b.vocoder_head(zz))
        finally:
            b.close()

    def encode(self, audios, audio_lengths=None):
        """FireflyArchitecture.encode (firefly.py:560-574) -> ((indices int [B, 8, T], None), feature lengths); the quantised
        latent the reference returns beside the indices is not produced (no caller on the path reads it, :168-171, 431-434)."""
        x = _np(audios, np.float32)
        x = x.reshape(1, -1) if x.ndim == 1 else x.reshape(x.shape[0], -1)
        lens = np.full(x.shape[0], x.shape[1], np.int64) if audio_lengths is None else _np(audio_lengths, np.int64).reshape(-1)
        out = np.concatenate([self._w.wav2target_fn(np.where(np.arange(x.shape[1]) < lens[i], x[i], 0.0)) for i in range(x.shape[0])])
        return (_like(audios, out), None), _like(audios, lens // 2048)

    def remove_parametrizations(self):
        return self          # weight-norm pairs are folded when the engine packs its weights


class InferenceWrapper:
    SAMPLES_PER_FRAME = 2048       # evaluations/infer_arvc.py:28
    NUM_CODEBOOKS = 8
    RESAMPLE_FREQ = 16000
    MEL_BINS = 80

    def __init__(self, config_path=None, checkpoint_path=None, compile_encoder=False, compile_decoder=False, compile_ar=False,
                 fp16=False, weights: dict | None = None, device: int = 0):
        """Same signature as the reference (:33) plus ``weights``: a dict of state-dict tensors keyed
        'arvc.*' / 'tok.*' / 'voc.*' (real checkpoints are loaded with load_checkpoints())."""
        if weights is None:
            weights = self.load_checkpoints(config_path, checkpoint_path)
        self.sr = 44100
        self.device = f"cuda:{device}"
        # fp16: the reference's `self.model.half()` (:62-63) -> fp16 AR weights + fp16 KV cache (sva_config.ar_dtype = 1)
        self.engine = E.Engine(weights, device=device, ar_dtype=1 if fp16 else 0)
        self.use_graph = bool(compile_ar or compile_decoder or compile_encoder)   # the reference's --compile
        self.batch = None
        self._win_batch = None            # (Wp, Batch) of the last whole-utterance encode: see _window_batch
        self._prompt = None
        # the module seams the reference's hot loop crosses (:506-508, 535-537, 175)
        from .arvc_wrapper import ARVCWrapper

        self.model = ARVCWrapper(self.engine)
        self.speech_tokenizer = _SpeechTokenizerSeam(self)
        self.firefly = _FireflySeam(self)
        # the two speaker-embedding encoders of the prompt path (:96-125, 179-223) run on the device when their weights are given
        # ("style.*" = CAM++, "timbre.*" = SparkTTS SpeakerEncoder); a caller may also install any callable wav16k -> embedding
        self.style_encoder = self.timbre_encoder = None
        if any(k.startswith("style.") for k in weights):
            from .prompt_encoders import StyleEncoder

            self.style_encoder = StyleEncoder(self.engine, weights)
        if any(k.startswith("timbre.") for k in weights):
            from .prompt_encoders import TimbreEncoder

            self.timbre_encoder = TimbreEncoder(self.engine, weights)

    def calculate_style_vec(self, audio_16k_tensor, wave_lens=None):
        """:179-211 Kaldi fbank (80 bins, mean-subtracted) -> CAM++ -> [1, 192] (batch 1 like every call site)"""
        if self.style_encoder is None:
            raise NotImplementedError("CAM++ style encoder weights ('style.*', SURVEY.md 8f N1 iii) were not loaded: pass style_vectors= "
                                      "or set InferenceWrapper.style_encoder to a callable wav16k -> [1, 192]")
        x = _np(audio_16k_tensor, np.float32).reshape(-1)
        if wave_lens is not None:
            x = x[:int(_np(wave_lens).reshape(-1)[0])]
        return _like(audio_16k_tensor, np.asarray(self.style_encoder(x), np.float32).reshape(1, -1))

    def calculate_timbre_latent(self, audio_16k_tensor, wave_lens=None):
        """:213-223 SpeakerEncoder.tokenize_wav -> zq.mT [1, 32, 128]"""
        if self.timbre_encoder is None:
            raise NotImplementedError("SparkTTS timbre encoder weights ('timbre.*', SURVEY.md 8f N1 iv) were not loaded: pass timbre_latents= "
                                      "or set InferenceWrapper.timbre_encoder to a callable wav16k -> [1, 32, 128]")
        x = _np(audio_16k_tensor, np.float32).reshape(-1)
        if wave_lens is not None:
            x = x[:int(_np(wave_lens).reshape(-1)[0])]
        return _like(audio_16k_tensor, np.asarray(self.timbre_encoder(x), np.float32).reshape(1, 32, -1))

    def code2wav_fn(self, code):
        """:173-176 firefly.head(firefly.quantizer.decode(code))"""
        return self.firefly.head(self.firefly.quantizer.decode(code))

    @staticmethod
    def load_checkpoints(config_path, checkpoint_path):
        """Reads the five `.pth` files named by the reference YAML (config_firefly_arvcasr_8192_delay0_8.yaml:43-57)
        into the prefixed key space, unwrapping 'net' / 'module.' like infer_arvc.py:70-78 does."""
        import torch
        import yaml

        cfg = yaml.safe_load(open(config_path))
        out = {}
        sd = torch.load(checkpoint_path, map_location="cpu")
        out.update({"arvc." + k: v for k, v in sd.items()})
        tok = torch.load(cfg["speech_tokenizer"]["checkpoint_path"], map_location="cpu")
        tok = tok.get("net", tok)
        out.update({"tok." + (k[7:] if k.startswith("module.") else k): v for k, v in tok.items()})
        voc = torch.load(cfg["firefly"]["checkpoint_path"], map_location="cpu")
        out.update({"voc." + k: v for k, v in voc.items()})     # weight-norm pairs are folded by the engine
        # speaker-embedding encoders (:96-125); CAM++ checkpoints of an older layout keep `stats` / `dense` under `xvector.`
        # (modules/campplus/DTDNN.py:107-124)
        for net, prefix in (("style_encoder", "style."), ("timbre_encoder", "timbre.")):
            path = (cfg.get(net) or {}).get("checkpoint_path")
            if path and os.path.exists(path):
                sd = torch.load(path, map_location="cpu")
                for k, v in sd.items():
                    if k.startswith("xvector.stats"):
                        k = k.replace("xvector.stats", "stats")
                    elif k.startswith("xvector.dense"):
                        k = k.replace("xvector.dense", "dense")
                    out[prefix + k] = v
        return {k: v for k, v in out.items() if hasattr(v, "dtype") and v.dtype.is_floating_point}

    # ---- prompt ------------------------------------------------------------------------------------------
    def _window_batch(self, Wp):
        """The batch behind the whole-utterance seams (`firefly.encode`, `speech_tokenizer.encode`): creating one costs 3-4 ms of
        allocations, which was half of either call, and calculate_prompt makes both on the same window length -- the last one is
        kept (closed when another length is asked for, or by close())."""
        if self._win_batch is not None and self._win_batch[0] == Wp:
            return self._win_batch[1]
        if self._win_batch is not None:
            self._win_batch[1].close()
            self._win_batch = None
        b = E.Batch(self.engine, n_streams=1, encode_window_frames=Wp)
        self._win_batch = (Wp, b)
        return b

    def close(self):
        """Release the batches this wrapper holds (the engine and its weights stay with `self.engine`)."""
        if self._win_batch is not None:
            self._win_batch[1].close()
            self._win_batch = None
        if self.batch is not None:
            self.batch.close()
            self.batch = None

    def wav2target_fn(self, waves):
        """:168-171 firefly.encode of a whole prompt -> acoustic codes int32 [1, 8, R], R = len // 2048 (right-padded with
        zeros to a multiple of 4 frames for the stride-4 front-end; causal, so the first R columns are unaffected)."""
        wav = np.asarray(waves.detach().cpu().numpy() if hasattr(waves, "detach") else waves, dtype=np.float32).reshape(-1)
        R = wav.shape[0] // self.SAMPLES_PER_FRAME
        Wp = ((R + 3) // 4) * 4
        buf = np.zeros(Wp * self.SAMPLES_PER_FRAME, np.float32)
        buf[:R * self.SAMPLES_PER_FRAME] = wav[:R * self.SAMPLES_PER_FRAME]
        return self._window_batch(Wp).firefly_encode(buf[None])[:, :, :R]

    def calculate_prompt(self, ref_wav_tensors, alpha=1.0, spk_emb_collate_type="concat_mel", style_vectors=None,
                         timbre_latents=None):
        """:382-441.  The two code streams of the prompt (firefly.encode audio codes, speech-tokenizer content codes) are
        computed on the device, and so are the CAM++ style vector and the SparkTTS timbre latents (prompt_encoders.py, SURVEY 8f N1
        iii / iv) when the engine holds the `style.*` / `timbre.*` weights: `self.style_encoder(wav)` / `self.timbre_encoder(wav)`.
        The `style_vectors` / `timbre_latents` arguments (or caller-installed encoder callables) override them; without weights and
        without overrides the call raises.  Alpha noise mixing (:426-427) is applied to the embeddings here."""
        import torch

        ref_list = ref_wav_tensors if isinstance(ref_wav_tensors, (list, tuple)) else [ref_wav_tensors]
        ref = np.concatenate([np.asarray(r.detach().cpu().numpy() if hasattr(r, "detach") else r, dtype=np.float32).reshape(-1)
                              for r in ref_list])            # :411 / :415 torch.cat(ref_wav_list, dim=-1)
        if spk_emb_collate_type == "avg" and len(ref_list) > 1:
            # reference quirk (vi): THIS function's 'avg' branch falls through to an undefined variable (:389-424) -- only 'concat_mel' works
            # for the streaming prompt upstream.  The offline infer() has its own, working 'avg' branch (:284-307): see infer / _avg_embeddings
            raise NotImplementedError("spk_emb_collate_type='avg' with several references raises NameError in the reference's streaming "
                                      "calculate_prompt (evaluations/infer_arvc.py:389-424); use 'concat_mel' (offline infer() supports 'avg')")
        if style_vectors is None or timbre_latents is None:
            from . import audio_io

            ref16 = audio_io.resample(ref, self.sr, self.RESAMPLE_FREQ)                    # :415-417
            # the device encoders replay one captured graph each on the engine's ops stream (prompt_encoders._Plan): enqueue both, compute the
            # two code streams on the batch streams meanwhile, fetch the embeddings last -- four independent functions of the same audio
            se, te = self.style_encoder, self.timbre_encoder
            split = style_vectors is None and timbre_latents is None and all(hasattr(x, "prepare") for x in (se, te))
            if split:
                hs, ht = se.prepare(ref16), te.prepare(ref16)
                se.launch(hs); te.launch(ht)
                ref_audio_codes = self.wav2target_fn(ref)
                ref_content_codes = self.encode_content(ref)
                style_vectors = np.asarray(se.finish(hs), np.float32).reshape(1, -1)
                timbre_latents = np.asarray(te.finish(ht), np.float32).reshape(1, 32, -1)
            else:
                if style_vectors is None:
                    style_vectors = self.calculate_style_vec(ref16)
                if timbre_latents is None:
                    timbre_latents = self.calculate_timbre_latent(ref16)
        else:
            split = False
        style_vectors = self.apply_noise_mixing(torch.as_tensor(np.asarray(style_vectors), dtype=torch.float32), alpha)
        timbre_latents = self.apply_noise_mixing(torch.as_tensor(np.asarray(timbre_latents), dtype=torch.float32), alpha)
        if not split:
            ref_audio_codes = self.wav2target_fn(ref)                       # :431-434
            ref_content_codes = self.encode_content(ref)                    # :436-439
        return ref_audio_codes, ref_content_codes, style_vectors, timbre_latents, ref

    def _avg_embeddings(self, refs, style_vectors=None, timbre_latents=None):
        """infer's 'avg' collation (:284-303): per reference resample to 16 kHz -> calculate_style_vec / calculate_timbre_latent,
        then torch.mean over the stack (fp32, reference order)."""
        import torch
        from . import audio_io

        sv, tl = [], []
        for w in refs:
            w16 = audio_io.resample(np.asarray(w, np.float32).reshape(-1), self.sr, self.RESAMPLE_FREQ)
            if style_vectors is None:
                sv.append(torch.as_tensor(np.asarray(self.calculate_style_vec(w16), np.float32)).reshape(1, -1))
            if timbre_latents is None:
                tl.append(torch.as_tensor(np.asarray(self.calculate_timbre_latent(w16), np.float32)).reshape(1, 32, -1))
        if style_vectors is None:
            style_vectors = torch.mean(torch.stack(sv, dim=0), dim=0).numpy()
        if timbre_latents is None:
            timbre_latents = torch.mean(torch.stack(tl, dim=0), dim=0).numpy()
        return style_vectors, timbre_latents

    def apply_noise_mixing(self, tensor, alpha, gauss=None):
        """:228-232 -- alpha*x + (1-alpha)*(randn*std + mean), global mean / unbiased std."""
        import torch

        mean, std = tensor.mean(), tensor.std()
        noise = (torch.randn_like(tensor) if gauss is None else gauss) * std + mean
        return alpha * tensor + (1 - alpha) * noise

    def prefill_prompt(self, ref_wav_tensors=None, max_prompt_frames=256, delay=4, alpha=1.0, spk_emb_collate_type="concat_mel",
                       prompt=None, noise_seed=0):
        if prompt is None:
            prompt = self.calculate_prompt(ref_wav_tensors, alpha=alpha, spk_emb_collate_type=spk_emb_collate_type)
        ref_audio_codes, ref_content_codes, style_vectors, timbre_latents = prompt[:4]
        self._prompt = tuple(np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x) for x in
                             (ref_audio_codes, ref_content_codes, style_vectors, timbre_latents))
        self.max_prompt_frames = max_prompt_frames
        self.delay = int(delay)
        self._noise_seed = noise_seed
        print(f"Setting delay to {self.delay} frames")

    def setup_stream_caches(self, encode_window_frames=96, decode_window_frames=64, max_seq_frames=768, buffer_frames=32,
                            decode_chunk_frames=1, delay=None, pipeline=False):
        assert self._prompt is not None, "call prefill_prompt first (as stream_infer does, :631-645)"
        if delay is not None:
            self.delay = int(delay)
        if self.batch is not None:
            self.batch.close()
        self.decode_chunk_frames = decode_chunk_frames
        self.batch = E.Batch(self.engine, n_streams=1, encode_window_frames=encode_window_frames,
                             decode_window_frames=decode_window_frames, chunk_frames=decode_chunk_frames, delay=self.delay,
                             max_seq_frames=max_seq_frames, buffer_frames=buffer_frames, max_prompt_frames=self.max_prompt_frames,
                             use_graph=self.use_graph, pipeline=pipeline and not self.use_graph)
        ac, cc, st, tm = self._prompt
        self.batch.prefill_prompt(0, cc.reshape(-1), ac.reshape(8, -1), st.reshape(-1), tm.reshape(32, -1), noise_seed=self._noise_seed)
        self.batch.begin()

    # ---- files ---------------------------------------------------------------------------
    def _load_src(self, src):
        """librosa.load(src_path, sr=self.sr) (:274, 615) when given a path; arrays pass through."""
        if isinstance(src, (str, os.PathLike)):
            from . import audio_io

            return audio_io.load(os.fspath(src), self.sr)[0], os.fspath(src)
        return np.asarray(src, dtype=np.float32).reshape(-1), None

    def load_and_crop_references(self, ref_paths, crop_lengths):
        """:250-260 -- paths are loaded at self.sr and cropped to crop_len seconds; arrays pass through the same crop."""
        from . import audio_io

        out = []
        for ref_p, crop_len in zip(ref_paths, crop_lengths):
            w = audio_io.load(os.fspath(ref_p), self.sr)[0] if isinstance(ref_p, (str, os.PathLike)) else np.asarray(ref_p, np.float32).reshape(-1)
            if crop_len is not None:
                w = w[:int(crop_len * self.sr)]
            out.append(w)
        return out

    def process_ref_paths(self, ref_path, ref_crop_lengths=None):
        """:234-248 -- one reference or a list; one crop length for all or one per reference."""
        ref_paths = list(ref_path) if isinstance(ref_path, (list, tuple)) else [ref_path]
        if ref_crop_lengths is None or not isinstance(ref_crop_lengths, (list, tuple)):
            crop = [ref_crop_lengths] * len(ref_paths)
        else:
            assert len(ref_crop_lengths) == len(ref_paths)
            crop = list(ref_crop_lengths)
        return ref_paths, crop

    def _save(self, pred_wave, src_path, ref_path, out_dir, output_path=None):
        """:363-379 / :676-688 output naming + torchaudio.save (32-bit float WAVE)."""
        from . import audio_io

        src_name = os.path.splitext(os.path.basename(src_path))[0] if src_path else "src"
        refs = ref_path if isinstance(ref_path, (list, tuple)) else [ref_path]
        ref_name = "_".join(os.path.splitext(os.path.basename(os.fspath(r)))[0] if isinstance(r, (str, os.PathLike)) else "ref" for r in refs)
        out_path = output_path or os.path.join(out_dir or (os.path.dirname(src_path) if src_path else "."), f"{src_name}_{ref_name}.wav")
        audio_io.write_wav(out_path, pred_wave, self.sr)
        print(f"Output saved to {out_path}")
        return out_path

    # ---- offline -----------------------------------------------------------------------------------------
    def encode_content(self, wav):
        """speech_tokenizer.encode on a whole utterance (:334-339) -> int64 codes [S], S = len // 2048.  The utterance is
        right-padded with zeros to a multiple of 4 frames (causal encoder: earlier codes are unaffected).  Beyond 256 frames the
        transformer runs the tiled attention kernel with the 512-token causal window of WindowLimitedTransformer
        (modules/vqgan/windowed_transformer.py:291-304)."""
        wav = np.asarray(wav, dtype=np.float32).reshape(-1)
        S = wav.shape[0] // self.SAMPLES_PER_FRAME
        Wp = ((S + 3) // 4) * 4
        if Wp > 2048:
            raise ValueError(f"utterance of {S} frames: the tokenizer's transformer has rotary tables for 2048 positions (95 s)")
        buf = np.zeros(Wp * self.SAMPLES_PER_FRAME, np.float32)
        buf[:S * self.SAMPLES_PER_FRAME] = wav[:S * self.SAMPLES_PER_FRAME]
        return self._window_batch(Wp).encode_window(buf[None])[0, :S]

    def infer(self, src, ref_path=None, out_dir=None, output_path=None, delay=None, ref_crop_lengths=None, alpha=1.0,
              spk_emb_collate_type="concat_mel", save_result=True, prompt=None, noise_seed=0, **sampling_kwargs):
        """:261-380 offline conversion: encode the source, ARVCWrapper.generate, code2wav.  `src` is a 44.1 kHz mono float
        array and the prompt is given as codes/embeddings (file I/O, resampling and the wav -> prompt encoders are rows
        N1/N2).  Returns the converted waveform as a numpy array like the reference."""
        src, src_path = self._load_src(src)
        if prompt is None:
            refs = self.load_and_crop_references(*self.process_ref_paths(ref_path, ref_crop_lengths))
            style_vectors = sampling_kwargs.pop("style_vectors", None)
            timbre_latents = sampling_kwargs.pop("timbre_latents", None)
            if spk_emb_collate_type == "avg" and len(refs) > 1 and (style_vectors is None or timbre_latents is None):
                # :284-307 -- the embeddings of every reference on its own, averaged; the two code streams still come from the
                # concatenated audio (:305-306, 326-339).  (Only this offline branch works upstream: the streaming calculate_prompt's
                # 'avg' branch, :389-424, reads an undefined variable, and stays refused there.)
                style_vectors, timbre_latents = self._avg_embeddings(refs, style_vectors, timbre_latents)
            prompt = self.calculate_prompt(refs, alpha=alpha, spk_emb_collate_type="concat_mel" if spk_emb_collate_type == "avg" else spk_emb_collate_type,
                                           style_vectors=style_vectors, timbre_latents=timbre_latents)
        ref_audio_codes, ref_content_codes, style_vectors, timbre_latents = [
            np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x) for x in prompt[:4]]
        src_codes = self.encode_content(src)
        S = src_codes.shape[0]
        d = 2 if delay is None else int(delay)
        kw = check_sampling_kwargs(sampling_kwargs)
        edits = kw.pop("edits", None)
        b = E.Batch(self.engine, n_streams=1, delay=d, voc_max_frames=S, **kw)
        try:
            if edits:
                b.set_sampler_edits(**edits)
            codes = b.generate(ref_content_codes.reshape(-1), ref_audio_codes.reshape(8, -1), src_codes, style_vectors.reshape(-1),
                               timbre_latents.reshape(32, -1), noise_seed=noise_seed)
            wav = b.vocode_window(codes[None])[0]
        finally:
            b.close()
        if save_result and (src_path or out_dir or output_path):      # (array input with nowhere to write: an extension of this mirror)
            self._save(wav, src_path, ref_path, out_dir, output_path)
        return wav

    # ---- per chunk ---------------------------------------------------------------------------------------
    def process_one_chunk(self, src_wav_chunk, pitch_shift=0.0):
        """src_wav_chunk [1, 2048*c] (torch or numpy) -> same type/shape (:492-596): zeros for the first `delay` chunks."""
        is_torch = hasattr(src_wav_chunk, "detach")
        x = src_wav_chunk.detach().cpu().numpy() if is_torch else np.asarray(src_wav_chunk)
        out = self.batch.step(x.reshape(1, -1).astype(np.float32))
        if is_torch:
            import torch

            return torch.from_numpy(out).to(src_wav_chunk.device)
        return out

    def stream_infer(self, src, ref_path=None, out_dir=None, encode_window_frames=128, decode_window_frames=64, max_prompt_frames=256,
                     max_seq_frames=768, buffer_frames=32, decode_chunk_frames=1, delay=None, ref_crop_lengths=None, alpha=1.0,
                     spk_emb_collate_type="concat_mel", save_result=True, prompt=None, noise_seed=0, style_vectors=None,
                     timbre_latents=None):
        """:598-689.  `src` / `ref_path`: wav paths (loaded and resampled to 44.1 kHz, audio_io.py) or float arrays already at
        44.1 kHz; `style_vectors` / `timbre_latents` stand in for the CAM++ / SparkTTS encoders (N1 iii/iv) unless the
        prompt is given whole."""
        src, src_path = self._load_src(src)
        if prompt is None:
            refs = self.load_and_crop_references(*self.process_ref_paths(ref_path, ref_crop_lengths))
            prompt = self.calculate_prompt(refs, alpha=alpha, spk_emb_collate_type=spk_emb_collate_type, style_vectors=style_vectors,
                                           timbre_latents=timbre_latents)
        self.prefill_prompt(None, max_prompt_frames=max_prompt_frames,
                            delay=2 if delay is None else delay, alpha=alpha, spk_emb_collate_type=spk_emb_collate_type,
                            prompt=prompt, noise_seed=noise_seed)
        self.setup_stream_caches(encode_window_frames, decode_window_frames, max_seq_frames, buffer_frames, decode_chunk_frames,
                                 pipeline=True)
        n = self.SAMPLES_PER_FRAME * decode_chunk_frames
        pad = n - (src.shape[0] % n)              # :648-649 pads a FULL extra chunk when already aligned
        src = np.concatenate([np.zeros(pad, np.float32), src])
        # the chunk loop (:650-675) in one engine call: the whole file is known, so the stages of consecutive chunks
        # overlap on the GPU; chunk by chunk through process_one_chunk gives the same samples
        pred = self.batch.stream_chunks(src[None])[0]
        if save_result and (src_path or out_dir):
            self._save(pred, src_path, ref_path, out_dir)
        return pred


def main(argv=None, weights=None, style_vectors=None, timbre_latents=None):
    """The reference's command line (evaluations/infer_arvc.py:691-743), flag for flag.  `weights` / `style_vectors` /
    `timbre_latents` exist for tests and for deployments without the speaker-encoder checkpoints."""
    import argparse
    from pathlib import Path

    parser = argparse.ArgumentParser(description="Inference Wrapper")
    parser.add_argument("--config_path", type=str, default="configs/config_firefly_arvcasr_8192_delay0_8.yaml")
    parser.add_argument("--checkpoint_path", type=str, default="pretrained_checkpoints/dual_ar_delay_0_8.pth")
    parser.add_argument("--src_path", type=str, default="./test_waves/azuma_0.wav")
    parser.add_argument("--ref_path", type=str, nargs="+", default="./test_waves/trump_0.wav", help="One or more reference audio paths")
    parser.add_argument("--out_dir", type=str, default="./audio_outputs/")
    parser.add_argument("--compile", action="store_true", help="Compile the model (here: replay the captured hipGraph of the steady step)")
    parser.add_argument("--delay", type=int, default=2, help="Delay for the decoder (in frames), 0 means no delay")
    parser.add_argument("--ref_crop_lengths", type=float, nargs="+", default=None, help="Crop lengths in seconds for each reference audio")
    parser.add_argument("--alpha", type=float, default=1.0, help="Noise mixing coefficient for speaker embeddings (1.0 = no noise, lower = more anonymization)")
    parser.add_argument("--simulate_streaming", action="store_true", help="Simulate streaming inference")
    parser.add_argument("--encode_window_frames", type=int, default=128, help="Encoder context window size in frames")
    parser.add_argument("--decode_window_frames", type=int, default=64, help="Vocoder context window size in frames")
    parser.add_argument("--max_prompt_frames", type=int, default=256, help="Maximum prompt length in frames")
    parser.add_argument("--max_seq_frames", type=int, default=768, help="Maximum sequence length in frames")
    parser.add_argument("--buffer_frames", type=int, default=32, help="Buffer frames when refilling prompt")
    parser.add_argument("--decode_chunk_frames", type=int, default=1, help="Decode chunk size in frames")
    args = parser.parse_args(argv)
    infer_wrapper = InferenceWrapper(
        args.config_path, args.checkpoint_path, compile_ar=args.compile,
        compile_decoder=args.compile if args.simulate_streaming else False,
        compile_encoder=args.compile if args.simulate_streaming else False, weights=weights)
    ref_path = args.ref_path if isinstance(args.ref_path, list) and len(args.ref_path) > 1 else args.ref_path[0] if isinstance(args.ref_path, list) else args.ref_path
    Path(args.out_dir).mkdir(parents=True, exist_ok=True)
    extra = {k: v for k, v in (("style_vectors", style_vectors), ("timbre_latents", timbre_latents)) if v is not None}
    if args.simulate_streaming:
        vc_wav = infer_wrapper.stream_infer(
            args.src_path, ref_path, args.out_dir, encode_window_frames=args.encode_window_frames,
            decode_window_frames=args.decode_window_frames, max_prompt_frames=args.max_prompt_frames, max_seq_frames=args.max_seq_frames,
            buffer_frames=args.buffer_frames, decode_chunk_frames=args.decode_chunk_frames, delay=args.delay,
            ref_crop_lengths=args.ref_crop_lengths, alpha=args.alpha, **extra)
    else:
        vc_wav = infer_wrapper.infer(args.src_path, ref_path, args.out_dir, delay=args.delay, ref_crop_lengths=args.ref_crop_lengths,
                                     alpha=args.alpha, **extra)
    return vc_wav


if __name__ == "__main__":
    main()
