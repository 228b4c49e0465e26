"""Host-side mirror of ``InferenceWrapper`` (evaluations/infer_arvc.py:26-689) on top of the HIP engine:
the chunk-by-chunk streaming surface -- ``prefill_prompt`` / ``setup_stream_caches`` /
``process_one_chunk`` / ``stream_infer`` -- with the reference's names, defaults and quirks, so callers
such as the CLI (__main__, :691-743) or the GUI's ``custom_infer`` (real-time-gui.py:32-49) can switch
by changing one import.

Not built in this round: the wav -> prompt encoders (CAM++ style vector,
SparkTTS timbre latents, ``firefly.encode`` audio codes).  ``prefill_prompt`` therefore also accepts
the prompt as codes/embeddings via ``prompt=`` (what ``calculate_prompt`` returns, :382-441); calling it
with raw reference audio raises NotImplementedError naming the missing rows.
"""
from __future__ import annotations

import os

import numpy as np

from . import engine as E


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
        self.engine = E.Engine(weights, device=device, ar_dtype=0)
        self.use_graph = bool(compile_ar or compile_decoder or compile_encoder)   # the reference's --compile
        self.batch = None
        self._prompt = None

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
        return {k: v for k, v in out.items() if hasattr(v, "dtype") and v.dtype.is_floating_point}

    # ---- prompt ------------------------------------------------------------------------------------------
    def wav2target_fn(self, waves):
        """:168-171 firefly.encode of a whole prompt -> acoustic codes int32 [1, 8, R], R = len // 2048 (right-padded with
        zeros to a multiple of 4 frames for the stride-4 front-end; causal, so the first R columns are unaffected)."""
        wav = np.asarray(waves.detach().cpu().numpy() if hasattr(waves, "detach") else waves, dtype=np.float32).reshape(-1)
        R = wav.shape[0] // self.SAMPLES_PER_FRAME
        Wp = ((R + 3) // 4) * 4
        buf = np.zeros(Wp * self.SAMPLES_PER_FRAME, np.float32)
        buf[:R * self.SAMPLES_PER_FRAME] = wav[:R * self.SAMPLES_PER_FRAME]
        b = E.Batch(self.engine, n_streams=1, encode_window_frames=Wp)
        try:
            return b.firefly_encode(buf[None])[:, :, :R]
        finally:
            b.close()

    def calculate_prompt(self, ref_wav_tensors, alpha=1.0, spk_emb_collate_type="concat_mel", style_vectors=None,
                         timbre_latents=None):
        """:382-441.  The two code streams of the prompt (firefly.encode audio codes, speech-tokenizer content codes) are
        computed on the device.  The CAM++ style vector and the SparkTTS timbre latents (N1 iii/iv, not built) come from
        `self.style_encoder(wav)` / `self.timbre_encoder(wav)` callables if the caller installed them, or from the
        `style_vectors` / `timbre_latents` arguments; alpha noise mixing (:426-427) is applied to them here."""
        import torch

        ref_list = ref_wav_tensors if isinstance(ref_wav_tensors, (list, tuple)) else [ref_wav_tensors]
        ref = np.concatenate([np.asarray(r.detach().cpu().numpy() if hasattr(r, "detach") else r, dtype=np.float32).reshape(-1)
                              for r in ref_list])            # :411 / :415 torch.cat(ref_wav_list, dim=-1)
        if style_vectors is None:
            if getattr(self, "style_encoder", None) is None:
                raise NotImplementedError("CAM++ style encoder (SURVEY.md §8f N1 iii) is not built: pass style_vectors= or set "
                                          "InferenceWrapper.style_encoder to a callable wav -> [1, 192]")
            style_vectors = self.style_encoder(ref)
        if timbre_latents is None:
            if getattr(self, "timbre_encoder", None) is None:
                raise NotImplementedError("SparkTTS timbre encoder (SURVEY.md §8f N1 iv) is not built: pass timbre_latents= or set "
                                          "InferenceWrapper.timbre_encoder to a callable wav -> [1, 32, 128]")
            timbre_latents = self.timbre_encoder(ref)
        style_vectors = self.apply_noise_mixing(torch.as_tensor(np.asarray(style_vectors), dtype=torch.float32), alpha)
        timbre_latents = self.apply_noise_mixing(torch.as_tensor(np.asarray(timbre_latents), dtype=torch.float32), alpha)
        ref_audio_codes = self.wav2target_fn(ref)                       # :431-434
        ref_content_codes = self.encode_content(ref)                    # :436-439
        return ref_audio_codes, ref_content_codes, style_vectors, timbre_latents, ref

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
        right-padded with zeros to a multiple of 4 frames (causal encoder: earlier codes are unaffected); limited to 256
        frames by the LDS-resident encoder attention (longer inputs are row N3 follow-up work)."""
        wav = np.asarray(wav, dtype=np.float32).reshape(-1)
        S = wav.shape[0] // self.SAMPLES_PER_FRAME
        Wp = ((S + 3) // 4) * 4
        if Wp > 256:
            raise NotImplementedError("offline encode of more than 256 frames (11.9 s) needs the tiled attention kernel (N3 follow-up)")
        buf = np.zeros(Wp * self.SAMPLES_PER_FRAME, np.float32)
        buf[:S * self.SAMPLES_PER_FRAME] = wav[:S * self.SAMPLES_PER_FRAME]
        b = E.Batch(self.engine, n_streams=1, encode_window_frames=Wp)
        try:
            codes = b.encode_window(buf[None])[0, :S]
        finally:
            b.close()
        return codes

    def infer(self, src, ref_path=None, out_dir=None, output_path=None, delay=None, ref_crop_lengths=None, alpha=1.0,
              spk_emb_collate_type="concat_mel", save_result=False, prompt=None, noise_seed=0, **sampling_kwargs):
        """:261-380 offline conversion: encode the source, ARVCWrapper.generate, code2wav.  `src` is a 44.1 kHz mono float
        array and the prompt is given as codes/embeddings (file I/O, resampling and the wav -> prompt encoders are rows
        N1/N2).  Returns the converted waveform as a numpy array like the reference."""
        src, src_path = self._load_src(src)
        if prompt is None:
            refs = self.load_and_crop_references(*self.process_ref_paths(ref_path, ref_crop_lengths))
            prompt = self.calculate_prompt(refs, alpha=alpha, spk_emb_collate_type=spk_emb_collate_type,
                                           style_vectors=sampling_kwargs.pop("style_vectors", None),
                                           timbre_latents=sampling_kwargs.pop("timbre_latents", None))
        ref_audio_codes, ref_content_codes, style_vectors, timbre_latents = [
            np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x) for x in prompt[:4]]
        src_codes = self.encode_content(src)
        S = src_codes.shape[0]
        d = 2 if delay is None else int(delay)
        kw = {k: sampling_kwargs[k] for k in ("temperature", "top_p") if k in sampling_kwargs}
        b = E.Batch(self.engine, n_streams=1, delay=d, voc_max_frames=S, **kw)
        try:
            codes = b.generate(ref_content_codes.reshape(-1), ref_audio_codes.reshape(8, -1), src_codes, style_vectors.reshape(-1),
                               timbre_latents.reshape(32, -1), noise_seed=noise_seed)
            wav = b.vocode_window(codes[None])[0]
        finally:
            b.close()
        if save_result:
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
                     spk_emb_collate_type="concat_mel", save_result=False, prompt=None, noise_seed=0, style_vectors=None,
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
        if save_result:
            self._save(pred, src_path, ref_path, out_dir)
        return pred
