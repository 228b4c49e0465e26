"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference
through tools/ref_harness.py) on CPU with the synthetic weights of
streamvoiceanon_amd.synth_weights.  Runs only in the build container; the fixtures it writes
are data (inputs are regenerated from seeds; outputs are the reference's tensors).

    python tools/make_golden.py            # writes tests/golden/*.npz   (~10 min on 8 vCPU; idempotent: `git diff --stat tests/golden` stays empty)

Fixture contents (SURVEY.md §8c "Golden vectors to capture"):
  encoder_s{0,1}.npz   BSQ indices [128] + pre-sign u [128,13] + sampled mel / backbone values
  vocoder_s0.npz       windowed vocoder: z samples, last-frame PCM, strided PCM samples
  stream_s0.npz        24-chunk stream (delay 2, chunk 1): content codes, audio codes, PCM of
                       3 frames + per-chunk checksums, per-frame top-32 slow/fast logits
  stream_reprefill.npz stream with max_seq_frames small enough to trigger a re-prefill
  stream_long_reprefill.npz  672-chunk stream at the reference's default max_seq_frames = 768 (first re-prefill at chunk ~ 646)
  stream_chunk4.npz    chunk = 4 stream (config-5 shape), delay 2
  offline_s0.npz       offline ARVCWrapper.generate codes for a short source
  offline_avg_s0.npz   the reference's offline infer() with spk_emb_collate_type="avg" and two references
  melfb.npz            mel filterbank checksums
  sampler_edits.npz    logits_to_probs with previous_tokens / repetition_penalty / suppress_tokens: probabilities per case
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_harness as rh  # noqa: E402
from streamvoiceanon_amd import specs, synth_weights as sw  # noqa: E402
from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def top32(x: torch.Tensor):
    v, i = torch.topk(x.flatten().float(), 32)
    return v.numpy(), i.numpy().astype(np.int32)


class NoiseFeed:
    """Replacement for modules.dual_ar_stream.multinomial_sample_one_no_sync (:1092-1096) that
    takes the Exp(1) noise from a queue instead of torch's global generator."""

    def __init__(self):
        self.q, self.live = [], False

    def __call__(self, probs):
        if self.live and self.q:
            nz = torch.from_numpy(self.q.pop(0))
        else:
            nz = torch.ones_like(probs)
        assert nz.shape == probs.shape
        return torch.argmax(probs / nz, dim=-1, keepdim=True).to(dtype=torch.int)


def check_specs(w):
    sp = specs.all_specs(prompt_path=True)
    for net, pre in ((w.model, "arvc."), (w.speech_tokenizer, "tok."), (w.firefly, "voc.")):
        sd = net.state_dict()
        for n, s in sp.items():
            if n.startswith(pre):
                assert tuple(sd[n[len(pre):]].shape) == tuple(s), n
    return len(sp)


def encoder_fixture(w, wseed, useed):
    x = torch.from_numpy(synth_utterance(useed, 262144))[None]
    tok = w.speech_tokenizer
    codes, lens = tok.encode(x, torch.LongTensor([262144]))
    mel = tok.spec_transform(x)
    feat = tok.backbone(mel)
    q = tok.quantizer
    z = q.pre_module(q.downsample(feat))
    u = torch.nn.functional.normalize(q.residual_bsq.rvqs[0].project_in(z.mT).float(), dim=-1)
    mel_idx = (np.arange(64) * 1279) % mel.numel()
    feat_idx = (np.arange(64) * 4093) % feat.numel()
    np.savez_compressed(
        os.path.join(OUT, f"encoder_s{wseed}.npz"),
        weight_seed=wseed, audio_seed=useed, n_samples=262144,
        codes=codes[0, 0].numpy(), u=u[0].numpy(),
        mel_idx=mel_idx, mel_val=mel.flatten()[mel_idx].numpy(), mel_sum=float(mel.double().sum()),
        feat_idx=feat_idx, feat_val=feat.flatten()[feat_idx].numpy(),
        z_last=z[0, :, -1].numpy(),
    )
    print("encoder fixture", wseed, "distinct codes", len(set(codes.flatten().tolist())),
          "min|u| q01", float(np.quantile(u.abs().min(-1).values.numpy(), 0.01)))


def encoder_long_fixture(w, wseed=0, useed=1005, frames=560):
    """Whole-utterance encode beyond the transformer's 512-token causal window (WindowLimitedTransformer,
    modules/vqgan/windowed_transformer.py:291-304): the offline `infer` path encodes the entire source at once (:334-339)."""
    n = frames * 2048
    x = torch.from_numpy(synth_utterance(useed, n))[None]
    codes, lens = w.speech_tokenizer.encode(x, torch.LongTensor([n]))
    q = w.speech_tokenizer.quantizer
    z = q.pre_module(q.downsample(w.speech_tokenizer.backbone(w.speech_tokenizer.spec_transform(x))))
    u = torch.nn.functional.normalize(q.residual_bsq.rvqs[0].project_in(z.mT).float(), dim=-1)
    np.savez_compressed(os.path.join(OUT, f"encoder_long_s{wseed}.npz"), weight_seed=wseed, audio_seed=useed, n_samples=n,
                        codes=codes[0, 0].numpy(), min_abs_u=u[0].abs().min(-1).values.numpy())
    print("long encoder fixture", codes.shape, "distinct", len(set(codes.flatten().tolist())), "min|u|", float(u.abs().min()))


def vocoder_fixture(w, wseed):
    cseed = 77
    codes = np.floor(sw.uniform01(cseed, "voc.codes", 8 * 64).astype(np.float64) * 1000).astype(np.int64).reshape(1, 8, 64)
    ct = torch.from_numpy(codes)
    z = w.firefly.quantizer.decode(ct)
    wav = w.firefly.head(z)
    z_idx = (np.arange(64) * 2039) % z.numel()
    s_idx = (np.arange(512) * 257) % wav.numel()
    np.savez_compressed(
        os.path.join(OUT, f"vocoder_s{wseed}.npz"),
        weight_seed=wseed, code_seed=cseed, codes=codes,
        z_idx=z_idx, z_val=z.flatten()[z_idx].numpy(),
        pcm_idx=s_idx, pcm_val=wav.flatten()[s_idx].numpy(),
        pcm_last_frame=wav[0, 0, -2048:].numpy(), pcm_sum=float(wav.double().sum()),
    )
    print("vocoder fixture", float(wav.std()))


def stream_fixture(w, feed, name, wseed, useed, pseed, n_chunks, chunk=1, delay=2, max_seq_frames=768,
                   buffer_frames=32, prompt_frames=107, full_pcm_frames=(2, 3, -1), logit_frames=None):
    """logit_frames: (first, last) decoded-frame range whose top-32 logits are stored (default: every frame); the
    long fixture keeps the frames around its re-prefill only, codes / hidden states / PCM checksums for all."""
    import modules.dual_ar_stream as das

    ac, cc, style, timbre = synth_prompt(pseed, prompt_frames)
    tup = (torch.from_numpy(ac)[None], torch.from_numpy(cc)[None], torch.from_numpy(style)[None],
           torch.from_numpy(timbre)[None], torch.zeros(1, prompt_frames * 2048))
    had_cp = "calculate_prompt" in w.__dict__
    saved_cp = w.__dict__.get("calculate_prompt")
    w.calculate_prompt = lambda ref, alpha=1.0, spk_emb_collate_type="concat_mel": tup
    ar = w.model.decoder.model
    rec = dict(slow_v=[], slow_i=[], fast_v=[], fast_i=[], hidden=[])
    state = dict(frame=0, live=False)
    orig_fg, orig_ff = ar.forward_generate, ar.forward_generate_fast

    def fg(*a, **k):
        r = orig_fg(*a, **k)
        if state["live"]:
            v, i = top32(r.logits)
            rec["slow_v"].append(v); rec["slow_i"].append(i)
            rec["hidden"].append(r.hidden_states.flatten()[:16].numpy().copy())
        return r

    def ff(*a, **k):
        r = orig_ff(*a, **k)
        if state["live"]:
            v, i = top32(r)
            rec["fast_v"].append(v); rec["fast_i"].append(i)
        return r

    ar.forward_generate, ar.forward_generate_fast = fg, ff
    orig_decode_one = w.model.__class__.decode_one

    def decode_one(code):
        ns, nf = frame_noise(useed, state["frame"])
        state["frame"] += 1
        feed.q = [ns] + [nf[i] for i in range(8)]
        feed.live = state["live"] = True
        r = orig_decode_one(w.model, code)
        feed.live = state["live"] = False
        return r

    w.model.decode_one = decode_one
    # prompt prefill: capture the last-token logits of the prefill pass
    pre = {}
    orig_fg2 = ar.forward_generate

    def fg_pre(*a, **k):
        r = orig_fg2(*a, **k)
        pre.setdefault("v", top32(r.logits))
        return r

    ar.forward_generate = fg_pre
    w.prefill_prompt([None], max_prompt_frames=256, delay=delay)
    ar.forward_generate = fg
    w.setup_stream_caches(encode_window_frames=128, decode_window_frames=64, max_seq_frames=max_seq_frames,
                          buffer_frames=buffer_frames, decode_chunk_frames=chunk)
    n = 2048 * chunk
    src = torch.from_numpy(synth_utterance(useed, n * n_chunks))[None]
    content, pcm_sum, pcm_abs, pcm_full = [], [], [], {}
    t0 = time.time()
    for i in range(n_chunks):
        out = w.process_one_chunk(src[:, i * n:(i + 1) * n])
        content.append(w.src_content_codes[0, -chunk:].numpy().copy())
        pcm_sum.append(float(out.double().sum()))
        pcm_abs.append(float(out.double().abs().sum()))
        pcm_full[i] = out[0].numpy().copy()
    keep = sorted({(f if f >= 0 else n_chunks + f) for f in full_pcm_frames})
    audio_codes = w.pred_codes[0].numpy().copy()          # [8, frames decoded]
    lf0, lf1 = logit_frames if logit_frames else (0, len(rec["slow_v"]))
    extra = dict(logit_first=lf0) if logit_frames else {}
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        weight_seed=wseed, audio_seed=useed, prompt_seed=pseed, prompt_frames=prompt_frames,
        n_chunks=n_chunks, chunk=chunk, delay=delay, max_seq_frames=max_seq_frames, buffer_frames=buffer_frames,
        content_codes=np.concatenate(content), audio_codes=audio_codes,
        pcm_sum=np.array(pcm_sum), pcm_abs=np.array(pcm_abs),
        pcm_full_idx=np.array(keep), pcm_full=np.stack([pcm_full[k] for k in keep]),
        slow_top_v=np.stack(rec["slow_v"])[lf0:lf1], slow_top_i=np.stack(rec["slow_i"])[lf0:lf1],
        fast_top_v=np.stack(rec["fast_v"]).reshape(-1, 8, 32)[lf0:lf1], fast_top_i=np.stack(rec["fast_i"]).reshape(-1, 8, 32)[lf0:lf1],
        hidden16=np.stack(rec["hidden"]),
        prefill_top_v=pre["v"][0], prefill_top_i=pre["v"][1],
        final_pos=int(w.model.decoder.cached_kv_pos[-1]), **extra,
    )
    ar.forward_generate, ar.forward_generate_fast = orig_fg, orig_ff
    del w.model.decode_one
    # every patch of this function is undone: the wrapper goes on to other fixtures (prompt_fixture calls the REAL
    # calculate_prompt; round 4's recipe left the lambda in place and a full run wrote the synthetic prompt into prompt_s0.npz)
    if had_cp:
        w.calculate_prompt = saved_cp
    else:
        del w.calculate_prompt
    print(name, "frames", audio_codes.shape, "time %.1fs" % (time.time() - t0), "final pos", int(w.model.decoder.cached_kv_pos[-1]))


def long_reprefill_fixture(w, feed):
    """One stream at the reference's DEFAULT max_seq_frames = 768 / buffer_frames = 32 (evaluations/infer_arvc.py:443-460) run past
    its first re-prefill (:547-564): prompt 107 frames -> 33 + 2 * 107 = 247 positions, +2 per source frame, re-prefill when
    pos // 2 >= 768, i.e. in chunk ~ 646; 672 chunks = 31.2 s of audio.  Full PCM of the chunks around the re-prefill, top-32
    logits of the frames [628, 670)."""
    stream_fixture(w, feed, "stream_long_reprefill", 0, useed=1006, pseed=2006, n_chunks=672, max_seq_frames=768,
                   buffer_frames=32, full_pcm_frames=(640, 644, 645, 646, 647, 648, 650, -1), logit_frames=(628, 670))


def offline_fixture(w, feed, wseed, useed, pseed, src_frames=12, prompt_frames=40, delay=2):
    import modules.dual_ar_stream as das

    ac, cc, style, timbre = synth_prompt(pseed, prompt_frames)
    src_codes = np.floor(sw.uniform01(useed, "offline.src", src_frames).astype(np.float64) * 8192).astype(np.int64)
    # noise for step s = frame_noise(useed, s); decode_one_token_ar draws 1 + 8 times per step
    step = dict(i=0)
    orig = das.decode_one_token_ar

    def patched(*a, **k):
        ns, nf = frame_noise(useed, step["i"])
        step["i"] += 1
        feed.q = [ns] + [nf[j] for j in range(8)]
        feed.live = True
        r = orig(*a, **k)
        feed.live = False
        return r

    das.decode_one_token_ar = patched
    w.model.set_delay(delay=delay)
    codes = w.model.generate(
        ref_content_codes=torch.from_numpy(cc)[None], ref_audio_codes=torch.from_numpy(ac)[None],
        src_content_codes=torch.from_numpy(src_codes)[None], style_vectors=torch.from_numpy(style)[None],
        timbre_latents=torch.from_numpy(timbre)[None])
    das.decode_one_token_ar = orig
    wav = w.code2wav_fn(codes)
    np.savez_compressed(os.path.join(OUT, f"offline_s{wseed}.npz"), weight_seed=wseed, audio_seed=useed,
                        prompt_seed=pseed, prompt_frames=prompt_frames, delay=delay, src_codes=src_codes,
                        codes=codes.numpy(), pcm_sum=float(wav.double().sum()), pcm_last=wav[0, 0, -2048:].numpy())
    print("offline fixture", codes.shape)


def offline_avg_fixture(w, feed, wseed=0, useed=1010, ref_seeds=(7910, 7911), src_frames=12, ref_frames=(22, 17), delay=2):
    """The reference's offline infer() with spk_emb_collate_type="avg" and TWO references (evaluations/infer_arvc.py:261-380, the
    branch :284-307): its own control flow end to end -- per-reference calculate_style_vec / calculate_timbre_latent through the
    reference's CAM++ and SpeakerEncoder modules (synthetic weights, as in prompt_encoder_fixture), torch.mean over the stack, both code
    streams from the concatenated audio, generate, code2wav.  Third-party calls it makes are served by the harness: librosa.load
    returns the synthetic utterance named by the "path", torchaudio.functional.resample is streamvoiceanon_amd.audio_io.resample (a
    stated deviation from soxr / torchaudio's kernel: the fixture pins the reference's FLOW, the resampler is the build's own),
    kaldi.fbank / MelSpectrogram are oracle/prompt_oracle.py's restatements."""
    import evaluations.infer_arvc as ria
    import modules.dual_ar_stream as das
    from modules.bicodec_speaker_encoder.speaker_encoder import SpeakerEncoder
    from modules.campplus.DTDNN import CAMPPlus
    from oracle import prompt_oracle as PO
    from streamvoiceanon_amd import audio_io, specs

    class Mel(torch.nn.Module):
        hop_length = 320

        def forward(self, wav):
            return torch.stack([PO.mel_spectrogram_16k(x) for x in wav])

    W = {k: torch.from_numpy(sw.generate(wseed, k, shp)) for k, shp in specs.prompt_encoder_specs().items()}
    cam = CAMPPlus(feat_dim=80, embedding_size=192).eval()
    cam.load_state_dict({k[6:]: v for k, v in W.items() if k.startswith("style.")}, strict=False)
    spk = SpeakerEncoder(mel_fn=Mel(), input_dim=128, out_dim=1024, latent_dim=128, token_num=32, fsq_levels=[4] * 6, fsq_num_quantizers=1).eval()
    missing, unexpected = spk.load_state_dict({k[7:]: v for k, v in W.items() if k.startswith("timbre.")}, strict=False)
    assert not unexpected
    audio = {"src": synth_utterance(useed, src_frames * 2048)}
    for i, (rs, rf) in enumerate(zip(ref_seeds, ref_frames)):
        audio[f"ref{i}"] = synth_utterance(rs, rf * 2048 + 777 * i)            # (the second one not a whole number of frames)
    saved = dict(load=getattr(sys.modules["librosa"], "load", None), resample=sys.modules["torchaudio"].functional.resample,
                 fbank=getattr(ria.kaldi, "fbank", None), dec=das.decode_one_token_ar, gen=w.model.generate)
    inst = {k: w.__dict__.pop(k) for k in ("calculate_style_vec", "calculate_timbre_latent") if k in w.__dict__}      # the class's own methods run
    sys.modules["librosa"].load = ria.librosa.load = lambda path, sr=None: (audio[str(path)].copy(), sr)
    sys.modules["torchaudio"].functional.resample = lambda x, orig_freq=None, new_freq=None: torch.from_numpy(
        np.stack([audio_io.resample(r, orig_freq, new_freq) for r in x.numpy()]))
    ria.kaldi.fbank = lambda wav, num_mel_bins=80, dither=0, sample_frequency=16000: PO.kaldi_fbank(wav)
    w.style_encoder, w.timbre_encoder = cam, spk
    cap, step = {}, dict(i=0)

    def patched(*a, **k):
        ns, nf = frame_noise(useed, step["i"])
        step["i"] += 1
        feed.q = [ns] + [nf[j] for j in range(8)]
        feed.live = True
        r = saved["dec"](*a, **k)
        feed.live = False
        return r

    def gen(**k):
        cap.update({n: k[n].clone() for n in ("style_vectors", "timbre_latents", "ref_content_codes", "ref_audio_codes", "src_content_codes")})
        cap["codes"] = saved["gen"](**k)
        return cap["codes"]

    das.decode_one_token_ar = patched
    w.model.generate = gen
    try:
        wav = w.infer("src", ["ref0", "ref1"], delay=delay, alpha=1.0, spk_emb_collate_type="avg", save_result=False)
    finally:
        das.decode_one_token_ar = saved["dec"]
        w.model.generate = saved["gen"]
        sys.modules["torchaudio"].functional.resample = saved["resample"]
        ria.kaldi.fbank = saved["fbank"]
        if saved["load"] is None:
            del sys.modules["librosa"].load
        else:
            sys.modules["librosa"].load = saved["load"]
        w.__dict__.update(inst)
        del w.style_encoder, w.timbre_encoder
    np.savez_compressed(os.path.join(OUT, f"offline_avg_s{wseed}.npz"), weight_seed=wseed, audio_seed=useed, ref_seeds=np.array(ref_seeds),
                        src_samples=audio["src"].shape[0], ref_samples=np.array([audio["ref0"].shape[0], audio["ref1"].shape[0]]), delay=delay,
                        style=cap["style_vectors"].numpy(), timbre=cap["timbre_latents"].numpy(),
                        ref_content_codes=cap["ref_content_codes"].numpy(), ref_audio_codes=cap["ref_audio_codes"].numpy().astype(np.int32),
                        src_content_codes=cap["src_content_codes"].numpy(), codes=cap["codes"].numpy(),
                        pcm_sum=float(np.asarray(wav, np.float64).sum()), pcm_last=np.asarray(wav, np.float32).reshape(-1)[-2048:])
    print("offline avg fixture", tuple(cap["codes"].shape), "style", tuple(cap["style_vectors"].shape), "timbre", tuple(cap["timbre_latents"].shape))


def prompt_fixture(w, wseed, useed, frames=48):
    """calculate_prompt (evaluations/infer_arvc.py:382-441) with the CAM++ / SparkTTS encoders stubbed by the harness:
    pins firefly.encode (wav2target_fn, :168-171) and speech_tokenizer.encode of the PROMPT."""
    wav = torch.from_numpy(synth_utterance(useed, frames * 2048))[None]
    ac, cc, style, timbre, _ = w.calculate_prompt(wav, alpha=1.0)
    voc = w.firefly
    z = voc.quantizer.downsample(voc.backbone(voc.spec_transform(wav)))
    z_idx = (np.arange(64) * 2039) % z.numel()
    np.savez_compressed(os.path.join(OUT, f"prompt_s{wseed}.npz"), weight_seed=wseed, audio_seed=useed, n_samples=frames * 2048,
                        ref_audio_codes=ac[0].numpy().astype(np.int32), ref_content_codes=cc.reshape(-1).numpy(),
                        z_idx=z_idx, z_val=z.flatten()[z_idx].numpy(), z_last=z[0, :, -1].numpy())
    print("prompt fixture", tuple(ac.shape), tuple(cc.shape), "distinct audio codes", len(set(ac.flatten().tolist())))


def prompt_encoder_fixture(wseed=0):
    """The reference's CAM++ and SparkTTS SpeakerEncoder modules (modules/campplus/DTDNN.py, modules/bicodec_speaker_encoder/*) with
    synthetic weights, driven as calculate_style_vec / calculate_timbre_latent drive them (evaluations/infer_arvc.py:179-223).
    torchaudio is not installed: its two front-ends (kaldi.fbank, transforms.MelSpectrogram) are the restatements of
    oracle/prompt_oracle.py (parity unpinned for those two functions); everything behind them is the reference's own code."""
    from modules.bicodec_speaker_encoder.speaker_encoder import SpeakerEncoder
    from modules.campplus.DTDNN import CAMPPlus
    from oracle import prompt_oracle as PO
    from streamvoiceanon_amd import specs

    class Mel(torch.nn.Module):
        hop_length = 320

        def forward(self, wav):
            return torch.stack([PO.mel_spectrogram_16k(w) for w in wav])

    W = {k: torch.from_numpy(sw.generate(wseed, k, shp)) for k, shp in specs.prompt_encoder_specs().items()}
    cam = CAMPPlus(feat_dim=80, embedding_size=192).eval()
    sd = cam.state_dict()
    spec_s = specs.style_specs()
    assert {("style." + k) for k in sd if not k.endswith("num_batches_tracked")} == set(spec_s)
    for k, v in sd.items():
        if not k.endswith("num_batches_tracked"):
            assert tuple(v.shape) == spec_s["style." + k], k
    cam.load_state_dict({k[6:]: v for k, v in W.items() if k.startswith("style.")}, strict=False)
    spk = SpeakerEncoder(mel_fn=Mel(), input_dim=128, out_dim=1024, latent_dim=128, token_num=32, fsq_levels=[4] * 6, fsq_num_quantizers=1).eval()
    sd = spk.state_dict()
    for k, shp in specs.timbre_specs().items():
        assert tuple(sd[k[7:]].shape) == shp, k
    missing, unexpected = spk.load_state_dict({k[7:]: v for k, v in W.items() if k.startswith("timbre.")}, strict=False)
    assert not unexpected
    out = dict(weight_seed=wseed)
    for tag, useed, n in (("a", 7900, 16000 * 3), ("b", 7901, 16000 * 2 + 3333)):       # 298 (even) / 219 (odd) fbank frames
        wav = torch.from_numpy(synth_utterance(useed, n))[None]
        feat = PO.kaldi_fbank(wav)
        feat = feat - feat.mean(dim=0, keepdim=True)
        style = cam(feat[None], torch.tensor([feat.shape[0] // 2]))          # evaluations/infer_arvc.py:196-210
        zq, idx = spk.tokenize_wav(wav, torch.tensor([wav.shape[1]]))          # :218-222
        out.update({f"{tag}_audio_seed": useed, f"{tag}_n": n, f"{tag}_style": style[0].numpy(), f"{tag}_timbre": zq.mT[0].numpy(),
                    f"{tag}_fsq_idx": idx[0].numpy(), f"{tag}_feat_sum": feat.double().sum(0).numpy(), f"{tag}_feat_row7": feat[7].numpy()})
    np.savez_compressed(os.path.join(OUT, "prompt_encoders_s0.npz"), **out)
    print("prompt encoder fixture", {k: getattr(v, "shape", v) for k, v in out.items()})


def sampler_edits_fixture(das):
    """logits_to_probs (modules/dual_ar_stream.py:1099-1132) with its optional edits, on seeded logits: the reference's output
    probabilities per case (inputs are stored too: they are data, 1000-8192 floats)."""
    rng = np.random.default_rng(77)
    out = {}
    cases = [(1000, 48, 0, 1.5, 0.7, 0.7), (8192, 64, 300, 50.0, 1.1, 0.9), (1000, 0, 0, 1.5, 0.7, 0.7), (8192, 0, 17, 1.5, 0.7, 0.7),
             (1000, 200, 0, 0.5, 1.3, 0.95)]
    for k, (V, W, ns, pen, temp, top_p) in enumerate(cases):
        logits = (rng.standard_normal(V) * 3.0).astype(np.float32)
        top = np.argsort(-logits)[:max(W // 2, 1)]
        prev = np.concatenate([top, top[: W // 4], rng.integers(0, V, W)])[:W].astype(np.int64)     # strong candidates, duplicates, random ones
        sup = rng.integers(0, V, ns).astype(np.int64)
        probs = das.logits_to_probs(torch.from_numpy(logits.copy()), previous_tokens=torch.from_numpy(prev) if W else None,
                                    suppress_tokens=[int(t) for t in sup] if ns else None, temperature=temp, top_p=top_p,
                                    repetition_penalty=pen)
        out[f"logits{k}"] = logits; out[f"prev{k}"] = prev; out[f"suppress{k}"] = sup
        out[f"params{k}"] = np.array([pen, temp, top_p], np.float64); out[f"probs{k}"] = probs.numpy()
        out[f"noise{k}"] = rng.exponential(1.0, V).astype(np.float32)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "sampler_edits.npz"), **out)
    print("sampler_edits.npz written")


def main():
    os.makedirs(OUT, exist_ok=True)
    rh.install_stubs()
    import modules.dual_ar_stream as das

    feed = NoiseFeed()
    das.multinomial_sample_one_no_sync = feed
    fb = rh.melscale_fbanks(1025, 0.0, 22050.0, 160, 44100, norm="slaney", mel_scale="slaney")
    np.savez_compressed(os.path.join(OUT, "melfb.npz"), col_sum=fb.sum(0).numpy(), row_sum=fb.sum(1).numpy(),
                        peak=fb.max(0).values.numpy(), argpeak=fb.argmax(0).numpy())
    only = sys.argv[1] if len(sys.argv) > 1 else None     # e.g. `make_golden.py prompt` adds one fixture without rewriting the rest
    if only in (None, "sampler_edits"):
        sampler_edits_fixture(das)
        if only:
            return
    if only == "encoder_long":
        encoder_long_fixture(rh.build_wrapper(seed=0))
        return
    if only == "stream_long":
        long_reprefill_fixture(rh.build_wrapper(seed=0), feed)
        return
    if only == "offline_avg":
        offline_avg_fixture(rh.build_wrapper(seed=0), feed)
        return
    if only in (None, "prompt_encoders"):
        prompt_encoder_fixture(0)
        if only:
            return
    for wseed in (0, 1):
        w = rh.build_wrapper(seed=wseed)
        if only == "prompt":
            if wseed == 0:
                prompt_fixture(w, 0, useed=1004)
            continue
        print("spec table entries checked:", check_specs(w))
        encoder_fixture(w, wseed, 1000 + wseed)
        if wseed == 0:
            vocoder_fixture(w, wseed)
            stream_fixture(w, feed, "stream_s0", 0, useed=1000, pseed=2000, n_chunks=24)
            stream_fixture(w, feed, "stream_reprefill", 0, useed=1001, pseed=2001, n_chunks=30,
                           max_seq_frames=136, buffer_frames=32, full_pcm_frames=(-1,))
            stream_fixture(w, feed, "stream_chunk4", 0, useed=1002, pseed=2002, n_chunks=6, chunk=4,
                           full_pcm_frames=(-1,))
            offline_fixture(w, feed, 0, useed=1003, pseed=2003)
            encoder_long_fixture(w)
            prompt_fixture(w, 0, useed=1004)
            offline_avg_fixture(w, feed)
            long_reprefill_fixture(w, feed)


if __name__ == "__main__":
    main()
