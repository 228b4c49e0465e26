"""The CPU oracle (oracle/sva_oracle.py) replayed against outputs of the REAL reference
captured by tools/make_golden.py (tests/golden/*.npz).  This is what pins the oracle:
the reference repository itself holds no tests or golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import sva_oracle as O
from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance
from streamvoiceanon_amd import synth_weights as sw

torch.set_grad_enabled(False)


def test_mel_filterbank_matches_reference():
    g = load_golden("melfb")
    fb = O.slaney_mel_fb()
    assert fb.shape == (1025, 160)
    np.testing.assert_allclose(fb.sum(0).numpy(), g["col_sum"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(fb.sum(1).numpy(), g["row_sum"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(fb.argmax(0).numpy(), g["argpeak"])


def test_mel_filterbank_published_values():
    """torchaudio is not installable here, so the slaney filterbank is a restatement ("parity unpinned").  What CAN be
    pinned from published material: (1) the example in librosa's `filters.mel` documentation -- `mel(sr=22050, n_fft=2048)` prints
    `[[0., 0.016, ..., 0., 0.], ...]` (128 mels, htk=False, norm="slaney"; torchaudio's slaney/slaney bank is its transpose);
    (2) the two anchors that DEFINE the Slaney scale (Auditory Toolbox): 1 kHz = mel 15, 6.4 kHz = mel 42, linear 200/3 Hz per mel
    below 1 kHz; (3) Slaney normalisation = unit area of every triangle in Hz."""
    fb = O.slaney_mel_fb(n_freqs=1025, f_min=0.0, f_max=11025.0, n_mels=128, sample_rate=22050)
    assert fb.shape == (1025, 128)
    assert round(float(fb[0, 0]), 3) == 0.0 and round(float(fb[1, 0]), 3) == 0.016 and float(fb[-1, 0]) == 0.0
    df = 22050 / 2048
    area = fb.sum(0) * df                              # Riemann sum of each triangle
    assert float((area[:100] - 1.0).abs().max()) < 0.12     # (sampling error of narrow low filters; exact area is 1)
    assert float((area[60:] - 1.0).abs().max()) < 0.02
    peaks = fb.argmax(0).double() * df                 # centre frequencies
    low = peaks[peaks < 900]
    steps = low[1:] - low[:-1]
    assert float((steps - steps.mean()).abs().max()) <= df + 1e-6      # linear region: equal spacing (to one FFT bin)
    # 160-mel bank of the path (spectrogram.py:93-101): centres cross 1 kHz where mel = 15 of hz_to_mel(22050) * m / 161
    fb160 = O.slaney_mel_fb()
    m_max = 15.0 + np.log(22050.0 / 1000.0) / (np.log(6.4) / 27.0)
    k = int(np.ceil(15.0 * 161 / m_max))               # first filter whose centre is >= 1 kHz
    c = fb160.argmax(0).double() * (44100 / 2048)
    assert float(c[k - 2]) < 1000.0 <= float(c[k - 1]) + 44100 / 2048


@pytest.mark.parametrize("wseed", [0, 1])
def test_encoder_indices_bit_exact(wseed, weights0, weights1):
    g = load_golden(f"encoder_s{wseed}")
    W = weights0 if wseed == 0 else weights1
    x = torch.from_numpy(synth_utterance(int(g["audio_seed"]), int(g["n_samples"])))[None]
    taps = {}
    codes = O.encode_window(x, W, taps=taps)
    assert codes.shape == (1, 1, 128) and codes.dtype == torch.int64
    np.testing.assert_array_equal(codes[0, 0].numpy(), g["codes"])           # BSQ indices: bit-exact
    np.testing.assert_allclose(taps["u"][0].numpy(), g["u"], atol=2e-5)
    np.testing.assert_allclose(taps["mel"].flatten()[g["mel_idx"]].numpy(), g["mel_val"], atol=1e-4)
    np.testing.assert_allclose(taps["feat"].flatten()[g["feat_idx"]].numpy(), g["feat_val"], atol=1e-4)


def test_vocoder_window(weights0):
    g = load_golden("vocoder_s0")
    codes = torch.from_numpy(g["codes"])
    taps = {}
    wav = O.vocode_window(codes, weights0, taps=taps)
    assert wav.shape == (1, 1, 131072)
    np.testing.assert_allclose(taps["z"].flatten()[g["z_idx"]].numpy(), g["z_val"], atol=1e-5)
    np.testing.assert_allclose(wav[0, 0, -2048:].numpy(), g["pcm_last_frame"], atol=1e-5)   # fp32 path tolerance
    np.testing.assert_allclose(wav.flatten()[g["pcm_idx"]].numpy(), g["pcm_val"], atol=1e-5)


def _run_stream(g, W, forced=False):
    useed, pseed = int(g["audio_seed"]), int(g["prompt_seed"])
    ac, cc, style, timbre = synth_prompt(pseed, int(g["prompt_frames"]))
    chunk, n_chunks = int(g["chunk"]), int(g["n_chunks"])
    sess = O.StreamSession(
        W, torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(style), torch.from_numpy(timbre),
        noise_fn=lambda f: tuple(torch.from_numpy(a) for a in frame_noise(useed, f)),
        delay=int(g["delay"]), max_seq_frames=int(g["max_seq_frames"]), buffer_frames=int(g["buffer_frames"]),
        decode_chunk_frames=chunk)
    n = 2048 * chunk
    src = torch.from_numpy(synth_utterance(useed, n * n_chunks))[None]
    outs = []
    for i in range(n_chunks):
        outs.append(sess.process_one_chunk(src[:, i * n:(i + 1) * n]))
    return sess, outs


@pytest.mark.parametrize("name", ["stream_s0", "stream_reprefill", "stream_chunk4"])
def test_stream_matches_reference(name, weights0):
    g = load_golden(name)
    sess, outs = _run_stream(g, weights0)
    content = torch.cat([r["content"] for r in sess.trace]).numpy()
    np.testing.assert_array_equal(content, g["content_codes"])
    np.testing.assert_array_equal(sess.pred_codes.numpy(), g["audio_codes"])       # same noise -> same codes
    assert sess.ar.last_pos == int(g["final_pos"])
    for k, idx in enumerate(g["pcm_full_idx"]):
        np.testing.assert_allclose(outs[int(idx)][0].numpy(), g["pcm_full"][k], atol=1e-5)
    sums = np.array([float(o.double().sum()) for o in outs])
    np.testing.assert_allclose(sums, g["pcm_sum"], atol=1e-3)
    if name == "stream_reprefill":
        assert sess.n_reprefill > 0
    # seam fixtures captured inside the reference's forward passes: last-token logits of the prompt prefill, the pre-norm hidden
    # state and the top-32 slow / fast logits of every decoded frame (sampled, not teacher-forced: same codes => same inputs)
    np.testing.assert_allclose(sess.prefill_logits.numpy()[g["prefill_top_i"]], g["prefill_top_v"], atol=2e-4)
    assert int(sess.prefill_logits.argmax()) == int(g["prefill_top_i"][0])
    hid = [h for r in sess.trace for h in r["hidden"]]
    slow = [x for r in sess.trace for x in r["slow_logits"]]
    fast = [x for r in sess.trace for x in r["fast_logits"]]
    assert len(hid) == g["hidden16"].shape[0] == g["slow_top_v"].shape[0]
    for f in range(len(hid)):
        np.testing.assert_allclose(hid[f][:16].numpy(), g["hidden16"][f], atol=2e-4)
        np.testing.assert_allclose(slow[f].numpy()[g["slow_top_i"][f]], g["slow_top_v"][f], atol=2e-4)
        for cb in range(8):
            np.testing.assert_allclose(fast[f][cb].numpy()[g["fast_top_i"][f, cb]], g["fast_top_v"][f, cb], atol=2e-4)


def test_long_stream_default_reprefill_matches_reference(weights0):
    """The 672-chunk fixture at the reference's default max_seq_frames = 768 (tools/make_golden.py stream_long): the oracle's AR state
    machine -- prompt prefill, delay fill, 645 decoded frames, the re-prefill of evaluations/infer_arvc.py:547-564 when pos // 2 >= 768,
    24 more frames on the rebuilt cache -- against the reference's codes, KV position, hidden state of every frame and the top-32
    logits of the frames around the re-prefill.  To keep the CPU suite short the content codes come from the fixture (the encoder has
    its own fixtures) and the vocoder runs for the stored chunks only."""
    g = load_golden("stream_long_reprefill")
    useed, pseed = int(g["audio_seed"]), int(g["prompt_seed"])
    ac, cc, style, timbre = synth_prompt(pseed, int(g["prompt_frames"]))
    n_chunks = int(g["n_chunks"])
    sess = O.StreamSession(weights0, torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(style), torch.from_numpy(timbre),
                           noise_fn=lambda f: tuple(torch.from_numpy(a) for a in frame_noise(useed, f)), delay=int(g["delay"]),
                           max_seq_frames=int(g["max_seq_frames"]), buffer_frames=int(g["buffer_frames"]))
    keep = {int(i): k for k, i in enumerate(g["pcm_full_idx"])}
    zero = torch.zeros(1, 2048)
    content = torch.from_numpy(g["content_codes"])
    for i in range(n_chunks):
        out = sess.process_one_chunk(zero, content_override=content[i:i + 1], vocode=i in keep)
        if i in keep:
            np.testing.assert_allclose(out[0].numpy(), g["pcm_full"][keep[i]], atol=1e-5)
    assert sess.n_reprefill == 1
    np.testing.assert_array_equal(sess.pred_codes.numpy(), g["audio_codes"])
    assert sess.ar.last_pos == int(g["final_pos"])
    hid = [h for r in sess.trace for h in r["hidden"]]
    slow = [x for r in sess.trace for x in r["slow_logits"]]
    fast = [x for r in sess.trace for x in r["fast_logits"]]
    assert len(hid) == g["hidden16"].shape[0] == n_chunks - 2
    lf = int(g["logit_first"])
    for f in range(len(hid)):
        np.testing.assert_allclose(hid[f][:16].numpy(), g["hidden16"][f], atol=2e-4)
        if lf <= f < lf + g["slow_top_v"].shape[0]:
            np.testing.assert_allclose(slow[f].numpy()[g["slow_top_i"][f - lf]], g["slow_top_v"][f - lf], atol=2e-4)
            for cb in range(8):
                np.testing.assert_allclose(fast[f][cb].numpy()[g["fast_top_i"][f - lf, cb]], g["fast_top_v"][f - lf, cb], atol=2e-4)


def test_offline_generate_matches_reference(weights0):
    g = load_golden("offline_s0")
    useed = int(g["audio_seed"])
    ac, cc, style, timbre = synth_prompt(int(g["prompt_seed"]), int(g["prompt_frames"]))
    ar = O.DualAR(weights0)
    codes = ar.generate(torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(g["src_codes"]),
                        torch.from_numpy(style), torch.from_numpy(timbre), int(g["delay"]),
                        noise_fn=lambda s: tuple(torch.from_numpy(a) for a in frame_noise(useed, s)))
    np.testing.assert_array_equal(codes.numpy(), g["codes"])
    wav = O.vocode_window(codes.long(), weights0)
    np.testing.assert_allclose(wav[0, 0, -2048:].numpy(), g["pcm_last"], atol=1e-5)


def test_offline_avg_collate_matches_reference(weights0):
    """The reference's offline infer(spk_emb_collate_type="avg") with two references (evaluations/infer_arvc.py:284-307; captured by
    tools/make_golden.py offline_avg): per-reference style / timbre embeddings averaged, code streams from the concatenated audio.  The
    oracle's restatements replay every stage: embeddings (prompt_oracle), both prompt code streams, the source codes, generate, PCM."""
    from oracle import prompt_oracle as PO
    from streamvoiceanon_amd import audio_io, specs

    g = load_golden("offline_avg_s0")
    useed = int(g["audio_seed"])
    W = {k: torch.from_numpy(sw.generate(int(g["weight_seed"]), k, shp)) for k, shp in specs.prompt_encoder_specs().items()}
    refs = [synth_utterance(int(s), int(n)) for s, n in zip(g["ref_seeds"], g["ref_samples"])]
    sv, tl = [], []
    for r in refs:
        r16 = torch.from_numpy(audio_io.resample(r, 44100, 16000))[None]
        sv.append(PO.style_vector(r16, W))
        tl.append(PO.timbre_latents(r16, W))
    style = torch.mean(torch.stack(sv, dim=0), dim=0)
    timbre = torch.mean(torch.stack(tl, dim=0), dim=0)
    np.testing.assert_allclose(style.numpy(), g["style"], atol=1e-5)
    np.testing.assert_allclose(timbre.numpy(), g["timbre"], atol=1e-5)
    ref = torch.from_numpy(np.concatenate(refs))[None]
    ac = O.firefly_encode(ref, weights0)
    np.testing.assert_array_equal(ac.numpy(), g["ref_audio_codes"])
    R = ref.shape[1] // 2048                           # (the second reference is not a whole number of frames: causal, the tail is dropped)
    cc = O.encode_window(ref[:, :R * 2048], weights0)[0, 0]
    np.testing.assert_array_equal(cc.numpy(), g["ref_content_codes"].reshape(-1))
    src = torch.from_numpy(synth_utterance(useed, int(g["src_samples"])))[None]
    sc = O.encode_window(src, weights0)[0, 0]
    np.testing.assert_array_equal(sc.numpy(), g["src_content_codes"].reshape(-1))
    ar = O.DualAR(weights0)
    codes = ar.generate(torch.from_numpy(g["ref_content_codes"].reshape(-1)), torch.from_numpy(g["ref_audio_codes"][0]).long(), sc,
                        torch.from_numpy(g["style"][0]), torch.from_numpy(g["timbre"][0]), int(g["delay"]),
                        noise_fn=lambda s: tuple(torch.from_numpy(a) for a in frame_noise(useed, s)))
    np.testing.assert_array_equal(codes.numpy(), g["codes"])
    wav = O.vocode_window(codes.long(), weights0)
    np.testing.assert_allclose(wav[0, 0, -2048:].numpy(), g["pcm_last"], atol=1e-5)


def test_prompt_codes_match_reference(weights0):
    """calculate_prompt of the reference (firefly.encode + speech_tokenizer.encode of the prompt wav), SURVEY.md §8f N1."""
    g = load_golden("prompt_s0")
    x = torch.from_numpy(synth_utterance(int(g["audio_seed"]), int(g["n_samples"])))[None]
    ac, margin = O.firefly_encode(x, weights0, return_margin=True)
    assert ac.dtype == torch.int32 and tuple(ac.shape) == (1, 8, int(g["n_samples"]) // 2048)
    np.testing.assert_array_equal(ac[0].numpy(), g["ref_audio_codes"])        # FSQ indices: bit-exact
    assert ac.min() >= 0 and ac.max() < 1000
    cc = O.encode_window(x, weights0)[0, 0]
    np.testing.assert_array_equal(cc.numpy(), g["ref_content_codes"])
    # FSQ quantise -> dequantise round trip: re-encoding the decoded latent of one group reproduces the index
    z = O.fsq_decode(ac, weights0)                                            # [1, 512, R] = project_out(code)
    assert z.shape == (1, 512, ac.shape[-1]) and float(margin.min()) > 0


def test_sampler_rules():
    # nucleus cut has NO right shift: the entry that crosses top_p is dropped too, rank 0 always kept
    logits = torch.log(torch.tensor([0.5, 0.3, 0.15, 0.05]))
    noise = torch.ones(4)
    assert O.sample_token(logits, noise, 1.0, 0.7) == 0          # cum = .5,.8 -> only rank 0 survives
    noise = torch.tensor([100.0, 1e-3, 1.0, 1.0])                # would pick 1 if it survived
    assert O.sample_token(logits, noise, 1.0, 0.7) == 0
    assert O.sample_token(logits, noise, 1.0, 0.85) == 1         # cum=.5,.8 kept


def test_sampler_edits_match_reference():
    """previous_tokens / repetition_penalty / suppress_tokens: the oracle's logits_to_probs against the reference's own function
    (fixture: tools/make_golden.py sampler_edits) -- the same support, probabilities to 1e-6, and the same sampled token."""
    g = load_golden("sampler_edits")
    for k in range(int(g["n_cases"])):
        pen, temp, top_p = [float(x) for x in g[f"params{k}"]]
        prev, sup = g[f"prev{k}"], g[f"suppress{k}"]
        logits = torch.from_numpy(g[f"logits{k}"])
        kw = dict(previous_tokens=torch.from_numpy(prev) if prev.size else None, repetition_penalty=pen,
                  suppress_tokens=[int(t) for t in sup] if sup.size else None)
        p = O.token_probs(logits, temp, top_p, **kw)
        ref = torch.from_numpy(g[f"probs{k}"])
        assert torch.equal(p > 0, ref > 0)
        assert float((p - ref).abs().max()) < 1e-6
        noise = torch.from_numpy(g[f"noise{k}"])
        assert O.sample_token(logits, noise, temp, top_p, **kw) == int(torch.argmax(ref / noise))
        assert torch.equal(logits, torch.from_numpy(g[f"logits{k}"]))          # the oracle leaves its argument alone


def test_prompt_encoders_match_reference():
    """Style (CAM++) and timbre (SparkTTS) encoders of the prompt path (SURVEY.md 8f N1 iii / iv): the oracle's restatement against
    outputs of the reference's own modules (tools/make_golden.py prompt_encoders; torchaudio's two front-ends are restatements on
    both sides -- parity unpinned for kaldi.fbank / MelSpectrogram themselves)."""
    from oracle import prompt_oracle as PO
    from streamvoiceanon_amd import specs

    g = load_golden("prompt_encoders_s0")
    W = {k: torch.from_numpy(sw.generate(int(g["weight_seed"]), k, shp)) for k, shp in specs.prompt_encoder_specs().items()}
    for tag in ("a", "b"):
        wav = torch.from_numpy(synth_utterance(int(g[f"{tag}_audio_seed"]), int(g[f"{tag}_n"])))[None]
        feat = PO.kaldi_fbank(wav)
        feat = feat - feat.mean(dim=0, keepdim=True)
        np.testing.assert_allclose(feat[7].numpy(), g[f"{tag}_feat_row7"], atol=1e-5)
        style = PO.style_vector(wav, W)
        assert tuple(style.shape) == (1, 192)
        np.testing.assert_allclose(style[0].numpy(), g[f"{tag}_style"], atol=1e-5)
        timbre = PO.timbre_latents(wav, W)
        assert tuple(timbre.shape) == (1, 32, 128)
        np.testing.assert_allclose(timbre[0].numpy(), g[f"{tag}_timbre"], atol=1e-5)
    # Kaldi mel banks: triangular, unit peak spacing, last (Nyquist) column empty
    banks = PO.kaldi_mel_banks()
    assert banks.shape == (80, 257) and float(banks[:, -1].abs().max()) == 0.0 and float(banks.max()) <= 1.0 + 1e-6


def test_long_encode_window_limited_attention(weights0):
    """Whole-utterance encode of 560 frames: beyond 512 tokens the tokenizer's transformer attends to the newest 512 keys only
    (WindowLimitedTransformer, windowed_transformer.py:291-304); codes bit-exact against the reference's."""
    g = load_golden("encoder_long_s0")
    x = torch.from_numpy(synth_utterance(int(g["audio_seed"]), int(g["n_samples"])))[None]
    codes = O.encode_window(x, weights0)
    np.testing.assert_array_equal(codes[0, 0].numpy(), g["codes"])


def test_reference_formulation_under_fp16_autocast_deviation():
    """How far the reference's own vocoder formulation moves when it runs as the reference runs it -- under torch.autocast(fp16)
    (evaluations/infer_arvc.py:493; here on the CPU, same cast rules: fp16 conv operands and fp16 activations between layers) --
    from its fp32 run: the yardstick for the GPU gate of the fp16-operand vocoder mode (tests/test_gpu_parity.py: VOC_FP16_TOL)."""
    from streamvoiceanon_amd import specs

    torch.set_grad_enabled(False)
    W = O.load_synth_weights(0, specs.all_specs())
    g = load_golden("vocoder_s0")
    codes = torch.from_numpy(g["codes"])[:, :, :24]
    ref = O.vocode_window(codes, W)[:, 0].numpy()
    with torch.autocast("cpu", dtype=torch.float16):
        out = O.vocode_window(codes, W)[:, 0].float().numpy()
    err = float(np.abs(out - ref).max())
    assert 2e-3 <= err <= 2e-2, err          # measured 3.9e-3: the fp16 path of the reference is NOT within 1e-3 of its fp32 path
