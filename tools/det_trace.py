"""Per-step fingerprints of the determinism scenario (B = 1 batch, then a B = 2 batch with 5 synchronous + 5 pipelined steps): content
codes, audio codes and a PCM checksum per slot and step -- run repeatedly to find WHERE two processes first disagree."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from streamvoiceanon_amd import engine as E, specs, synth_weights
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

W = synth_weights.generate_all(0, specs.all_specs())
eng = E.Engine(W)
out_lines = []
for B in (1, 2):
    b = E.Batch(eng, n_streams=B, pipeline=True)
    for s in range(B):
        ac, cc, style, timbre = synth_prompt(2000 + s, 107)
        b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + s)
    b.begin()
    src = np.stack([synth_utterance(1000 + s, 2048 * 24)[:2048 * 10] for s in range(B)])
    d_in = torch.from_numpy(src).cuda(); d_out = torch.empty(B, 2048, device="cuda")
    for i in range(10):
        if i < 5:
            out = b.step(src[:, i * 2048:(i + 1) * 2048])
        else:
            chunk = d_in[:, i * 2048:(i + 1) * 2048].contiguous()
            torch.cuda.synchronize()             # the engine runs on its own streams: the caller's buffer must be complete before the call
            b.step_device(chunk.data_ptr(), d_out.data_ptr()); b.sync(); out = d_out.cpu().numpy()
        cc_ = b.tap("content_codes", (B, 1), np.int32).reshape(-1)
        ac_ = b.tap("audio_codes", (B, 8, 1), np.int32).reshape(B, 8)
        hid = b.tap("hidden", (B, 768))
        out_lines.append(f"B{B} step{i} content {cc_.tolist()} audio {ac_.tolist()} hid {[hashlib.md5(h.tobytes()).hexdigest()[:6] for h in hid]} pcm {[hashlib.md5(np.ascontiguousarray(o).tobytes()).hexdigest()[:6] for o in out]}")
    b.close()
print("\n".join(out_lines))
