"""Checksums of engine outputs over a spread of workloads (tools/waitcnt_audit.sh runs it against two builds of the library).
Everything printed is a function of the computed values only (no timings)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw  # noqa: E402
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance  # noqa: E402


def h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


W = sw.generate_all(0, specs.all_specs(prompt_path=True))
for ar_dtype in (0, 1):
    eng = E.Engine(W, ar_dtype=ar_dtype)
    for B, chunk, steps, R in ((1, 1, 14, 107), (2, 1, 8, 60), (8, 1, 6, 60), (12, 1, 6, 60), (16, 1, 5, 60), (24, 1, 4, 60), (16, 4, 3, 80), (64, 1, 4, 107)):
        b = E.Batch(eng, n_streams=B, chunk_frames=chunk)
        for s in range(B):
            ac, cc, style, timbre = synth_prompt(2000 + s % 3, R)
            b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + s)
        b.begin()
        n = 2048 * chunk
        srcs = [synth_utterance(1000 + s % 5, n * steps) for s in range(B)]
        pcm = []
        for i in range(steps):
            pcm.append(b.step(np.stack([x[i * n:(i + 1) * n] for x in srcs])))
        codes = np.stack([b.pred_codes(s) for s in range(B)])
        print(f"ar_dtype={ar_dtype} B={B} chunk={chunk} decode_path={b.decode_path()}: codes {h(codes)} pcm {h(np.stack(pcm))} content {h(b.tap('content_codes', (B, chunk), np.int32))}", flush=True)
        b.close()
    # offline path + whole-utterance seams + re-prefill inside a stream
    ac, cc, style, timbre = synth_prompt(2100, 168)
    b = E.Batch(eng, n_streams=1, delay=2, voc_max_frames=40)
    src_codes = (np.arange(40, dtype=np.int64) * 2654435761 % 8192).astype(np.int64)
    codes = b.generate(cc, ac, src_codes, style, timbre, noise_seed=5)
    print(f"ar_dtype={ar_dtype} offline: codes {h(codes)} pcm {h(b.vocode_window(codes[None]))}", flush=True)
    b.close()
    b = E.Batch(eng, n_streams=1, max_seq_frames=136, buffer_frames=32)
    ac, cc, style, timbre = synth_prompt(2001, 60)
    b.prefill_prompt(0, cc, ac, style, timbre, noise_seed=7)
    b.begin()
    src = synth_utterance(1001, 2048 * 40)
    pcm = [b.step(src[None, i * 2048:(i + 1) * 2048]) for i in range(40)]
    print(f"ar_dtype={ar_dtype} re-prefill stream: codes {h(b.pred_codes(0))} pcm {h(np.stack(pcm))}", flush=True)
    b.close()
    wb = E.Batch(eng, n_streams=1, encode_window_frames=300)
    wav = synth_utterance(1200, 2048 * 300)
    print(f"ar_dtype={ar_dtype} long encode {h(wb.encode_window(wav[None]))} firefly.encode {h(wb.firefly_encode(wav[None]))}", flush=True)
    wb.close()
    eng.close()

# round 4: the fp16-operand vocoder (gemm_planes.hip, one plane), the range-safe bf16 kernels at batch scale, a whole batch re-prefilling
# in one pass
lib = E.load_library()
cases = [("voc_dtype=1", dict(voc_dtype=1)), ("mm_mode=0", dict(mm_mode=0))]
for tag, kw in cases:
    eng = E.Engine(W, **kw)
    for B, steps, msf in ((64, 4, 768), (3, 6, 768), (16, 40, 100)):
        b = E.Batch(eng, n_streams=B, max_seq_frames=msf, buffer_frames=32)
        for s in range(B):
            ac, cc, style, timbre = synth_prompt(2000 + s % 3, 60)
            b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + s)
        b.begin()
        srcs = [synth_utterance(1000 + s % 5, 2048 * steps) for s in range(B)]
        pcm = [b.step(np.stack([x[i * 2048:(i + 1) * 2048] for x in srcs])) for i in range(steps)]
        codes = np.stack([b.pred_codes(s) for s in range(B)])
        print(f"{tag} B={B} steps={steps}: codes {h(codes)} pcm {h(np.stack(pcm))}", flush=True)
        b.close()
    eng.close()
