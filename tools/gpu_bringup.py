"""First-light diagnostics on a real MI355X: every stage of the HIP engine against the CPU oracle.
Run with:  gpurun -- 'python tools/gpu_bringup.py > gpurun_out/bringup.log 2>&1'"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)

from oracle import sva_oracle as O  # noqa: E402
from streamvoiceanon_amd import engine as E, specs  # noqa: E402
from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance  # noqa: E402

STAGES = sys.argv[1:] or ["gemm", "enc", "voc", "stream"]


def section(name):
    print("\n==== " + name + " ====", flush=True)


def run(name, fn):
    if name not in STAGES:
        return
    section(name)
    t = time.time()
    try:
        fn()
    except Exception:
        traceback.print_exc()
    print(f"[{name}] {time.time() - t:.1f}s", flush=True)


def t_gemm():
    rng = np.random.RandomState(0)
    for (M, N, K) in [(512, 512, 128), (512, 2048, 512), (100, 13, 512), (2, 2304, 768), (7, 8192, 768), (3000, 16, 48),
                      (64, 1000, 768), (2048, 32, 96), (33, 768, 128), (512, 160, 1088)]:
        A = rng.randn(M, K).astype(np.float32)
        W = rng.randn(N, K).astype(np.float32)
        b = rng.randn(N).astype(np.float32)
        out = E.test_gemm(A, W, b)
        ref = A.astype(np.float64) @ W.astype(np.float64).T + b
        err = np.abs(out - ref).max() / np.abs(ref).max()
        print(f"gemm M={M} N={N} K={K} rel_err={err:.2e}", "OK" if err < 1e-5 else "FAIL", flush=True)


W = None
eng = None


def setup():
    global W, eng
    if eng is None:
        t = time.time()
        W = O.load_synth_weights(0, specs.all_specs())
        print("weights generated", time.time() - t, flush=True)
        t = time.time()
        eng = E.Engine(W)
        print("engine finalized", time.time() - t, flush=True)


def t_enc():
    setup()
    b = E.Batch(eng, n_streams=2)
    x = np.stack([synth_utterance(1000, 262144), synth_utterance(1001, 262144)])
    t = time.time()
    codes, u = b.encode_window(x, return_u=True)
    print("encode_window wall", time.time() - t, b.timings(), b.gemm_stats(), flush=True)
    taps = {}
    ref = O.encode_window(torch.from_numpy(x), W, taps=taps)
    print("codes equal:", np.array_equal(codes, ref[0].numpy()), "mismatches", int((codes != ref[0].numpy()).sum()))
    mel = b.tap("mel", (2, 6 + 512, 160))[:, 6:]
    print("mel maxdiff", np.abs(mel - taps["mel"].transpose(1, 2).numpy()).max())
    mag = b.tap("mag", (2, 512, 1088))[:, :, :1025]
    print("mag maxdiff", np.abs(mag - O.stft_magnitude(torch.from_numpy(x)).transpose(1, 2).numpy()).max(), "mag max", mag.max())
    feat = b.tap("feat", (2, 512, 512))
    print("feat maxdiff", np.abs(feat - taps["feat"].transpose(1, 2).numpy()).max())
    print("u maxdiff", np.abs(u - taps["u"].numpy()).max(), "min|u|", np.abs(taps["u"].numpy()).min())
    for i in range(3):
        t = time.time(); b.encode_window(x); print("encode again wall", time.time() - t, b.timings()["encoder"], flush=True)
    b.close()


def t_voc():
    setup()
    for T in (4, 64):
        b = E.Batch(eng, n_streams=1, voc_max_frames=T)
        rng = np.random.RandomState(5)
        codes = rng.randint(0, 1000, size=(1, 8, T)).astype(np.int32)
        t = time.time()
        pcm = b.vocode_window(codes)
        print(f"T={T} vocode_window wall", time.time() - t, flush=True)
        ref = O.vocode_window(torch.from_numpy(codes).long(), W)[:, 0].numpy()
        print(f"T={T} pcm maxdiff", np.abs(pcm - ref).max(), "ref std", ref.std(), flush=True)
        if T == 64:
            # streaming vs windowed: feed frame by frame
            b.vocode_reset()
            outs = [b2 for b2 in []]
        b.close()
    b = E.Batch(eng, n_streams=1, voc_max_frames=1)
    rng = np.random.RandomState(5)
    codes = rng.randint(0, 1000, size=(1, 8, 64)).astype(np.int32)
    b.vocode_reset()
    t = time.time()
    outs = [b.vocode_stream(codes[:, :, i:i + 1]) for i in range(64)]
    print("64 streaming steps wall", time.time() - t)
    ref = O.vocode_window(torch.from_numpy(codes).long(), W)[:, 0].numpy()
    print("stream vs window (last frame) maxdiff", np.abs(outs[-1] - ref[:, -2048:]).max(),
          "all frames", np.abs(np.concatenate(outs, 1) - ref).max(), flush=True)
    b.close()


def t_stream():
    setup()
    useed, pseed, n_chunks = 1000, 2000, 10
    ac, cc, style, timbre = synth_prompt(pseed, 107)
    sess = O.StreamSession(W, torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(style), torch.from_numpy(timbre),
                           noise_fn=lambda f: tuple(torch.from_numpy(a) for a in frame_noise(useed, f)), delay=2)
    b = E.Batch(eng, n_streams=1)
    t = time.time()
    b.prefill_prompt(0, cc, ac, style, timbre, noise_seed=useed)
    b.begin()
    print("prefill+begin wall", time.time() - t, flush=True)
    src = synth_utterance(useed, 2048 * n_chunks)
    frame = 0
    for i in range(n_chunks):
        ch = src[i * 2048:(i + 1) * 2048]
        ref = sess.process_one_chunk(torch.from_numpy(ch)[None])[0].numpy()
        ns, nf = frame_noise(useed, frame)
        noise = np.concatenate([ns, nf.reshape(-1)])[None]
        t = time.time()
        out = b.step(ch[None], noise=noise)
        wall = time.time() - t
        cc_g = b.tap("content_codes", (1, 1), np.int32)
        msg = f"chunk {i}: wall {wall*1e3:.1f}ms {b.timings()} content {int(cc_g[0,0])} vs {int(sess.src_content_codes[-1])}"
        if sess.pred_codes.shape[1] > 0 and i >= 2:
            ag = b.tap("audio_codes", (1, 8, 1), np.int32)[0, :, 0]
            ao = sess.pred_codes[:, -1].numpy()
            msg += f" audio_eq {np.array_equal(ag, ao)} {ag.tolist()} vs {ao.tolist()}"
            frame += 1
        msg += f" pcm maxdiff {np.abs(out[0] - ref).max():.3e}"
        print(msg, flush=True)
    b.close()


run("gemm", t_gemm)
run("enc", t_enc)
run("voc", t_voc)
run("stream", t_stream)
