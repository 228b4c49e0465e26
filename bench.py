#!/usr/bin/env python
"""bench.py -- StreamVoiceAnon chunk-by-chunk infer_arvc hot path on MI355X.

A "step" is one pass of the hot path (content encoder -> dual AR -> vocoder) over one batch of
synthetic 44.1 kHz audio chunks: every stream of every rank consumes one 2048*chunk-sample chunk
and emits one converted chunk (InferenceWrapper.process_one_chunk, evaluations/infer_arvc.py:492-596).

    python bench.py --gpus 1 --steps 50 --warmup 5 [--streams 1] [--chunk 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload at N=1: BASELINE.json configs[1] -- `infer_arvc --simulate_streaming --decode_chunk_frames 1`,
delay 2, a single stream (B=1) on one MI355X.  `--streams 64` gives configs[2].  With N>1 every
rank runs the same number of independent streams (utterance-parallel, weak scaling) and the only
collective is the gather of per-utterance results + a MAX of the wall time.

Prints ONE JSON line (rank 0).  `value` = frames/s aggregated over all streams of all ranks with
inputs already resident in HBM; `rtf` = per-stream real-time factor = step time / chunk duration.
"""
from __future__ import annotations

import argparse
import json
import os

# before the HIP runtime starts (torch import): one hardware queue per engine stream, see streamvoiceanon_amd/engine.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_S = 2048 / 44100.0
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0              # HBM3E ~8 TB/s (same guide)
# unique weight elements per chunk-step: encoder 52.0 M, AR slow 92.03 + head 6.29 + fast 30.68 + fast_out 0.77 M,
# vocoder 22.35 M
ENC_PARAMS, AR_PARAMS, VOC_PARAMS = 52.0e6, 129.77e6, 22.35e6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1, help="concurrent streams per GPU (B)")
    ap.add_argument("--chunk", type=int, default=1, help="decode_chunk_frames")
    ap.add_argument("--prompt-frames", type=int, default=107)
    ap.add_argument("--ar-dtype", type=int, default=0, choices=(0, 1),
                    help="0: fp32 AR weights + fp32 KV (parity mode, the headline); 1: fp16 AR weights + fp16 slow KV cache, as the "
                         "reference decodes under torch.autocast(fp16) (evaluations/infer_arvc.py:55-59, 483)")
    ap.add_argument("--config", type=int, default=None, choices=(4, 5),
                    help="build BASELINE.json configs[3] / configs[4] exactly (1-based 4 / 5), per GPU: 4 = 64 concurrent 10 s utterances, chunk 1 "
                         "(512 over 8 GPUs); 5 = the anonymisation path, 32 utterances, chunk 4, alpha = 0.7 noise-mixed speaker embeddings of a prompt "
                         "made of three concatenated references (256 over 8 GPUs).  Sets --streams / --chunk / the prompt; --steps defaults to the "
                         "whole utterance (216 / 54 chunk-steps); combine with --gpus N")
    ap.add_argument("--mm-mode", type=int, default=None, choices=(0, 1),
                    help="sva_config.mm_mode (default: the library's): batch-scale encoder / vocoder GEMM format, csrc/gemm_planes.hip")
    ap.add_argument("--voc-dtype", type=int, default=None, choices=(0, 1),
                    help="sva_config.voc_dtype: 1 = fp16-operand vocoder GEMMs, the reference's autocast precision (infer_arvc.py:493)")
    ap.add_argument("--semantic-head", dest="skip_semantic", action="store_false",
                    help="also compute the semantic-token head and its sample in the timed run.  Every caller of the reference discards that sample "
                         "(modules/dual_ar_stream.py:833) and the engine's counter RNG makes it side-effect free, so the headline leaves it out "
                         "(sva_stream_params.skip_semantic: codes and PCM are identical either way -- tests/test_gpu_parity.py); the line also carries "
                         "the timing with the head computed (`semantic_head_on`)")
    ap.add_argument("--skip-semantic", dest="skip_semantic", action="store_true", help="(default) see --semantic-head")
    ap.set_defaults(skip_semantic=True)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=60, help="CPU-baseline sample size (chunk-steps)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the informational 64-stream run that accompanies the B=1 headline")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="do not overlap encoder / AR / vocoder of consecutive chunk-steps (default: overlapped on three streams "
                         "-- same results, it is the throughput of simulated streaming; the "
                         "latency a caller sees when it synchronises every chunk is reported as sync_latency_ms either way)")
    ap.add_argument("--no-pin", dest="pin", action="store_false",
                    help="leave the enqueueing thread where the OS put it (default: move it to the core group with the cheapest launches)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the captured single-stream hipGraph of the whole steady step (3.8 ms at B=1 against 3.6 ms for eager "
                         "serial stepping and 1.6 ms for the pipelined default, see DESIGN.md)")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--no-torch-gpu-baseline", dest="torch_gpu_baseline", action="store_false",
                    help="skip the second baseline: the reference's formulation under PyTorch-ROCm on this GPU, eager and "
                         "torch.compile(mode='reduce-overhead') (N=1 only; the compiled leg runs in a time-boxed subprocess)")
    ap.add_argument("--torch-gpu-baseline", dest="torch_gpu_baseline", action="store_true", help="(default)")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false",
                    help="skip the two rocprofv3 --pmc passes (subprocesses of this script) that measure roofline.traffic")
    ap.add_argument("--no-offline", dest="offline", action="store_false", help="skip the offline infer() block (BASELINE.json configs[0] on the GPU)")
    ap.add_argument("--compiled-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--compile-timeout", type=float, default=240.0, help="wall-clock cap (s) of the torch.compile baseline subprocess")
    ap.set_defaults(torch_gpu_baseline=True)
    return ap.parse_args()


def cpu_baseline(args, W):
    """The oracle ("port": build-owned CPU restatement in the reference's formulation -- sliding-window
    recompute of encoder AND vocoder, fp32 PyTorch-CPU) timed on this box's host cores, B=1."""
    import torch

    from oracle import sva_oracle as O
    from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance

    torch.set_grad_enabled(False)
    threads = torch.get_num_threads()
    useed = 1000
    ac, cc, style, timbre = synth_prompt(2000, args.prompt_frames)
    sess = O.StreamSession(W, torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(style), torch.from_numpy(timbre),
                           noise_fn=lambda f: tuple(torch.from_numpy(a) for a in frame_noise(useed, f)), delay=2,
                           decode_chunk_frames=args.chunk)
    n = 2048 * args.chunk
    warm = 3                        # delay warm-up chunks + one steady step (not timed)
    cands = sorted({t for t in (8, 16, 32, 64, threads) if t <= threads})
    extra = len(cands)
    src = torch.from_numpy(synth_utterance(useed, n * (warm + extra + args.cpu_steps)))[None]
    for i in range(warm):
        sess.process_one_chunk(src[:, i * n:(i + 1) * n])
    # a fair baseline: the intra-op thread count that runs this (small-op, latency-bound) workload fastest on this host,
    # not blindly every core -- one probe step per candidate, then the timed sample with the winner
    probe = {}
    for j, t in enumerate(cands):
        torch.set_num_threads(t)
        t1 = time.perf_counter()
        sess.process_one_chunk(src[:, (warm + j) * n:(warm + j + 1) * n])
        probe[t] = time.perf_counter() - t1
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    for i in range(warm + extra, warm + extra + args.cpu_steps):
        sess.process_one_chunk(src[:, i * n:(i + 1) * n])
    dt = time.perf_counter() - t0
    fps = args.cpu_steps * args.chunk / dt
    return {
        "value": round(fps, 4), "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": f"{args.cpu_steps} steady-state chunk-steps of one stream (B=1, chunk={args.chunk}, window recompute as in the reference), "
                  f"{dt:.1f} s wall, torch {torch.__version__} CPU fp32, host cpu_count={os.cpu_count()}, intra-op threads chosen by a one-step probe "
                  f"({', '.join(f'{t}: {v:.2f} s' for t, v in probe.items())})",
        "rtf": round(dt / args.cpu_steps / (args.chunk * FRAME_S), 3),
    }


def offline_cpu_baseline(W, threads):
    """BASELINE.json configs[0]'s own leg: OFFLINE infer (evaluations/infer_arvc.py:261-380) on the host cores through the oracle -- whole-utterance
    content encode, DualARWrapper.generate (prompt prefill + S decode steps), whole-utterance vocode -- for the shape the GPU `offline` block runs
    (prompt R = 168, source S = 153 frames).  A bounded sample by construction (one 7.1 s utterance, ~15-25 s of CPU work)."""
    import torch

    from oracle import sva_oracle as O
    from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance

    torch.set_grad_enabled(False)
    torch.set_num_threads(threads)
    R, S = 168, 153
    ac, cc, style, timbre = synth_prompt(2100, R)
    src = torch.from_numpy(synth_utterance(1100, 2048 * S))[None]
    t0 = time.perf_counter()
    src_codes = O.encode_window(src, W)[0, 0]
    t1 = time.perf_counter()
    ar = O.DualAR(W)
    codes = ar.generate(torch.from_numpy(cc), torch.from_numpy(ac), src_codes, torch.from_numpy(style), torch.from_numpy(timbre), 2,
                        noise_fn=lambda s_: tuple(torch.from_numpy(a) for a in frame_noise(7, s_)))
    t2 = time.perf_counter()
    wav = O.vocode_window(codes.long(), W)
    t3 = time.perf_counter()
    tot = t3 - t0
    return {"value": round(S / tot, 3), "unit": "frames/s", "cores": threads, "kind": "port", "total_ms": round(tot * 1e3, 1),
            "encode_ms": round((t1 - t0) * 1e3, 1), "generate_ms": round((t2 - t1) * 1e3, 1), "vocode_ms": round((t3 - t2) * 1e3, 1),
            "rtf": round(tot / (S * FRAME_S), 4), "pcm_samples": int(wav.numel()),
            "sample": f"one offline utterance, R = {R} / S = {S} frames ({S * FRAME_S:.2f} s of audio), oracle restatement on {threads} intra-op threads, fp32"}


def torch_gpu_baseline(args, W, steps=20):
    """Second baseline: the reference's FORMULATION run by PyTorch-ROCm eager on the same MI355X -- the oracle
    (the reference itself cannot travel to the GPU box) with its tensors on cuda:0, i.e. sliding-window recompute of encoder
    and vocoder, one aten kernel per op, fp32, no torch.compile.  Reported, never the target."""
    import torch

    from oracle import sva_oracle as O
    from streamvoiceanon_amd.synth_audio import frame_noise, synth_prompt, synth_utterance

    os.environ.setdefault("MIOPEN_LOG_LEVEL", "1")          # (its workspace warnings on every conv of the eager leg bury the JSON line in a log tail)
    os.environ.setdefault("MIOPEN_ENABLE_LOGGING", "0")
    dev = torch.device("cuda")
    Wd = {k: v.to(dev) for k, v in W.items()}
    useed = 1000
    ac, cc, style, timbre = synth_prompt(2000, args.prompt_frames)
    n = 2048 * args.chunk
    warm = 5
    with dev:                       # factory calls inside the oracle (arange, zeros, windows, masks) land on the GPU
        sess = O.StreamSession(Wd, torch.from_numpy(cc).to(dev), torch.from_numpy(ac).to(dev), torch.from_numpy(style).to(dev),
                               torch.from_numpy(timbre).to(dev),
                               noise_fn=lambda f: tuple(torch.from_numpy(a).to(dev) for a in frame_noise(useed, f)), delay=2,
                               decode_chunk_frames=args.chunk)
        src = torch.from_numpy(synth_utterance(useed, n * (warm + steps))).to(dev)[None]
        for i in range(warm):
            sess.process_one_chunk(src[:, i * n:(i + 1) * n])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warm, warm + steps):
            sess.process_one_chunk(src[:, i * n:(i + 1) * n])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": round(steps * args.chunk / dt, 3), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "kind": "port",
            "sample": f"{steps} steady chunk-steps, B=1, chunk={args.chunk}: the oracle's torch restatement of the reference's window-recompute "
                      f"formulation, eager PyTorch {torch.__version__} on cuda:0 (fp32, no torch.compile, synchronised at the ends only)",
            "rtf": round(dt / steps / (args.chunk * FRAME_S), 4)}


def pmc_traffic(B, chunk):
    """HBM bytes per conv-GEMM launch, measured NOW: two rocprofv3 --pmc passes (kernel trace + counters only, as
    MI355X_MICROARCH.md 'HBM' prescribes: separate passes; reads = 32 B x RDREQ_32B + 2 x 64 B x (RDREQ - RDREQ_32B), the x2 being the
    guide's gfx950 correction for wide coalesced requests; writes = WRITE_SIZE KiB) over a short single-stream-engine run of this
    script, reduced over the steady-state steps.  Returns (bytes per launch | None, info dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    tmp = tempfile.mkdtemp(prefix="sva_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", SVA_DEBUG="concurrency=0")
    acc = {}
    t0 = time.perf_counter()
    try:
        for name, counters in (("RD", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]), ("WR", ["WRITE_SIZE"])):
            d = os.path.join(tmp, name)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "24", "--warmup", "3", "--streams", str(B), "--chunk", str(chunk),
                   "--no-cpu-baseline", "--no-roofline", "--no-pipeline", "--no-batched", "--no-pmc", "--no-torch-gpu-baseline", "--no-offline", "--no-pin"]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"error": f"pass {name}: rc {r.returncode}, {len(files)} counter files", "stderr_tail": r.stderr.decode(errors="replace")[-300:]}
            rows = list(csv.DictReader(open(files[0])))
            # steady state: every non-pipelined step starts with one ring_write_kernel; skip prefill / delay fill / warm-up and the tail
            marks = sorted({int(x["Dispatch_Id"]) for x in rows if "ring_write" in x["Kernel_Name"]})
            lo, hi = (marks[8], marks[min(8 + 24, len(marks) - 1)]) if len(marks) > 12 else (0, 1 << 62)
            for x in rows:
                kn = x["Kernel_Name"]
                if not any(t in kn for t in ("gemm_kernel", "split_ws_kernel", "planes_dma_kernel", "voc_conv_kernel")) or not (lo <= int(x["Dispatch_Id"]) < hi):
                    continue
                a_ = acc.setdefault(x["Counter_Name"], [0, 0.0])
                a_[0] += 1
                a_[1] += float(x["Counter_Value"])
    except Exception as ex:
        return None, {"error": f"{type(ex).__name__}: {ex}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not all(k in acc and acc[k][0] for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "WRITE_SIZE")):
        return None, {"error": "counters missing from the rocprofv3 output", "have": sorted(acc)}
    rd = acc["TCC_EA0_RDREQ_sum"][1] / acc["TCC_EA0_RDREQ_sum"][0]
    rd32 = acc["TCC_EA0_RDREQ_32B_sum"][1] / acc["TCC_EA0_RDREQ_32B_sum"][0]
    wr = acc["WRITE_SIZE"][1] / acc["WRITE_SIZE"][0] * 1024.0
    reads = rd32 * 32.0 + (rd - rd32) * 64.0 * 2.0
    return reads + wr, {"read_bytes_per_launch": round(reads, 1), "write_bytes_per_launch": round(wr, 1), "launches_counted": acc["WRITE_SIZE"][0],
                        "source": "rocprofv3 --kernel-trace --pmc, two passes run by this bench.py invocation (steady-state steps of a 24-step single-queue run)",
                        "seconds": round(time.perf_counter() - t0, 1)}


class CompiledReference:
    """The reference's fast mode (modules/arvc_wrapper.py:36-41, evaluations/infer_arvc.py:128-142): torch.compile(fullgraph,
    mode='reduce-overhead') around decode_one_token_ar, firefly.head and speech_tokenizer.encode.  The reference cannot travel to the
    GPU box, so these are torch restatements in the reference's own compile-friendly form: a STATIC slow KV cache [layers, H, S, 64]
    written with index_copy_ and read in full under the causal mask of the query positions (dual_ar_stream.py:312-356, 895-936), the
    8-step fast AR with its 8-slot cache, window recompute of encoder (128 frames) and vocoder (64 frames).  Sampling = argmax(p / q)
    with Exp(1) noise drawn on the device, as multinomial_sample_one_no_sync does."""

    def __init__(self, W, dev, S=2048, fullgraph=True):
        import torch
        import torch.nn.functional as F
        from oracle import sva_oracle as O

        self.torch, self.F, self.O, self.W, self.dev = torch, F, O, W, dev
        cfg = O.ARConfig()
        self.cfg, self.hd, self.S = cfg, cfg.dim // cfg.n_head, S
        self.k = torch.zeros(cfg.n_layer, cfg.n_head, S, self.hd, device=dev)
        self.v = torch.zeros_like(self.k)
        self.tab = O.rope_table(S, self.hd).to(dev)
        self.fast_tab = O.rope_table(cfg.num_codebooks, self.hd).to(dev)
        self.fullgraph = fullgraph
        O._FB_CACHE.clear()                      # the mel filterbank as a resident device tensor (the oracle caches it per process)
        fb_key = (1025, 0.0, 22050.0, O.N_MELS, O.SR)
        fb = O.slaney_mel_fb()
        O._FB_CACHE[fb_key] = fb.to(dev)
        self.ar_step = torch.compile(self._ar_step, fullgraph=fullgraph, mode="reduce-overhead")
        self.encode = torch.compile(lambda win: O.encode_window(win, W), fullgraph=fullgraph, mode="reduce-overhead")
        self.vocode = torch.compile(lambda codes: O.vocode_window(codes, W), fullgraph=fullgraph, mode="reduce-overhead")

    def _block(self, x, p, tab, kc, vc, pos, L):
        torch, F, W, H, hd = self.torch, self.F, self.W, self.cfg.n_head, self.hd
        M = x.shape[0]
        h = self.O.rms_norm(x, W[p + "attention_norm.weight"])
        q, k, v = F.linear(h, W[p + "attention.wqkv.weight"]).split([H * hd] * 3, dim=-1)
        q = self.O.apply_rope(q.view(M, H, hd), tab[pos]).transpose(0, 1)
        k = self.O.apply_rope(k.view(M, H, hd), tab[pos]).transpose(0, 1)
        v = v.view(M, H, hd).transpose(0, 1)
        kc.index_copy_(1, pos, k)
        vc.index_copy_(1, pos, v)
        mask = torch.arange(L, device=x.device)[None, :] <= pos[:, None]
        y = F.scaled_dot_product_attention(q[None], kc[None], vc[None], attn_mask=mask[None, None])[0]
        x = x + F.linear(y.transpose(0, 1).reshape(M, H * hd), W[p + "attention.wo.weight"])
        h = self.O.rms_norm(x, W[p + "ffn_norm.weight"])
        return x + F.linear(F.silu(F.linear(h, W[p + "feed_forward.w1.weight"])) * F.linear(h, W[p + "feed_forward.w3.weight"]), W[p + "feed_forward.w2.weight"])

    def _sample(self, logits):
        torch = self.torch
        s, idx = torch.sort(logits, descending=True)
        rm = torch.cumsum(torch.softmax(s, dim=-1), dim=-1) > 0.7
        rm[0] = False
        logits = logits.masked_fill(torch.zeros_like(rm).scatter(0, idx, rm), -float("inf")) / 0.7
        return torch.argmax(torch.softmax(logits, dim=-1) / torch.empty_like(logits).exponential_(1))

    def _ar_step(self, x, pos):
        """decode_one_token_ar (dual_ar_stream.py:1168-1219): x [2, 768] at positions pos [2] -> codes [8]"""
        torch, W, cfg = self.torch, self.W, self.cfg
        for l in range(cfg.n_layer):
            x = self._block(x, f"arvc.decoder.model.layers.{l}.", self.tab, self.k[l], self.v[l], pos, self.S)
        hidden = x[-1]
        sem = self._sample(self.F.linear(self.O.rms_norm(hidden, W["arvc.decoder.model.norm.weight"]), W["arvc.decoder.model.output.weight"]))
        kc = torch.zeros(cfg.n_fast_layer, cfg.n_head, cfg.num_codebooks, self.hd, device=x.device)
        vc = torch.zeros_like(kc)
        xx, codes = hidden, []
        for cb in range(cfg.num_codebooks):
            p1 = torch.full((1,), cb, device=x.device, dtype=torch.long)
            h = xx[None]
            for l in range(cfg.n_fast_layer):
                h = self._block(h, f"arvc.decoder.model.fast_layers.{l}.", self.fast_tab, kc[l], vc[l], p1, cfg.num_codebooks)
            tok = self._sample(self.F.linear(self.O.rms_norm(h[0], W["arvc.decoder.model.fast_norm.weight"]), W["arvc.decoder.model.fast_output.weight"]))
            codes.append(tok)
            xx = W["arvc.decoder.model.fast_embeddings.weight"][tok]
        return torch.stack(codes), sem


def compiled_baseline_worker(args):
    """child process of bench.py: time the compiled formulation, print one JSON line"""
    import torch

    from oracle import sva_oracle as O
    from streamvoiceanon_amd import specs, synth_weights
    from streamvoiceanon_amd.synth_audio import synth_utterance

    torch.set_grad_enabled(False)
    dev = torch.device("cuda")
    W = {k: torch.from_numpy(v).to(dev) for k, v in synth_weights.generate_all(0, specs.all_specs()).items()}
    t_c0 = time.perf_counter()
    with dev:
        ref = CompiledReference(W, dev, fullgraph=True)
        n, steps, warm = 2048, 20, 4
        src = torch.from_numpy(synth_utterance(1000, n * (warm + steps))).to(dev)[None]
        window = torch.zeros(1, 128 * 2048, device=dev)
        pred = torch.zeros(8, 64, dtype=torch.long, device=dev)
        emb_c, emb_a = W["arvc.embedding.weight"], W["arvc.decoder.model.codebook_embeddings.weight"]
        cached = torch.zeros(1, 768, device=dev)
        pos0 = 33 + 2 * args.prompt_frames + 3
        offs = torch.arange(8, device=dev) * 1000

        def step(i, window, pred, cached):
            window = torch.cat([window[:, n:], src[:, i * n:(i + 1) * n]], dim=-1)
            code = ref.encode(window)[0, 0, -1]
            x = torch.cat([cached, emb_c[code][None]], dim=0)
            pos = torch.arange(2, device=dev) + (pos0 + 2 * i)
            codes, _ = ref.ar_step(x, pos)
            codes = codes.clone()
            cached = emb_a[codes + offs].sum(0, keepdim=True)
            pred = torch.cat([pred[:, 1:], codes[:, None]], dim=1)
            wav = ref.vocode(pred[None])
            return window, pred, cached, wav

        graph_note = "fullgraph=True"
        try:
            st_ = step(0, window, pred, cached)
        except Exception as ex:          # a graph break under fullgraph=True: the reference's flag, relaxed (reported)
            graph_note = f"fullgraph=False (fullgraph=True failed: {type(ex).__name__})"
            torch._dynamo.reset()
            ref = CompiledReference(W, dev, fullgraph=False)
            st_ = step(0, window, pred, cached)
        window, pred, cached, wav = st_
        for i in range(1, warm):
            window, pred, cached, wav = step(i, window, pred, cached)
        torch.cuda.synchronize()
        t_compile = time.perf_counter() - t_c0
        t0 = time.perf_counter()
        for i in range(warm, warm + steps):
            window, pred, cached, wav = step(i, window, pred, cached)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"value": round(steps / dt, 3), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "kind": "port",
                      "rtf": round(dt / steps / FRAME_S, 4), "compile_and_warmup_s": round(t_compile, 1),
                      "sample": f"{steps} steady chunk-steps, B=1, chunk=1: torch.compile({graph_note}, mode='reduce-overhead') around the three callables "
                                f"the reference compiles (decode_one_token_ar with a static 2048-slot KV cache, speech_tokenizer.encode on the 128-frame window, "
                                f"firefly.head(quantizer.decode) on the 64-frame window), torch {torch.__version__} on cuda:0, fp32"}))


def compiled_baseline(args):
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--compiled-baseline-worker", "--prompt-frames", str(args.prompt_frames)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.compile_timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"torch.compile baseline did not finish within {args.compile_timeout:.0f} s (Inductor / Triton compilation on ROCm)"}
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        err = [ln for ln in r.stderr.decode(errors="replace").strip().splitlines() if ln.strip() and "TORCHDYNAMO_VERBOSE" not in ln and "TORCH_LOGS" not in ln
               and "MIOpen" not in ln]
        return {"error": f"rc {r.returncode}: " + " | ".join(err[-4:])[-500:]}
    return json.loads(lines[-1])


def _wrapper_with_prompt_path():
    """the host mirror with every optional tensor loaded (firefly encoder of the prompt path, CAM++, SparkTTS speaker encoder)"""
    from streamvoiceanon_amd import specs, synth_weights
    from streamvoiceanon_amd.infer_arvc import InferenceWrapper

    W2 = synth_weights.generate_all(0, specs.all_specs(prompt_path=True))
    W2.update(synth_weights.generate_all(0, specs.prompt_encoder_specs()))
    return InferenceWrapper(weights=W2)


def offline_block(w, args):
    """BASELINE.json configs[0] on the GPU: offline InferenceWrapper.infer (evaluations/infer_arvc.py:261-380) of one utterance, the
    published example's shape (prompt R = 168 frames, source S = 153 frames): whole-utterance content encode -> DualARWrapper.generate
    (prompt prefill + S decode steps) -> whole-utterance vocode.  Each seam is a host-synchronous call; wall-clock per seam."""
    from streamvoiceanon_amd import engine as E
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    R, S = 168, 153
    ac, cc, style, timbre = synth_prompt(2100, R)
    src = synth_utterance(1100, 2048 * S)
    res = {}
    for rep in range(2):            # first pass warms allocations; the second is reported
        t0 = time.perf_counter()
        src_codes = w.encode_content(src)
        t1 = time.perf_counter()
        b = E.Batch(w.engine, n_streams=1, delay=2, voc_max_frames=S)
        t1b = time.perf_counter()
        out = b.generate(cc, ac, np.asarray(src_codes).reshape(-1), style, timbre, noise_seed=7)
        t2 = time.perf_counter()
        pcm = b.vocode_window(out[None])
        t3 = time.perf_counter()
        b.close()
        res = {"encode_ms": round((t1 - t0) * 1e3, 2), "batch_create_ms": round((t1b - t1) * 1e3, 2), "generate_ms": round((t2 - t1b) * 1e3, 2),
               "vocode_ms": round((t3 - t2) * 1e3, 2), "total_ms": round((t3 - t0) * 1e3, 2)}
    res.update({"workload": f"BASELINE.json configs[0] on the GPU: offline infer, delay=2, synthetic prompt R={R} frames, source S={S} frames "
                            f"({S * FRAME_S:.2f} s of audio), fp32",
                "frames_per_s": round(S / (res["total_ms"] * 1e-3), 1), "rtf": round(res["total_ms"] * 1e-3 / (S * FRAME_S), 5),
                "generate_ms_per_frame": round(res["generate_ms"] / S, 3), "pcm_samples": int(np.asarray(pcm).size)})
    return res


def prompt_latency_block(w):
    """Once-per-utterance latencies (evaluations/infer_arvc.py:382-441, 463-489): calculate_prompt of a reference wav (device: firefly
    encode + content encode + CAM++ style vector + SparkTTS timbre latents, incl. the 16 kHz resample on the host) and the KV prefill,
    R = 107 and 256 frames."""
    import torch

    from streamvoiceanon_amd import engine as E
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    res = {}
    for R in (107, 256):
        ac, cc, style, timbre = synth_prompt(2200 + R, R)
        b = E.Batch(w.engine, n_streams=1, chunk_frames=1, delay=2)
        v = []
        for rep in range(3):
            t0 = time.perf_counter()
            b.prefill_prompt(0, cc, ac, style, timbre, noise_seed=1)
            v.append((time.perf_counter() - t0) * 1e3)
        b.close()
        res[f"prefill_prompt_R{R}_ms"] = round(min(v), 3)
        wav = torch.from_numpy(synth_utterance(7300 + R, 2048 * R + 100))[None]
        v = []
        for rep in range(5):           # (the speaker encoders record on the first call of a length, capture on the second, replay a graph from the third)
            t0 = time.perf_counter()
            w.calculate_prompt(wav, alpha=1.0)
            v.append((time.perf_counter() - t0) * 1e3)
        res[f"calculate_prompt_R{R}_ms"] = round(min(v), 3)
        res[f"calculate_prompt_R{R}_first_call_ms"] = round(v[0], 3)
    return res


def reprefill_block(eng, B=64):
    """A whole batch re-prefilling on the same step (evaluations/infer_arvc.py:547-564; equal prompts, so every stream falls due together:
    SURVEY 8d config 3's re-prefill situation): synchronous step latency of the steady steps and of the re-prefill steps."""
    from streamvoiceanon_amd import engine as E
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    R, msf, n = 107, 160, 50
    b = E.Batch(eng, n_streams=B, max_seq_frames=msf, buffer_frames=32, pipeline=True)
    ac, cc, style, timbre = synth_prompt(2000, R)
    for s_ in range(B):
        b.prefill_prompt(s_, cc, ac, style, timbre, noise_seed=1 + s_)
    b.begin()
    src = np.stack([synth_utterance(1000 + s_ % 5, 2048 * n) for s_ in range(B)])
    lat, pos = [], []
    for i in range(n):
        t1 = time.perf_counter()
        b.step(src[:, i * 2048:(i + 1) * 2048])
        b.sync()
        lat.append((time.perf_counter() - t1) * 1e3)
        pos.append(int(b.tap("last_pos", (B,), np.int32)[0]))
    b.close()
    re = [i for i in range(1, n) if pos[i] < pos[i - 1]]
    base = float(np.median(lat[10:]))
    worst = max(lat[i] for i in re) if re else None
    return {"streams": B, "prompt_frames": R, "max_seq_frames": msf, "steady_sync_step_ms": round(base, 3), "reprefill_steps": len(re),
            "reprefill_step_ms": round(worst, 3) if worst else None, "over_steady": round(worst / base, 3) if worst else None,
            "note": "all streams due on the same step: one pass over the 2 x 32 appended rows of every stream against the cached prompt prefix, no host synchronisation"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the contract's
    torch.distributed.run line on 127.0.0.1 with a free port) and pass their exit status on.  Fails loudly when the box has fewer
    devices than ranks asked for -- a 1-rank run must never be reported as an N-GPU number."""
    import socket
    import subprocess

    import torch

    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {n_dev} HIP device(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.compiled_baseline_worker:
        return compiled_baseline_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}) or drop WORLD_SIZE and let bench.py spawn them")
    import torch
    import torch.distributed as dist

    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("SVA_FORCE_DIST") == "1"      # exercise the RCCL code path with a 1-rank group
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints banner lines ("Hostname : ...", "Librccl path : ...") on stdout when it initialises; keep stdout
        # clean for the single JSON line by pointing fd 1 at stderr until the communicator is up
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            t_ = torch.zeros(1, device="cuda")
            dist.all_reduce(t_)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # RCCL's banner sits in the C stdio buffer: flush it to stderr now
            except Exception:
                pass
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from streamvoiceanon_amd import engine as E, specs, synth_weights
    from streamvoiceanon_amd.sharding import gather_results, shard_utterances
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    if args.config == 4:
        args.streams, args.chunk = 64, 1
    elif args.config == 5:
        args.streams, args.chunk = 32, 4
    if args.config and "--steps" not in sys.argv:
        args.steps = (216 if args.config == 4 else 54) - args.warmup - 4          # 10 s = 215.3 frames; delay fill and warm-up come out of the utterance
    B, c = args.streams, args.chunk
    n = 2048 * c
    W = synth_weights.generate_all(0, specs.all_specs())
    eng = E.Engine(W, device=local_rank, ar_dtype=args.ar_dtype, mm_mode=args.mm_mode, voc_dtype=args.voc_dtype)
    pin_info = {}
    cpus_at_start = os.sched_getaffinity(0)

    def run_workload(B, steps, warmup, want_roofline, skip_semantic=None):
        """B streams per rank; returns (seconds for `steps` steps [max over ranks], stage timings, gathered count, roofline)"""
        pipelined = bool(args.pipeline) and not args.graph
        batch = E.Batch(eng, n_streams=B, chunk_frames=c, delay=2, use_graph=args.graph, pipeline=pipelined,
                        skip_semantic=args.skip_semantic if skip_semantic is None else skip_semantic)
        # utterances are global ids sharded over ranks (weak scaling: B per rank)
        my_utts = shard_utterances(list(range(world * B)), world)[rank]
        for s_, u in enumerate(my_utts):
            if args.config == 5:
                # three references concatenated (concat_mel semantics, evaluations/infer_arvc.py:413-424), speaker embeddings of the first one
                # noise-mixed with alpha = 0.7 (:228-232: alpha x + (1 - alpha)(randn std + mean)); the Gaussian is keyed by the utterance id
                parts = [synth_prompt(2000 + 10 * u + j, args.prompt_frames // 3) for j in range(3)]
                ac = np.concatenate([p_[0] for p_ in parts], axis=1)
                cc = np.concatenate([p_[1] for p_ in parts])
                gen = torch.Generator().manual_seed(500 + u)

                def mix(x, alpha=0.7):
                    t = torch.from_numpy(x)
                    return (alpha * t + (1 - alpha) * (torch.randn(t.shape, generator=gen) * t.std() + t.mean())).numpy()
                style, timbre = mix(parts[0][2]), mix(parts[0][3])
            else:
                ac, cc, style, timbre = synth_prompt(2000 + u, args.prompt_frames)
            batch.prefill_prompt(s_, cc, ac, style, timbre, noise_seed=1000 + u)
        batch.begin()
        n_delay = (2 + c - 1) // c              # chunks that only fill the delay (return zeros)
        n_lat = 40                              # synchronous per-chunk latency sample after the timed region
        prime = max(0, 4 - warmup)              # GEMM autotuning + pipeline start-up need a few steady steps: never inside the timed region
        total_chunks = n_delay + prime + warmup + steps + n_lat + 20
        audio = np.stack([synth_utterance(1000 + u, n * total_chunks) for u in my_utts])     # [B, n*total]
        d_audio = torch.from_numpy(audio).cuda().reshape(B, total_chunks, n).transpose(0, 1).contiguous()   # [chunks, B, n]
        d_out = torch.empty(B, n, device="cuda")
        torch.cuda.synchronize()

        def run(i):
            batch.step_device(d_audio[i].data_ptr(), d_out.data_ptr())

        k = 0
        for _ in range(n_delay + prime + warmup):
            run(k); k += 1
        batch.sync()
        if args.pin and not pin_info:
            # after the first steps (the runtime's helper threads exist and keep their placement): move this, the enqueueing,
            # thread to the core group with the cheapest kernel launches to this GPU -- kept only if enqueueing a real step
            # got cheaper (median of 5 steps into an idle queue), else the next-best group, else the original placement
            def enq_ms():
                nonlocal k
                v = []
                for _ in range(5):
                    batch.sync()
                    t1 = time.perf_counter()
                    run(k); k += 1
                    v.append(time.perf_counter() - t1)
                batch.sync()
                return sorted(v)[2] * 1e3
            allowed = sorted(os.sched_getaffinity(0))
            tried = [("os placement", set(allowed), enq_ms())]
            _, table = E.pin_enqueue_thread(local_rank)
            ranked = sorted(table, key=table.get)
            near = [g_ for g_ in ranked if table[g_] <= 1.1 * table[ranked[0]]] if ranked else []
            if world > 1 and near:          # ranks of one node spread over the near-best groups instead of piling onto the best one
                k0 = local_rank % len(near)
                ranked = near[k0:] + near[:k0] + [g_ for g_ in ranked if g_ not in near]
            for first in ranked[:2]:
                cpus = {c_ for c_ in allowed if first <= c_ < first + 8}
                os.sched_setaffinity(0, cpus)
                tried.append((f"cpus {first}-{first + 7}", cpus, enq_ms()))
                if tried[-1][2] < 0.9 * tried[0][2]:
                    break
            best = min(tried, key=lambda t_: t_[2])
            os.sched_setaffinity(0, best[1])
            pin_info.update({"placed_on": best[0], "step_enqueue_ms": {t_[0]: round(t_[2], 3) for t_ in tried},
                             "probe_us_per_launch_best": round(min(table.values()), 2) if table else None,
                             "probe_us_per_launch_worst": round(max(table.values()), 2) if table else None})
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run(k); k += 1
        t_enq = time.perf_counter() - t0        # host time to enqueue the K steps (the launches are asynchronous)
        batch.sync()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        # per-chunk latency as a live caller sees it (enqueue + execute + sync per chunk; outside the timed region)
        lat, enq = [], []
        for _ in range(n_lat):
            t1 = time.perf_counter()
            run(k); k += 1
            t2 = time.perf_counter()
            batch.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
            enq.append((t2 - t1) * 1e3)
        tm = batch.timings()          # of the last, individually synchronised step: its stages ran back to back, not overlapped
        lat.sort(); enq.sort()
        extra = {"host_enqueue_ms_per_step": round(t_enq / steps * 1e3, 4),          # back-to-back (includes queue back-pressure)
                 "host_enqueue_ms_idle_queue": round(enq[len(enq) // 2], 4),          # median with an empty queue: the pure host cost
                 "sync_latency_ms": {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 4),
                                     "n": n_lat}}
        dt_rank = dt
        if world > 1 or force_dist:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # this rank's own clock next to the job's (MAX over ranks): rank 0 reports every rank's N = 1-equivalent frames/s
            tl = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)] if rank == 0 else None
            dist.gather(torch.tensor([dt_rank], device="cuda", dtype=torch.float64), tl, dst=0)
            if rank == 0:
                extra["per_rank_frames_per_s"] = [round(B * c * steps / float(x.item()), 1) for x in tl]
        # the trivial gather of the per-utterance RESULTS to rank 0 (north_star configs[3]/[4]; SURVEY.md 8e: codes [8, T] per utterance):
        # every frame each utterance decoded since begin() -- delay fill, warm-up, timed and latency-sample steps
        n_res = min(batch.frames_decoded(s_) for s_ in range(B))
        if world > 1 or force_dist:           # ranks may have run a different number of placement-probe steps: gather the common length
            t = torch.tensor([n_res], device="cuda", dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_res = int(t.item())
        codes = np.stack([batch.pred_codes(s_, n_res) for s_ in range(B)])               # [B, 8, T]
        codes_dev = torch.from_numpy(codes).cuda()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gathered = gather_results(codes_dev, world, rank, force=force_dist)
        torch.cuda.synchronize()
        extra["gather_ms"] = round((time.perf_counter() - tg) * 1e3, 3)          # the job's only collective (outside the timed region): [B, 8, T] int32 per rank to rank 0
        n_gathered_frames = int(gathered.shape[0] * gathered.shape[2]) if gathered is not None else 0
        if rank == 0:
            assert gathered is not None and int(gathered.shape[0]) == world * B, f"gathered {None if gathered is None else tuple(gathered.shape)} != {world} x {B} utterances"
        roof = None
        if rank == 0 and want_roofline:
            # dominant kernel = conv_gemm_kernel (f32 MFMA): algorithmic FLOPs of all its launches in one step /
            # their summed duration, measured with hipEvents on the engine stream (single-stream profiled step)
            batch.profile_gemm(True)
            run(k); k += 1
            batch.sync()
            flops, launches = batch.gemm_stats()
            alg_bytes = batch.gemm_bytes()
            tot_ms, nl = batch.gemm_profile()
            tm_prof = batch.timings()          # stage times of this serial, event-bracketed step
            tab = batch.gemm_profile_table()
            pipes = {"f32_mfma": [0.0, 0.0, 0], "bf16_split": [0.0, 0.0, 0], "f16_weights": [0.0, 0.0, 0], "planes_bf16x3": [0.0, 0.0, 0],
                     "planes_f16x2": [0.0, 0.0, 0], "planes_f16x1": [0.0, 0.0, 0], "planes_dma_f16x2": [0.0, 0.0, 0], "planes_dma_f16x1": [0.0, 0.0, 0]}       # flops, us, launches
            KIND_PIPE = {4: "bf16_split", 5: "f16_weights", 7: "planes_f16x2", 8: "planes_f16x1", 9: "planes_dma_f16x2", 10: "planes_dma_f16x1"}
            for M_, N_, K_, taps_, mode_, us in tab:
                # 0 small-M, 1 tiled, 2 pipelined, 6 weight-streaming (gemm_stream.hip): v_mfma_f32_16x16x4_f32; 4: six bf16 part products split in the K loop; 5: fp16 weights x (hi + lo)
                # fp16 activations; 7 / 8: pre-split operand planes (gemm_planes.hip), register-staged tiles: fp16 x 2 (three products), fp16 x 1 (one);
                # 9 / 10: the same formats through the persistent LDS-DMA kernel (both operands as planes: the encoder's big GEMMs)
                kind = (int(mode_) >> 8) - 1
                pp = pipes[KIND_PIPE.get(kind, "f32_mfma")]
                pp[0] += 2.0 * M_ * N_ * K_; pp[1] += us; pp[2] += 1
            if os.environ.get("SVA_GEMM_TABLE"):
                agg = {}
                for M_, N_, K_, taps_, mode_, us in tab:
                    key = (int(M_), int(N_), int(K_), int(taps_), int(mode_) & 255, (int(mode_) >> 8) - 1)
                    a_ = agg.setdefault(key, [0, 0.0])
                    a_[0] += 1; a_[1] += us
                with open(os.environ["SVA_GEMM_TABLE"] + (f".b{B}" if B != args.streams else ""), "w") as f:
                    f.write("M,N,K,taps,mode,kernel_kind,calls,total_us,avg_us,TFLOPs\n")
                    for key, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                        fl = 2.0 * key[0] * key[1] * key[2] * cnt
                        f.write(",".join(map(str, key)) + f",{cnt},{us:.1f},{us / cnt:.2f},{fl / us / 1e6:.2f}\n")
            batch.profile_gemm(False)
            ach = flops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            # HBM bytes per conv-GEMM launch: measured by this invocation (two rocprofv3 --pmc child runs) when B is the headline
            # workload; the committed profile's number is reported under its own name, never as `traffic`
            import glob
            traffic, pmc_info = None, None
            if args.pmc and world == 1 and (B == args.streams or (B == 64 and args.streams == 1)):
                batch.sync()
                traffic, pmc_info = pmc_traffic(B, c)
            prof_stages = None
            prof_traffic, mfma_util, cands = None, None, sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_b{B}.json")))
            if cands and c == 1:
                pj = json.load(open(cands[-1]))
                prof_traffic = round(pj["hbm_bytes_per_launch"], 1)
                mfma_util = round(pj["gemm_mfma_util"], 4) if pj.get("gemm_mfma_util") is not None else None
                prof_stages = {k_: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v_.items()} for k_, v_ in (pj.get("stages") or {}).items()} or None
            alg_per_launch = alg_bytes / max(nl, 1)
            PEAK_SPLIT = 2500.0 / 6.0
            PEAK_F16W = 2500.0 / 2.0          # v_mfma_f32_16x16x32_f16, two part products (activation hi, lo) per weight block
            by_pipe = {}
            for name, (fl, us, cnt) in pipes.items():
                if cnt:
                    pk = {"f32_mfma": PEAK_F32_MFMA_TFLOPS, "bf16_split": PEAK_SPLIT, "planes_bf16x3": PEAK_SPLIT, "planes_f16x2": 2500.0 / 3.0,
                          "planes_f16x1": 2500.0, "planes_dma_f16x2": 2500.0 / 3.0, "planes_dma_f16x1": 2500.0, "f16_weights": PEAK_F16W}[name]
                    by_pipe[name] = {"launches": cnt, "ms": round(us * 1e-3, 4), "gflop": round(fl / 1e9, 3), "achieved": round(fl / us / 1e6, 3),
                                     "peak": round(pk, 1), "frac": round(fl / us / 1e6 / pk, 5)}
            # the fraction of what the launches COULD have done in their own time on the pipes they ran on
            cap = sum(v["ms"] * v["peak"] for v in by_pipe.values())
            frac_own = (flops / 1e9) / cap if cap > 0 else 0.0
            f32_only = set(by_pipe) <= {"f32_mfma"}
            roof = {"bound": "mfma", "kernel": "conv-GEMM family: pipe_gemm_kernel / conv_gemm_kernel / skinny_gemm_kernel (v_mfma_f32_16x16x4_f32) and split_gemm_kernel / "
                              "planes_gemm_kernel / planes_dma_kernel (the same fp32 problems on the 16-bit pipes: six bf16 part products split in the K loop, or "
                              "three fp16 part products from pre-split operand planes with sva_config.mm_mode = 1 -- fp32-grade results either way; one fp16 "
                              "product for a voc_dtype = 1 vocoder; planes_dma = the persistent LDS-DMA form the encoder's batch-scale GEMMs and, from 16 code "
                              "frames per step, the HiFiGAN ResBlock convs run in (incl. voc_conv_kernel for the C = 16 / 32 levels); the per-shape table picks)",
                    "peak_note": "one stream: peak = 157.3 TF/s, the dense f32-MFMA peak, `frac` = achieved / 157.3.  Batch scale: launches of the split-bf16 kernel run on "
                                 "the bf16 pipes (ceiling for this work 2500 / 6 = 416.7 TF/s), the planes kernels on the fp16 pipes (2500 / 3 = 833.3 for three part "
                                 "products): `by_pipe` prices each kernel family against its own pipe, `peak` is their time-weighted mean and `frac` = "
                                 "`frac_of_own_pipes`; an ar_dtype = 1 batch adds `f16_weights` (gemm_f16w.hip: fp16 weights, activations as hi + lo fp16 "
                                 "parts, ceiling 2500 / 2 TF/s -- those launches are decode-sized and bound by their weight stream, not by the pipe)",
                    "achieved": round(ach, 3),
                    # `frac` = achieved / peak OF THE PIPES THE LAUNCHES RAN ON: at one stream every launch is v_mfma_f32_16x16x4_f32 (peak 157.3);
                    # at batch scale most of the work runs as fp16 part products, and dividing that by the f32-MFMA peak gave a "fraction" near or
                    # above 1 -- there `peak` is the time-weighted peak of the pipes used and `frac` = frac_of_own_pipes
                    "peak": PEAK_F32_MFMA_TFLOPS if f32_only else round(cap / max(sum(v["ms"] for v in by_pipe.values()), 1e-9), 1), "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 5) if f32_only else round(frac_own, 5), "all_launches_on_f32_mfma": f32_only,
                    "arithmetic": "f32 (v_mfma_f32_16x16x4_f32) in every launch" if f32_only else
                                  "f32-grade: encoder / vocoder GEMMs from 10 streams as fp16 x 3 part products of exactly split fp32 operands (mm_mode 1; fp16 range, "
                                  "absolute error floor 2^-25 for small activations), f32 MFMA for the AR chain and the narrow layers",
                    "frac_of_own_pipes": round(frac_own, 5), "by_pipe": by_pipe,
                    "mode": "one serial step, every conv-GEMM launch bracketed by hipEvents on its launch stream (launches do not overlap)",
                    "traffic": round(traffic, 1) if traffic is not None else None, "traffic_measurement": pmc_info,
                    "traffic_from_committed_profile": prof_traffic, "committed_profile": os.path.basename(cands[-1]) if prof_traffic is not None else None,
                    "algorithmic_bytes_per_launch": round(alg_per_launch, 1),
                    "traffic_over_algorithmic": round(traffic / alg_per_launch, 3) if traffic else None,
                    "stages_pmc_committed_profile": prof_stages,     # per stage: achieved fabric-side GB/s, fraction of 8 TB/s, MFMA utilisation (tools/pmc.sh -> tools/pmc_agg.py)
                    "mfma_util_pmc_committed_profile": mfma_util,      # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024) over the same kernels (tools/pmc.sh)
                    "launches_per_step": int(nl), "avg_launch_us": round(tot_ms * 1e3 / max(nl, 1), 3),
                    "algorithmic_gflop_per_step": round(flops / 1e9, 3), "gemm_ms_per_step": round(tot_ms, 4),
                    "stage_ms_profiled_step": {k_: round(v, 4) for k_, v in tm_prof.items()}}
        batch.close()
        extra["gathered_frames"] = n_gathered_frames
        return dt, tm, (int(gathered.shape[0]) if gathered is not None else B), roof, extra

    dt, tm, n_gathered, roof, extra = run_workload(B, args.steps, args.warmup, not args.no_roofline)
    if rank != 0:
        if world > 1 or force_dist:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    fps = world * B * c * args.steps / dt
    out = {
        "metric": "aggregate converted frames/s (2048-sample frames @44.1 kHz), chunk-by-chunk streaming infer_arvc",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if not args.ar_dtype else "f32 (encoder, vocoder) / f16 weights + f16 KV, f32 accumulate (AR)", "data": "synthetic",
        "config": {"workload": ("" if not args.config else f"BASELINE.json configs[{args.config - 1}] per-GPU shape" + (", alpha = 0.7, three concatenated references" if args.config == 5 else "") + ": ") +
                               f"infer_arvc --simulate_streaming --decode_chunk_frames {c}, delay=2, {B} stream(s) per GPU, "
                               f"encode window 128 / vocoder window 64 frames, synthetic 44.1 kHz speech-like audio, "
                               f"synthetic prompt R={args.prompt_frames}, random-init weights of the reference architecture",
                   "streams_per_gpu": B, "chunk_frames": c, "parallelism": f"utterance-parallel x{world}",
                   "hipgraph": ("whole step as one captured graph" if args.graph else
                                "one captured graph per stage chain (encoder front, side chain, AR, vocoder), replayed every step" if args.pipeline else "none (eager launches)"),
                   "stage_pipelining": bool(args.pipeline) and not args.graph, "mm_mode": int(eng.cfg.mm_mode), "voc_dtype": int(eng.cfg.voc_dtype), "skip_semantic_head": bool(args.skip_semantic),
                   "enqueue_thread": pin_info or None},
        "rtf": round(ms * 1e-3 / (c * FRAME_S), 5), "x_realtime": round(fps * FRAME_S, 2),
        # what a caller that waits for every chunk sees (the reference's process_one_chunk contract): median synchronous latency / chunk duration
        "rtf_live": round(extra["sync_latency_ms"]["p50"] * 1e-3 / (c * FRAME_S), 5), "sync_latency_p50_ms": extra["sync_latency_ms"]["p50"],
        "stage_ms_last_step": {k_: round(v, 4) for k_, v in tm.items()},
        "gathered_utterances": n_gathered,
    }
    out.update(extra)

    def stage_rates(B_, ms_step, tm_, gflop_step, f32_only=True):
        """Rates of the TIMED (pipelined) configuration and of the stages of a synchronised step: the conv-GEMM FLOPs of a step
        are the encoder's and the vocoder's (+ the AR's at B > 2); the AR at B <= 2 is weight streaming."""
        wb = 2 if args.ar_dtype else 4
        p_tok = 33 + 2 * args.prompt_frames + 3 + 2 * 60                     # a typical slow-AR position during the timed region
        kv_bytes = 2 * 12 * (p_tok + 2) * 768 * wb * 2 * c                   # K and V of 12 layers, read once per decoded frame (two query rows share it)
        ar_bytes = AR_PARAMS * wb * c + B_ * kv_bytes
        weight_bytes = ENC_PARAMS * 4 + VOC_PARAMS * 4 + AR_PARAMS * wb * c
        enc_gflop = 13.9 * B_                                                # merged incremental pass: head 160 + 6 + 4c rows, 128-token transformer
        voc_gflop = 2.646 * c * B_
        r = {"timed": {"note": "the K timed steps as run (stages of consecutive steps overlapped): algorithmic conv-GEMM FLOPs of a step / ms_per_step", "ms_per_step": round(ms_step, 4), "algorithmic_gflop_per_step": round(gflop_step, 3),
                       "tflops": round(gflop_step / ms_step, 3), **({"frac_f32_mfma": round(gflop_step / ms_step / PEAK_F32_MFMA_TFLOPS, 5)} if f32_only else {}),
                       "unique_weight_bytes_per_step": int(weight_bytes), "weight_stream_GBs": round(weight_bytes / ms_step / 1e6, 1),
                       "frac_hbm": round(weight_bytes / ms_step / 1e6 / PEAK_HBM_GBS, 5)},
             "stages_synchronised_step": {
                 "encoder": {"ms": round(tm_["encoder"], 4), "gflop": round(enc_gflop, 2), "tflops": round(enc_gflop / max(tm_["encoder"], 1e-9), 2),
                             **({"frac_f32_mfma": round(enc_gflop / max(tm_["encoder"], 1e-9) / PEAK_F32_MFMA_TFLOPS, 5)} if f32_only else {})},
                 "ar": {"ms": round(tm_["ar"], 4), "bytes": int(ar_bytes), "GBs": round(ar_bytes / max(tm_["ar"], 1e-9) / 1e6, 1),
                        "frac_hbm": round(ar_bytes / max(tm_["ar"], 1e-9) / 1e6 / PEAK_HBM_GBS, 5),
                        "note": "unique AR weight bytes (fast layers counted once per codebook pass: 8 x 30.7 M elements stream from L2 / MALL) + slow KV read"},
                 "vocoder": {"ms": round(tm_["vocoder"], 4), "gflop": round(voc_gflop, 2), "tflops": round(voc_gflop / max(tm_["vocoder"], 1e-9), 2),
                             **({"frac_f32_mfma": round(voc_gflop / max(tm_["vocoder"], 1e-9) / PEAK_F32_MFMA_TFLOPS, 5)} if f32_only else {})}}}
        return r

    if roof:
        roof.update(stage_rates(B, ms, tm, roof["algorithmic_gflop_per_step"], roof["all_launches_on_f32_mfma"]))
        out["roofline"] = roof
    if world == 1 and args.skip_semantic and not args.no_batched and not args.config:
        # the same timed run with the semantic-token head and its (discarded) sample computed, as the reference does (dual_ar_stream.py:1181-1186, 833)
        dt_s, _, _, _, ex_s = run_workload(B, args.steps, args.warmup, False, skip_semantic=False)
        out["semantic_head_on"] = {"ms_per_step": round(dt_s / args.steps * 1e3, 4), "value": round(world * B * c * args.steps / dt_s, 3), "unit": "frames/s",
                                   "sync_latency_p50_ms": ex_s["sync_latency_ms"]["p50"],
                                   "note": "the headline leaves the semantic head out (its sample is discarded by every caller; codes and PCM are bit-identical: "
                                           "tests/test_gpu_parity.py::test_skip_semantic_head_changes_nothing_downstream)"}
    if world == 1 and B == 1 and not args.no_batched:
        # BASELINE.json configs[2] next to the headline single-stream workload: 64 concurrent streams on the same GPU
        # (the "frames/sec aggregate" half of the metric); informational, `value` above stays the configs[1] number
        dt2, tm2, _, roof2, extra2 = run_workload(64, 10, 3, not args.no_roofline)
        ms2 = dt2 / 10 * 1e3
        out["batched_64_streams"] = {"workload": "BASELINE.json configs[2]: 64 concurrent streams, chunk=1, 1 GPU", "ms_per_step": round(ms2, 4),
                                     "value": round(64 * c * 10 / dt2, 3), "unit": "frames/s", "rtf": round(ms2 * 1e-3 / (c * FRAME_S), 5),
                                     "x_realtime": round(64 * c * 10 / dt2 * FRAME_S, 2),
                                     "stage_ms_last_step": {k_: round(v, 4) for k_, v in tm2.items()},
                                     "roofline": ({**roof2, **stage_rates(64, ms2, tm2, roof2["algorithmic_gflop_per_step"], roof2["all_launches_on_f32_mfma"])} if roof2 else None), **extra2}
    if world == 1 and B == 1 and not args.no_batched:
        try:
            out["reprefill_64_streams"] = reprefill_block(eng)
        except Exception as ex:
            out["reprefill_64_streams"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if world == 1 and args.offline:
        import traceback
        wrap = None
        try:
            wrap = _wrapper_with_prompt_path()
            out["offline"] = offline_block(wrap, args)
        except Exception as ex:
            out["offline"] = {"error": f"{type(ex).__name__}: {ex}"[:300], "where": traceback.format_exc().strip().splitlines()[-3][:200]}
        try:
            if wrap is not None:
                out["prompt_latency"] = prompt_latency_block(wrap)
        except Exception as ex:
            out["prompt_latency"] = {"error": f"{type(ex).__name__}: {ex}"[:300], "where": traceback.format_exc().strip().splitlines()[-3][:200]}
        if wrap is not None:
            wrap.engine.close()
    if world == 1 and not args.no_cpu_baseline:
        os.sched_setaffinity(0, cpus_at_start)      # the CPU leg uses all host cores again
        import torch as _t
        Wt = {k: _t.from_numpy(v) for k, v in W.items()}
        out["cpu_baseline"] = cpu_baseline(args, Wt)
        if args.offline and isinstance(out.get("offline"), dict) and "error" not in out["offline"]:
            try:
                out["offline"]["cpu_baseline"] = offline_cpu_baseline(Wt, out["cpu_baseline"]["cores"])
                out["offline_vs_cpu"] = round(out["offline"]["cpu_baseline"]["total_ms"] / out["offline"]["total_ms"], 1)
            except Exception as ex:
                out["offline"]["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if world == 1 and args.torch_gpu_baseline:
        import torch as _t
        eng.close()
        os.sched_setaffinity(0, cpus_at_start)
        try:
            from oracle import sva_oracle as _O
            _O._FB_CACHE.clear()             # (the oracle caches its mel filterbank on the device it was first built on)
            out["torch_gpu_baseline"] = torch_gpu_baseline(args, {k: _t.from_numpy(v) for k, v in W.items()})
            _O._FB_CACHE.clear()
        except Exception as ex:          # a reported extra, never a reason to lose the bench line
            out["torch_gpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        _t.cuda.empty_cache()
        out["torch_compile_gpu_baseline"] = compiled_baseline(args)
    if "batched_64_streams" in out:
        # compact configs[2] figures as the LAST key of the line (a log tail keeps them)
        b64 = out["batched_64_streams"]
        r64 = b64.get("roofline") or {}
        out["b64"] = {"ms_per_step": b64["ms_per_step"], "value": b64["value"], "unit": "frames/s", "frac_of_own_pipes": r64.get("frac_of_own_pipes"),
                      "stage_ms": b64["stage_ms_last_step"]}
        # ... and flat scalar keys (a parser that keeps top-level scalars keeps these)
        out["b64_frames_per_s"], out["b64_ms_per_step"], out["b64_frac_of_own_pipes"] = b64["value"], b64["ms_per_step"], r64.get("frac_of_own_pipes")
        for k_, v_ in b64["stage_ms_last_step"].items():
            out[f"b64_{k_}_ms"] = v_
        dma = (r64.get("by_pipe") or {}).get("planes_dma_f16x2")
        if dma:          # the encoder's batch-scale GEMMs (persistent LDS-DMA planes kernel): algorithmic TF/s and fraction of the 833 TF/s three-product ceiling
            out["b64_planes_dma_tflops"], out["b64_planes_dma_frac"] = dma["achieved"], dma["frac"]
    print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
