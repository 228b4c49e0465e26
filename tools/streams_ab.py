"""Pipelined throughput of small batches with and without the semantic head (skip_semantic): is the semantic-head workgroup group what
slows 2-6 streams?  python tools/streams_ab.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
eng = E.Engine(sw.generate_all(0, specs.all_specs()))
for B in (1, 2, 4, 6):
    for skip in (0, 1):
        b = E.Batch(eng, n_streams=B, pipeline=True, skip_semantic=skip)
        for s in range(B):
            ac, cc, st, tm = synth_prompt(2000 + s, 107)
            b.prefill_prompt(s, cc, ac, st, tm, noise_seed=1000 + s)
        b.begin()
        n = 130
        audio = torch.from_numpy(np.stack([synth_utterance(1000 + s, 2048 * n) for s in range(B)])).cuda().reshape(B, n, 2048).transpose(0, 1).contiguous()
        out = torch.empty(B, 2048, device="cuda")
        torch.cuda.synchronize()
        for k in range(10): b.step_device(audio[k].data_ptr(), out.data_ptr())
        b.sync()
        t0 = time.perf_counter()
        for k in range(10, 130): b.step_device(audio[k].data_ptr(), out.data_ptr())
        b.sync()
        dt = time.perf_counter() - t0
        print(f"streams {B} skip_semantic {skip}: {120 * B / dt:.0f} frames/s, {dt / 120 * 1e3:.3f} ms/step", flush=True)
        b.close()
