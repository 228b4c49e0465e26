"""How well do independent kernel chains overlap on this GPU?  N single-stream batches (each its own HIP streams) are
stepped round-robin from one host thread; aggregate frames/s vs one batch.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sva_oracle as O
from streamvoiceanon_amd import engine as E, specs
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

W = O.load_synth_weights(0, specs.all_specs())
eng = E.Engine(W)
steps = 60
for nb in (1, 2, 3, 4):
    bs = []
    for i in range(nb):
        b = E.Batch(eng, n_streams=1)
        ac, cc, st, tm = synth_prompt(2000 + i, 107)
        b.prefill_prompt(0, cc, ac, st, tm, noise_seed=i)
        b.begin()
        bs.append(b)
    audio = torch.from_numpy(np.stack([synth_utterance(1000 + i, 2048 * (steps + 12)) for i in range(nb)])).cuda()
    out = torch.empty(nb, 2048, device="cuda")
    k = 0
    for _ in range(10):
        for i, b in enumerate(bs):
            b.step_device(audio[i, k * 2048:(k + 1) * 2048].data_ptr(), out[i].data_ptr())
        k += 1
    for b in bs: b.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i, b in enumerate(bs):
            b.step_device(audio[i, k * 2048:(k + 1) * 2048].data_ptr(), out[i].data_ptr())
        k += 1
    te = time.perf_counter() - t0
    for b in bs: b.sync()
    dt = time.perf_counter() - t0
    print(f"{nb} concurrent batches: {dt / steps * 1e3:.3f} ms per round ({nb * steps / dt:.1f} frames/s aggregate), host enqueue {te / steps * 1e3:.3f} ms per round", flush=True)
    for b in bs: b.close()
