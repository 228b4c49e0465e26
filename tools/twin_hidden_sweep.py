"""Twin-slot hidden-state sweep over batch sizes and AR precisions (slot s and s + B/2 carry the same stream): python tools/twin_hidden_sweep.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
W = {k: sw.generate(0, k, shp) for k, shp in specs.all_specs().items()}
W = {k: v for k, v in W.items() if v is not None}
for ar_dtype in (1, 0):
    eng = E.Engine(W, ar_dtype=ar_dtype); c = eng.cfg
    for B in (4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128):
        half, steps = B // 2, 4
        utts = [synth_utterance(3100 + u, 2048 * steps) for u in range(half)]
        prompts = [synth_prompt(3200 + u, 107) for u in range(half)]
        b = E.Batch(eng, n_streams=B, skip_semantic=True)
        for s_ in range(B):
            ac, cc, style, timbre = prompts[s_ % half]
            b.prefill_prompt(s_, cc, ac, style, timbre, noise_seed=700 + s_ % half)
        b.begin()
        x = np.stack([utts[s_ % half] for s_ in range(B)])
        res = "identical"
        for i in range(steps):
            b.step(x[:, i * 2048:(i + 1) * 2048])
            t = b.tap("hidden", (B, c.ar_dim), np.float32)
            d = (t[:half] != t[half:]).any(axis=1)
            if d.any():
                res = f"hidden differs at step {i} in pairs {np.nonzero(d)[0].tolist()}"; break
        print(f"ar_dtype {ar_dtype} streams {B:2d} path {b.decode_path()}: {res}", flush=True)
        b.close()
    eng.close()
