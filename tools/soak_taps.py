"""Step-by-step twin-slot comparison with taps:  python tools/soak_taps.py B steps [pipeline]   (first step at which slot s and s + B/2 differ, per tap)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
B, steps = int(sys.argv[1]), int(sys.argv[2])
pipe = len(sys.argv) > 3 and sys.argv[3] == "1"
W = {k: sw.generate(0, k, shp) for k, shp in specs.all_specs().items()}
W = {k: v for k, v in W.items() if v is not None}
eng = E.Engine(W)
half = B // 2
utts = [synth_utterance(1000 + u, 2048 * steps) for u in range(half)]
prompts = [synth_prompt(2000 + u, 107) for u in range(half)]
b = E.Batch(eng, n_streams=B, pipeline=pipe, skip_semantic=True)
for s in range(B):
    ac, cc, style, timbre = prompts[s % half]
    b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=9000 + s % half)
b.begin()
x = np.stack([utts[s % half] for s in range(B)])
c = eng.cfg
cfgs = [("content_codes", (B, 1), np.int32), ("hidden", (B, c.ar_dim), np.float32), ("fast_logits", (B, c.num_codebooks * c.codebook_size), np.float32),
        ("sampled_codes", (B, c.num_codebooks), np.int32), ("audio_codes", (B, c.num_codebooks, 1), np.int32)]
seen = {}
print("decode path", b.decode_path())
for i in range(steps):
    out = b.step(x[:, i * 2048:(i + 1) * 2048])
    for name, shp, dt in cfgs:
        try:
            t = b.tap(name, shp, dt).reshape(B, -1)
        except Exception as ex:       # noqa: BLE001
            if name not in seen: print("tap", name, "unavailable:", str(ex)[:80]); seen[name] = -2
            continue
        d = (t[:half] != t[half:]).any(axis=1)
        if d.any() and name not in seen:
            pr = int(np.argmax(d)); w = np.nonzero(t[pr] != t[pr + half])[0]
            seen[name] = i
            print(f"step {i}: tap {name}: pairs differing {np.nonzero(d)[0].tolist()}; pair {pr}: {w.size} elements differ, first idx {int(w[0])}, values {t[pr][w[0]]} vs {t[pr + half][w[0]]}", flush=True)
    d = (out[:half] != out[half:]).any(axis=1)
    if d.any() and "pcm" not in seen:
        seen["pcm"] = i; print(f"step {i}: pcm pairs differing {np.nonzero(d)[0].tolist()}", flush=True)
    if len([k for k in seen if seen[k] >= 0]) >= 4: break
b.close(); eng.close()
print("done", seen)
