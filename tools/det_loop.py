"""In-process determinism loop: the same B-stream scenario (n_sync synchronous + pipelined steps) run R times on one engine; every
run's per-step fingerprints are compared with the first run's.   DET_B=2 DET_SYNC=5 DET_R=40 python tools/det_loop.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from streamvoiceanon_amd import engine as E, specs, synth_weights
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

B = int(os.environ.get("DET_B", "2")); n_sync = int(os.environ.get("DET_SYNC", "5")); R = int(os.environ.get("DET_R", "40"))
n_steps = int(os.environ.get("DET_STEPS", "10"))
eng = E.Engine(synth_weights.generate_all(0, specs.all_specs()), ar_dtype=int(os.environ.get("DET_AR_DTYPE", "0")))
src = np.stack([synth_utterance(1000 + s, 2048 * 24)[:2048 * n_steps] for s in range(B)])
d_in = torch.from_numpy(src).cuda(); d_out = torch.empty(B, 2048, device="cuda")
prompts = [synth_prompt(2000 + s, 107) for s in range(B)]

def run():
    b = E.Batch(eng, n_streams=B, pipeline=True)
    for s in range(B):
        ac, cc, style, timbre = prompts[s]
        b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + s)
    b.begin()
    fp = []
    for i in range(n_steps):
        if i < n_sync:
            out = b.step(src[:, i * 2048:(i + 1) * 2048])
        else:
            chunk = d_in[:, i * 2048:(i + 1) * 2048].contiguous()
            torch.cuda.synchronize()             # the engine runs on its own streams: the caller's buffer must be complete before the call
            b.step_device(chunk.data_ptr(), d_out.data_ptr()); b.sync(); out = d_out.cpu().numpy()
        cc_ = b.tap("content_codes", (B, 1), np.int32).reshape(-1).tolist()
        u = b.tap("u", (B, 128, 13)) if os.environ.get("DET_U") else None
        fp.append((cc_, [hashlib.md5(np.ascontiguousarray(o).tobytes()).hexdigest()[:6] for o in out],
                   None if u is None else [hashlib.md5(np.ascontiguousarray(x[-1]).tobytes()).hexdigest()[:6] for x in u]))
    b.close()
    return fp

ref = run()
bad = 0
for r in range(1, R):
    got = run()
    if got != ref:
        bad += 1
        for i, (a, c) in enumerate(zip(ref, got)):
            if a != c:
                print(f"run {r}: first difference at step {i}: ref {a} got {c}", flush=True)
                break
print(f"B={B} n_sync={n_sync}: {bad} of {R - 1} repeated runs differ from the first")
