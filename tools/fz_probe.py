import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
W = sw.generate_all(0, specs.all_specs())          # (diagnostic: which workload of the forcezero build fails)
what = sys.argv[1]
if what == "planes_hook":
    rng = np.random.default_rng(1)
    A = rng.standard_normal((256, 128)).astype(np.float32); Wt = rng.standard_normal((128, 128)).astype(np.float32)
    for mode in (0, 1, 2):
        out, _ = E.test_gemm_planes(A, Wt, mode=mode, variant=3)
        print("planes mode", mode, "ok", float(np.abs(out - A @ Wt.T).max()), flush=True)
    sys.exit(0)
mm = int(sys.argv[2]); B = int(sys.argv[3])
eng = E.Engine(W, mm_mode=mm)
print("engine ok", flush=True)
b = E.Batch(eng, n_streams=B)
for s in range(B):
    ac, cc, style, timbre = synth_prompt(2000, 60)
    b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1)
b.begin()
src = np.stack([synth_utterance(1000, 2048 * 4)] * B)
for i in range(4):
    b.step(src[:, i * 2048:(i + 1) * 2048])
print(what, "mm", mm, "B", B, "steps ok", flush=True)
