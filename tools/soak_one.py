"""One configuration of tools/soak_slots.py with the first differing chunk per slot pair:  python tools/soak_one.py B chunk skip ar_dtype steps [repeats]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E, specs, synth_weights as sw
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
B, chunk, skip, ar_dtype, steps = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
W = {k: sw.generate(0, k, shp) for k, shp in specs.all_specs().items()}
W = {k: v for k, v in W.items() if v is not None}
eng = E.Engine(W, ar_dtype=ar_dtype, voc_dtype=ar_dtype)
half = B // 2
utts = [synth_utterance(1000 + u, 2048 * chunk * steps) for u in range(half)]
prompts = [synth_prompt(2000 + u, 107) for u in range(half)]
prev = None
for rep in range(reps):
    b = E.Batch(eng, n_streams=B, chunk_frames=chunk, pipeline=True, skip_semantic=bool(skip))
    for s in range(B):
        ac, cc, style, timbre = prompts[s % half]
        b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=9000 + s % half)
    b.begin()
    out = b.stream_chunks(np.stack([utts[s % half] for s in range(B)]))
    path = b.decode_path()
    codes = np.stack([b.pred_codes(s_, steps * chunk - 2) for s_ in range(B)])
    b.close()
    dc = (codes[:half] != codes[half:]).any(axis=1)
    for pr in range(half):
        x0, x1 = out[pr].reshape(steps, -1), out[pr + half].reshape(steps, -1)
        bad = np.nonzero((x0 != x1).any(axis=1))[0]
        if bad.size:
            c = int(bad[0]); w = np.nonzero(x0[c] != x1[c])[0]
            print(f"   pair {pr}: chunk {c}: {w.size} samples differ, first at {int(w[0])}, last at {int(w[-1])}, max |d| {np.abs(x0[c] - x1[c]).max():.3e}, |x| max {np.abs(x0[c]).max():.3f}; "
                  f"codes differ at frames {np.nonzero(dc[pr])[0][:8].tolist()} of {codes.shape[2]}")
    d = (out[:half] != out[half:]).reshape(half, steps, -1).any(axis=2)
    first = [int(np.argmax(r)) if r.any() else -1 for r in d]
    print(f"rep {rep}: decode path {path}; slot pairs differing {int(d.any(axis=1).sum())}; first differing chunk per pair {first}", flush=True)
    if prev is not None:
        dd = (out != prev).reshape(B, steps, -1).any(axis=2)
        print(f"   vs previous repeat: slots differing {int(dd.any(axis=1).sum())}, first chunks {[int(np.argmax(r)) if r.any() else -1 for r in dd]}")
    prev = out
eng.close()
