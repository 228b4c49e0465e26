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
for _ in range(int(os.environ.get('DUMMY', '0'))):          # batches created and destroyed before the measured ones (stream -> HW queue mapping)
    d = E.Batch(eng, n_streams=1, pipeline=os.environ.get('PIPE', '0') == '1'); d.close()
steps = int(os.environ.get('STEPS', '60'))
S = int(os.environ.get('STREAMS', '1'))          # streams per batch
PIPE = os.environ.get('PIPE', '0') == '1'
for nb in [int(x) for x in os.environ.get('NB', '1,2,3,4').split(',')]:
    bs = []
    for i in range(nb):
        b = E.Batch(eng, n_streams=S, pipeline=PIPE, skip_semantic=os.environ.get('SKIP_SEM', '0') == '1',
                    encode_window_frames=int(os.environ.get('ENC_WIN', '128')))
        for j in range(S):
            ac, cc, st, tm = synth_prompt(2000 + i * S + j, 107)
            b.prefill_prompt(j, cc, ac, st, tm, noise_seed=i * S + j)
        b.begin()
        bs.append(b)
    audio = torch.from_numpy(np.stack([np.stack([synth_utterance(1000 + (i * S + j) % 8, 2048 * (steps + 12)) for j in range(S)]) for i in range(nb)]))
    audio = audio.reshape(nb, S, steps + 12, 2048).permute(0, 2, 1, 3).contiguous().cuda()        # [batch][chunk][S][2048]
    out = torch.empty(nb, S, 2048, device="cuda")
    torch.cuda.synchronize()
    k = 0
    for _ in range(10):
        for i, b in enumerate(bs):
            b.step_device(audio[i, k].data_ptr(), out[i].data_ptr())
        k += 1
    for b in bs: b.sync()
    if os.environ.get('PIN', '0') == '1': print('pinned to', E.pin_enqueue_thread(0)[0][:1], flush=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        for i, b in enumerate(bs):
            b.step_device(audio[i, k].data_ptr(), out[i].data_ptr())
        k += 1
    te = time.perf_counter() - t0
    for b in bs: b.sync()
    dt = time.perf_counter() - t0
    print(f"{nb} concurrent batches x {S} streams: {dt / steps * 1e3:.3f} ms per round ({nb * S * steps / dt:.1f} frames/s aggregate), host enqueue {te / steps * 1e3:.3f} ms per round", flush=True)
    for b in bs: b.close()
