import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sva_oracle as O
from streamvoiceanon_amd import engine as E, specs
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
W = O.load_synth_weights(0, specs.all_specs())
eng = E.Engine(W)
n_chunks, B = int(os.environ.get("NCH", "40")), 2
msf = int(os.environ.get("MSF", "768"))
audio = torch.from_numpy(np.stack([synth_utterance(7600 + i, 2048 * n_chunks) for i in range(B)])).cuda()
chunks = audio.reshape(B, n_chunks, 2048).transpose(0, 1).contiguous()      # persistent [chunk][B][2048]: the engine reads it asynchronously
torch.cuda.synchronize()
def run(pipeline):
    b = E.Batch(eng, n_streams=B, max_seq_frames=msf, buffer_frames=16, pipeline=pipeline)
    for i in range(B):
        ac, cc, style, timbre = synth_prompt(2900 + i, 40 + 30 * i)
        b.prefill_prompt(i, cc, ac, style, timbre, noise_seed=500 + i)
    b.begin()
    out = torch.zeros(n_chunks, B, 2048, device="cuda")
    for k in range(n_chunks):
        b.step_device(chunks[k].data_ptr(), out[k].data_ptr())
    b.sync()
    res = out.cpu().numpy(); b.close(); return res
s = run(False); p = run(True); s2 = run(False)
print("serial vs serial maxdiff", np.abs(s - s2).max())
for k in range(n_chunks):
    d = np.abs(s[k] - p[k]).max(axis=1)
    if d.max() > 0: print("chunk", k, "diff per slot", d)
print("done")
