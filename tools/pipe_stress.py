"""Stress of the stage pipelining: serial vs pipelined stepping must agree bit for bit, repeatedly, for several batch sizes,
with re-prefills, in one process (so the GEMM tuning cache and the shared streams carry history).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sva_oracle as O
from streamvoiceanon_amd import engine as E, specs
from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

from streamvoiceanon_amd import synth_weights
W = synth_weights.generate_all(0, specs.all_specs())
eng = E.Engine(W)
n_chunks = int(os.environ.get("NCH", "60"))
reps = int(os.environ.get("REPS", "6"))
bad = 0
for B in (1, 2, 8):
    audio = torch.from_numpy(np.stack([synth_utterance(7600 + i, 2048 * n_chunks) for i in range(B)])).cuda()
    chunks = audio.reshape(B, n_chunks, 2048).transpose(0, 1).contiguous()
    torch.cuda.synchronize()

    def run(pipeline, msf):
        b = E.Batch(eng, n_streams=B, max_seq_frames=msf, buffer_frames=16, pipeline=pipeline)
        for i in range(B):
            ac, cc, style, timbre = synth_prompt(2900 + i, 40 + 9 * i)
            b.prefill_prompt(i, cc, ac, style, timbre, noise_seed=500 + i)
        b.begin()
        out = torch.zeros(n_chunks, B, 2048, device="cuda")
        torch.cuda.synchronize()
        for k in range(n_chunks):
            b.step_device(chunks[k].data_ptr(), out[k].data_ptr())
        b.sync()
        res = out.cpu().numpy(); b.close(); return res

    for msf in (768, 160):
        ref = run(False, msf)
        for r in range(reps):
            got = run(True, msf)
            if not np.array_equal(ref, got):
                d = np.abs(ref - got).reshape(n_chunks, -1).max(1)
                print(f"MISMATCH B={B} msf={msf} rep={r}: chunks {np.nonzero(d)[0][:10]} max {d.max():.3g}", flush=True)
                bad += 1
        print(f"B={B} msf={msf}: {reps} pipelined runs checked", flush=True)
print("mismatches:", bad)
