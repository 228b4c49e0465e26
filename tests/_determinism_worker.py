"""Worker of test_two_fresh_processes_bit_identical: one fresh process = one HIP runtime, one engine; prints digests of what
the stream_s0 configuration produces (serial steps, then pipelined device steps), to be compared across processes."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from streamvoiceanon_amd import engine as E, specs, synth_weights
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance
    import torch

    W = synth_weights.generate_all(0, specs.all_specs())
    eng = E.Engine(W)
    h_pcm, h_codes = hashlib.sha256(), hashlib.sha256()
    for B in [int(x) for x in os.environ.get("DET_B", "1,2").split(",")]:
        b = E.Batch(eng, n_streams=B, pipeline=True)
        for s in range(B):
            ac, cc, style, timbre = synth_prompt(2000 + s, 107)
            b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + s)
        b.begin()
        n_chunks = 10
        n_sync = int(os.environ.get("DET_SYNC", "5"))
        src = np.stack([synth_utterance(1000 + s, 2048 * 24)[:2048 * n_chunks] for s in range(B)])
        for i in range(n_sync):                              # caller-synchronised steps
            out = b.step(src[:, i * 2048:(i + 1) * 2048])
            h_pcm.update(np.ascontiguousarray(out).tobytes())
        d_in = torch.from_numpy(src).cuda()
        d_out = torch.empty(B, 2048, device="cuda")
        for i in range(n_sync, n_chunks):                    # overlapped (stage-pipelined) steps
            chunk = d_in[:, i * 2048:(i + 1) * 2048].contiguous()        # a torch kernel on torch's stream ...
            b.step_device_on(chunk.data_ptr(), d_out.data_ptr())        # ... ordered before the engine's read, the output before torch's copy
            h_pcm.update(d_out.cpu().numpy().tobytes())
        for s in range(B):
            h_codes.update(b.pred_codes(s).tobytes())
        b.close()
    eng.close()
    print("DIGEST", h_pcm.hexdigest(), h_codes.hexdigest())


if __name__ == "__main__":
    main()
