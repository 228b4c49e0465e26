"""Worker of test_rccl_gather_of_engine_results: one rank of a torch.distributed ("nccl" = RCCL) group.  Each rank runs its shard of
the utterances through the real engine, the per-utterance codes are gathered to rank 0 with sharding.gather_results (dist.gather; at
world 1 the forced one-rank group still goes through RCCL), and rank 0 prints a digest of the gathered codes in global utterance order
next to the digest of the same utterances decoded in ONE process without any communication."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def decode(eng, utts, n_chunks):
    from streamvoiceanon_amd import engine as E
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    B = len(utts)
    b = E.Batch(eng, n_streams=B, pipeline=True)
    for s, u in enumerate(utts):
        ac, cc, style, timbre = synth_prompt(2000 + u % 3, 60 + 10 * (u % 3))
        b.prefill_prompt(s, cc, ac, style, timbre, noise_seed=1000 + u)          # noise keyed by utterance id, never by rank or slot
    b.begin()
    src = np.stack([synth_utterance(1000 + u, 2048 * n_chunks) for u in utts])
    for i in range(n_chunks):
        b.step(src[:, i * 2048:(i + 1) * 2048])
    n = min(b.frames_decoded(s) for s in range(B))
    codes = np.stack([b.pred_codes(s, n) for s in range(B)])          # [B, 8, T]
    b.close()
    return codes


def main():
    import torch
    import torch.distributed as dist

    from streamvoiceanon_amd import engine as E, sharding, specs, synth_weights

    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    n_utt, n_chunks = 4 * world, 8
    parts = sharding.shard_utterances(list(range(n_utt)), world)
    W = synth_weights.generate_all(0, specs.all_specs())
    eng = E.Engine(W, device=local)
    mine = decode(eng, parts[rank], n_chunks)
    gathered = sharding.gather_results(torch.from_numpy(mine).cuda(), world, rank, force=True)
    torch.cuda.synchronize()
    if rank == 0:
        g = gathered.cpu().numpy()[sharding.unshard(parts)]                # rank-major -> global utterance order
        # the same utterances, one process, no communication (in the ranks' batch shapes, so that both runs take the same kernels)
        solo = np.concatenate([decode(eng, parts[r], n_chunks) for r in range(world)])[sharding.unshard(parts)]
        print("GATHER", world, g.shape, hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest(),
              hashlib.sha256(np.ascontiguousarray(solo).tobytes()).hexdigest(), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
