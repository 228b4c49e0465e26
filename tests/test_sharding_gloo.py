"""N>1 path on CPU: world_size-2 gloo processes shard utterances, run the per-utterance work with
the CPU oracle's AR sampler (noise keyed by utterance id), gather to rank 0, and the result must be
identical to the single-process run."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _work(utt):
    """cheap deterministic per-utterance result keyed ONLY by the utterance id"""
    sys.path.insert(0, ROOT)
    from oracle import sva_oracle as O
    from streamvoiceanon_amd.synth_audio import frame_noise

    rng = np.random.RandomState(utt)
    logits = torch.from_numpy(rng.randn(8, 1000).astype(np.float32) * 2.5)
    _, nf = frame_noise(1000 + utt, 0)
    return torch.tensor([O.sample_token(logits[i], torch.from_numpy(nf[i])) for i in range(8)], dtype=torch.int32)


def _rank_main(rank, world, port, n_utts, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from streamvoiceanon_amd.sharding import gather_results, shard_utterances, unshard

    shards = shard_utterances(list(range(n_utts)), world)
    local = torch.stack([_work(u) for u in shards[rank]])
    out = gather_results(local, world, rank)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        perm = unshard(shards)
        q.put((out[perm].numpy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _ragged_work(utt, length):
    """[8, length] int32 result keyed by the utterance id only"""
    rng = np.random.RandomState(100 + utt)
    return torch.from_numpy(rng.randint(0, 1000, size=(8, length)).astype(np.int32))


def _ragged_main(rank, world, port, lengths, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from streamvoiceanon_amd.sharding import gather_ragged, shard_utterances, unshard

    ids = list(range(len(lengths)))
    shards = shard_utterances(ids, world, lengths=lengths)          # LPT by length: ranks get different counts AND different lengths
    mine = [_ragged_work(u, lengths[u]) for u in shards[rank]]
    got = gather_ragged(mine, world, rank)
    if rank == 0:
        perm = unshard(shards)
        q.put(([got[i].numpy() for i in perm], [len(s) for s in shards]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ragged_gather_equals_single_process():
    """Utterances of unequal length (what LPT sharding is for): per-rank counts differ (1 vs 4 here) and every utterance keeps its own
    length through the gather -- the path configs[3] / configs[4] take when their utterances are not all 10 s long."""
    lengths = [400, 90, 110, 95, 101]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_main, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(counts) == [1, 4]
    assert len(got) == len(lengths)
    for u, n in enumerate(lengths):
        assert got[u].shape == (8, n)
        np.testing.assert_array_equal(got[u], _ragged_work(u, n).numpy())


def test_shard_utterances_lpt():
    sys.path.insert(0, ROOT)
    from streamvoiceanon_amd.sharding import shard_utterances

    s = shard_utterances(list(range(8)), 2)
    assert sorted(s[0] + s[1]) == list(range(8)) and len(s[0]) == len(s[1]) == 4
    s = shard_utterances([10, 11, 12, 13, 14], 2, lengths=[100, 10, 10, 10, 70])
    loads = [sum({10: 100, 11: 10, 12: 10, 13: 10, 14: 70}[u] for u in part) for part in s]
    assert max(loads) == 100 and sorted(s[0] + s[1]) == [10, 11, 12, 13, 14]
    assert shard_utterances(list(range(512)), 8)[3][:2] == [3, 11]     # 64 per GPU for config 4


def test_two_rank_gather_equals_single_process():
    n_utts, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n_utts, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.stack([_work(u) for u in range(n_utts)]).numpy()
    np.testing.assert_array_equal(got, ref)
    assert tmax == 2.0
