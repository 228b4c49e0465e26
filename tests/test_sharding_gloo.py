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


def _ragged_empty_main(rank, world, port, lengths, bad_rank, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from streamvoiceanon_amd.sharding import gather_ragged, shard_utterances, unshard

    shards = shard_utterances(list(range(len(lengths))), world, lengths=lengths)
    mine = [_ragged_work(u, lengths[u]) for u in shards[rank]]
    if rank == bad_rank:
        mine = [t.to(torch.int64) for t in mine]           # one rank with another dtype: EVERY rank must raise, nobody may hang
    try:
        got = gather_ragged(mine, world, rank)
        raised = False
    except ValueError:
        got, raised = None, True
    flag = torch.tensor([int(raised)])
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    if rank == 0:
        perm = unshard(shards)
        q.put(([got[i].numpy() for i in perm] if got is not None else None, [len(s_) for s_ in shards], int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_ragged_gather_with_empty_ranks_and_consistent_errors():
    """fewer utterances than ranks (world 4, 2 utterances: two ranks hold NOTHING and still take part in the collectives, learning
    the layout from the all_reduce), and a dtype mismatch on one rank raises ValueError on every rank instead of leaving the others blocked."""
    lengths = [37, 12]
    got, counts, n_raised = _run_world(_ragged_empty_main, 4, (lengths, -1))
    assert sorted(counts) == [0, 0, 1, 1] and n_raised == 0
    for u, n in enumerate(lengths):
        np.testing.assert_array_equal(got[u], _ragged_work(u, n).numpy())
    got, counts, n_raised = _run_world(_ragged_empty_main, 4, ([20, 30, 40, 50, 60], 2))
    assert got is None and n_raised == 4


def _bench_rank_logic_main(rank, world, port, n_utts, frames, q):
    """bench.py's rank logic for configs[3] / [4] on gloo: shard the utterance ids, produce each utterance's [8, T] codes, gather, unshard"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from streamvoiceanon_amd.sharding import gather_ragged, gather_results, shard_utterances, unshard

    shards = shard_utterances(list(range(n_utts)), world)
    local = torch.stack([_ragged_work(u, frames) for u in shards[rank]])
    out = gather_results(local, world, rank)
    rag = gather_ragged([_ragged_work(u, frames) for u in shards[rank]], world, rank)
    if rank == 0:
        perm = unshard(shards)
        assert out.shape[0] == n_utts == world * local.shape[0]                 # gathered_utterances == N x B
        ok = all(torch.equal(out[perm[u]], _ragged_work(u, frames)) and torch.equal(rag[perm[u]], _ragged_work(u, frames)) for u in range(n_utts))
        q.put((ok, [len(s_) for s_ in shards]))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_bench_rank_logic_512_and_256_utterances():
    """BASELINE configs[3] / [4] at their full utterance counts on 8 ranks (gloo): shard_utterances -> per-rank results -> gather ->
    unshard returns every utterance's own result in utterance order, 64 / 32 utterances per rank."""
    for n_utts, per in ((512, 64), (256, 32)):
        ok, counts = _run_world(_bench_rank_logic_main, 8, (n_utts, 5))
        assert ok and counts == [per] * 8


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
