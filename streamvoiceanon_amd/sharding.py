"""Utterance-parallel multi-GPU layer.

Streams share nothing but read-only weights, so the path shards by independent utterances: one
process per GPU, every rank holds a full weight replica and its own streams' state, and there is
no data-path collective.  The only communication is the gather of per-utterance results at the end
(RCCL over xGMI on MI355X; gloo in the CPU tests) and a MAX of the wall time in bench.py.
Sampler noise is keyed by utterance id (never rank or slot), so per-utterance outputs are identical
for any world size.
"""
from __future__ import annotations

from typing import List, Sequence


def shard_utterances(utt_ids: Sequence[int], world: int, lengths: Sequence[int] | None = None) -> List[List[int]]:
    """Longest-processing-time assignment of utterances to `world` ranks (round-robin when all lengths
    are equal).  Deterministic; returns one list of utterance ids per rank."""
    ids = list(utt_ids)
    if lengths is None:
        lengths = [1] * len(ids)
    order = sorted(range(len(ids)), key=lambda i: (-lengths[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], len(out[k]), k))
        out[r].append(ids[i])
        loads[r] += lengths[i]
    for r in range(world):
        out[r].sort()
    return out


def gather_results(local, world: int, rank: int, dst: int = 0, force: bool = False):
    """Gather equally-shaped per-rank result tensors [n_local, ...] to `dst` -> [world*n_local, ...]
    (None on other ranks).  Direct peer->root sends (dist.gather), not a ring: on MI355X the 7 peers
    arrive on 7 distinct xGMI links of the root."""
    import torch
    import torch.distributed as dist

    if (world == 1 and not force) or not dist.is_initialized():
        return local
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat(bufs, dim=0)


def gather_ragged(items, world: int, rank: int, dst: int = 0, force: bool = False, pad_value=0):
    """Gather per-utterance results of UNEQUAL length -- `items`: this rank's list of tensors [..., T_i] (same leading dims and dtype,
    time last; e.g. int16 / int32 codes [8, T_i] of utterances of different durations, which is what LPT sharding by length produces,
    and ranks may hold different NUMBERS of utterances) -- to `dst`.  Two fixed-shape collectives instead of a gather_object (device
    tensors stay on the device, RCCL-friendly): all ranks agree on (max count, max T) with one all_reduce(MAX), every rank sends
    [n_max, ..., T_max] padded with `pad_value` plus its lengths [n_max] (-1 = no utterance); `dst` cuts the padding off again.
    Returns the list of tensors in rank-major order on `dst` (sharding.unshard maps it back to utterance order), None elsewhere."""
    import torch
    import torch.distributed as dist

    if (world == 1 and not force) or not dist.is_initialized():
        return list(items)
    assert len(items) > 0 or world > 1
    dev = items[0].device if items else torch.device("cpu")
    dims = torch.tensor([len(items), max((int(t.shape[-1]) for t in items), default=0)], dtype=torch.int64, device=dev)
    dist.all_reduce(dims, op=dist.ReduceOp.MAX)
    n_max, t_max = int(dims[0]), int(dims[1])
    lead = tuple(items[0].shape[:-1]) if items else None
    if lead is None:          # a rank without utterances still takes part: learn the leading dims from the root's view (all ranks share them)
        raise ValueError("gather_ragged: every rank must hold at least one utterance (shard_utterances gives each rank one when n >= world)")
    dtype = items[0].dtype
    pad = torch.full((n_max,) + lead + (t_max,), pad_value, dtype=dtype, device=dev)
    lens = torch.full((n_max,), -1, dtype=torch.int64, device=dev)
    for i, t in enumerate(items):
        pad[i, ..., : t.shape[-1]] = t
        lens[i] = t.shape[-1]
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    lbufs = [torch.empty_like(lens) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    dist.gather(lens, lbufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        for i in range(n_max):
            n = int(lbufs[r][i])
            if n >= 0:
                out.append(bufs[r][i, ..., :n])
    return out


def unshard(gathered_ids: List[List[int]]):
    """Permutation that maps the rank-major gather order back to global utterance order."""
    flat = [u for part in gathered_ids for u in part]
    return sorted(range(len(flat)), key=lambda i: flat[i])
