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


def unshard(gathered_ids: List[List[int]]):
    """Permutation that maps the rank-major gather order back to global utterance order."""
    flat = [u for part in gathered_ids for u in part]
    return sorted(range(len(flat)), key=lambda i: flat[i])
