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


_DTYPES = ("int16", "int32", "int64", "float32", "float16", "uint8", "int8", "float64", "bfloat16")


def gather_ragged(items, world: int, rank: int, dst: int = 0, force: bool = False, pad_value=0, device=None):
    """Gather per-utterance results of UNEQUAL length -- `items`: this rank's list of tensors [..., T_i] (same leading dims and dtype,
    time last; e.g. int16 / int32 codes [8, T_i] of utterances of different durations, which is what LPT sharding by length produces,
    and ranks may hold different NUMBERS of utterances, ZERO included: fewer utterances than ranks) -- to `dst`.  Fixed-shape
    collectives instead of a gather_object (device tensors stay on the device, RCCL-friendly): one all_reduce(MAX) makes every rank
    agree on (max count, max T, leading dims, dtype) -- a rank without utterances learns the layout from it --, one all_reduce(MIN)
    of a consistency flag makes a mismatch raise on EVERY rank before any data moves (no rank is left waiting in a collective), then
    every rank sends [n_max, ..., T_max] padded with `pad_value` plus its lengths [n_max] (-1 = no utterance); `dst` cuts the padding
    off again.  `device`: where the collectives' tensors live on a rank with no items (default: the current CUDA device under the
    NCCL / RCCL backend, else CPU).  Returns the list of tensors in rank-major order on `dst` (sharding.unshard maps it back to
    utterance order), None elsewhere."""
    import torch
    import torch.distributed as dist

    if (world == 1 and not force) or not dist.is_initialized():
        return list(items)
    if items:
        dev = items[0].device
    elif device is not None:
        dev = torch.device(device)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    # meta = [count, T_max, n_lead, lead dims (<= 4, stored + 1), dtype code + 1]; zeros where this rank does not know
    meta = [len(items), max((int(t.shape[-1]) for t in items), default=0), 0, 0, 0, 0, 0, 0]
    own_ok = 1
    if items:
        lead = tuple(int(x) for x in items[0].shape[:-1])
        name = str(items[0].dtype).replace("torch.", "")
        own_ok = int(len(lead) <= 4 and name in _DTYPES and all(tuple(t.shape[:-1]) == lead and t.dtype == items[0].dtype for t in items))
        if own_ok:
            meta[2] = len(lead) + 1
            for i, x in enumerate(lead):
                meta[3 + i] = x + 1
            meta[7] = _DTYPES.index(name) + 1
    mt = torch.tensor(meta, dtype=torch.int64, device=dev)
    dist.all_reduce(mt, op=dist.ReduceOp.MAX)
    agreed = [int(x) for x in mt.tolist()]
    if items and own_ok and (agreed[2:] != meta[2:]):
        own_ok = 0                                   # another rank holds other leading dims / another dtype
    ok = torch.tensor([own_ok], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        raise ValueError("gather_ragged: ranks disagree on the leading dims / dtype of their items (or an item list is inconsistent)")
    n_max, t_max = agreed[0], agreed[1]
    if n_max == 0:                                   # nobody holds anything
        return [] if rank == dst else None
    lead = tuple(x - 1 for x in agreed[3:3 + agreed[2] - 1])
    dtype = getattr(torch, _DTYPES[agreed[7] - 1])
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
