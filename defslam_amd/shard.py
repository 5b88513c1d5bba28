"""Sharding of independent SfT problems over the GPUs of one node.

The path partitions over independent problems (frames x templates / keyframes): rank r owns a contiguous block of
problem ids, solves them on its own GPU with its own context, and only the (small) results are gathered to rank 0.
There is no collective on the data path; `torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is used for the
result gather and the timing barrier only.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block partition: the first (n_items % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def solve_sharded(problem_ids: Sequence[int], solve_batch: Callable[[List[int]], List[np.ndarray]], dist=None) -> List[np.ndarray] | None:
    """Every rank solves its block with `solve_batch(ids) -> [result vector per id]`; rank 0 returns the results of all
    problems in id order (other ranks return None).  `dist` is torch.distributed (initialised) or None for one process."""
    ids = list(problem_ids)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return solve_batch(ids)
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = [ids[i] for i in shard_range(len(ids), rank, world)]
    local = solve_batch(mine)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    # ragged gather: sizes first, then one padded all_gather
    width = max((int(np.asarray(v).size) for v in local), default=0)
    meta = torch.tensor([len(local), width], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_n = max(int(m[0]) for m in metas)
    max_w = max(int(m[1]) for m in metas)
    buf = torch.zeros((max_n, max_w + 1), dtype=torch.float64, device=dev)
    for i, v in enumerate(local):
        a = np.asarray(v, np.float64).ravel()
        buf[i, 0] = a.size
        buf[i, 1:1 + a.size] = torch.from_numpy(a).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != 0:
        return None
    out: List[np.ndarray] = []
    for r in range(world):
        b = bufs[r].cpu().numpy()
        for i in range(int(metas[r][0])):
            n = int(b[i, 0])
            out.append(b[i, 1:1 + n].copy())
    return out
