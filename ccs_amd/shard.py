"""ZMW sharding across GPUs / ranks.

ZMWs are independent (docs/how-does-ccs-work.md:12-17); the reference scales out with `--chunk i/N` and a
file merge, no communication (docs/faq/parallelize.md:8-29).  Here rank i of N takes a contiguous,
cost-balanced slice (cost = passes x length, the polish work) and results are concatenated in rank order,
so the output is independent of N.  No collective is on the data path.
"""
from __future__ import annotations

import numpy as np

from .api import Batch


def zmw_cost(batch: Batch) -> np.ndarray:
    """Estimated work per ZMW = total subread bases (passes x length)."""
    bo = batch.base_off
    ro = batch.read_off
    return (bo[ro[1:]] - bo[ro[:-1]]).astype(np.int64)


def shard_bounds(batch: Batch, world: int) -> np.ndarray:
    """[world+1] ZMW boundaries: contiguous slices of ~equal cost (every rank gets >= 0 ZMWs, order preserved)."""
    cost = zmw_cost(batch)
    cum = np.concatenate([[0], np.cumsum(cost)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        z = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(z, bounds[-1]), batch.n_zmw))
    bounds.append(batch.n_zmw)
    return np.asarray(bounds, np.int64)


def shard(batch: Batch, rank: int, world: int) -> Batch | None:
    b = shard_bounds(batch, world)
    if b[rank + 1] == b[rank]:
        return None
    return batch.slice(int(b[rank]), int(b[rank + 1]))
