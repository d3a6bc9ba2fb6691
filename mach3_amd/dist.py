"""Event sharding for the multi-GPU path (the reference is single device).

Events are independent up to the per-bin sum, so a sample is cut into contiguous event
blocks, one per rank; each rank fills its own partial bin vector and ONE all-reduce (sum) of
that vector precedes the bin-likelihood sum.  On the GPU the all-reduce is the in-library
ncclAllReduce of m3hip_comm_init (RCCL over xGMI); `allreduce_bins` below is the same step
expressed with torch.distributed so the partitioning logic can be exercised with gloo on CPUs."""
from __future__ import annotations

import numpy as np

GROUP = 1024  # shard boundaries stay on the engine's classification granularity


def shard_ranges(n_events: int, n_ranks: int, weights=None):
    """Contiguous [e0, e1) per rank.  Without `weights` the events are split evenly (rounded to
    GROUP); with per-event `weights` (e.g. responses per event for sparse samples) the split
    balances the summed weight instead of the event count."""
    if n_ranks <= 0:
        raise ValueError("n_ranks must be positive")
    if weights is None:
        per = -(-n_events // n_ranks)
        per = -(-per // GROUP) * GROUP
        cuts = [min(n_events, r * per) for r in range(n_ranks + 1)]
    else:
        w = np.asarray(weights, np.float64)
        assert len(w) == n_events
        c = np.concatenate([[0.0], np.cumsum(w)])
        cuts = [0]
        for r in range(1, n_ranks):
            target = c[-1] * r / n_ranks
            e = int(np.searchsorted(c, target))
            e = min(n_events, max(cuts[-1], int(round(e / GROUP)) * GROUP))
            cuts.append(e)
        cuts.append(n_events)
    return [(cuts[r], cuts[r + 1]) for r in range(n_ranks)]


def allreduce_bins(mc: np.ndarray, w2: np.ndarray | None = None):
    """In-place sum over ranks of the bin-rate vector (and w2 while it is being updated)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(mc)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if w2 is not None:
        t2 = torch.from_numpy(w2)
        dist.all_reduce(t2, op=dist.ReduceOp.SUM)
    return mc, w2
