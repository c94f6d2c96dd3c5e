"""Read sharding across GPUs: contiguous, record-aligned ranges, no exchange step.

Rank r of W owns global reads [lo, hi); the only cross-rank traffic is the sum of the
statistics counters on the host side (and, in bench.py, the barrier / max-over-ranks of the
elapsed time).  No RCCL collective touches read data.
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``n_total`` units over ``world`` ranks; the first ``n_total % world``
    ranks get one extra unit."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_stats(dicts):
    """Sum per-rank ``cs_stats.as_dict()`` dictionaries."""
    out = None
    for d in dicts:
        if out is None:
            out = {k: (list(v) if isinstance(v, list) else v) for k, v in d.items()}
            continue
        for k, v in d.items():
            if isinstance(v, list):
                out[k] = [a + b for a, b in zip(out[k], v)]
            else:
                out[k] += v
    return out
