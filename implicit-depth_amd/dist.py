"""Batch sharding + the path's single collective.

The hot path is embarrassingly parallel over the batch dimension (every op reduces over
channels / views / planes of ONE frame; no BatchNorm in eval), so multi-GPU = one process per
GPU, rank r takes frames [r*B/G, (r+1)*B/G), weights replicated, no activation exchange.  The
only message is an all-gather of the per-frame metric vectors after the forward — the analogue
of Lightning's ``sync_dist=True`` scalar reduction in the reference (bd_model.py:672,685) — a
~1 KB latency-bound message, so ring-vs-direct and xGMI link bandwidth are irrelevant.
Backend: ``nccl`` (= RCCL over xGMI on ROCm) on GPUs, ``gloo`` on CPU for tests.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (total % world) ranks get one extra frame."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch: Dict[str, torch.Tensor], world: int, rank: int, batch_size: int) -> Dict[str, torch.Tensor]:
    """Slice every tensor whose leading dim is the global batch; broadcastable entries
    (e.g. the (1,1,1,1) min/max depth tensors) pass through."""
    lo, hi = shard_range(batch_size, world, rank)
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.ndim > 0 and v.shape[0] == batch_size:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def all_gather_metrics(local: torch.Tensor, counts: Sequence[int] | None = None) -> torch.Tensor:
    """All-gather per-frame metric rows (B_local, M) -> (B_global, M), rank order = frame order.
    Ragged shards (B % G != 0) are padded to the largest shard for the collective and trimmed."""
    if not (dist.is_available() and dist.is_initialized()):
        return local  # single process without a process group; a 1-rank group still runs the collective
    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    if counts is None:
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local)
        counts = [int(s.item()) for s in sizes]
    mx = max(counts)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros(mx - local.shape[0], *local.shape[1:])], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def nanmean_rows(gathered: torch.Tensor) -> torch.Tensor:
    """Final averaging à la ResultsAverager.compute_final_average (metrics_utils.py:341-371):
    nan-aware mean over frames, per metric column."""
    return torch.nanmean(gathered, dim=0)
