"""Multi-GPU layer: independent capture streams shard embarrassingly across ranks (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  There is no
data-path collective — every stream is an independent recurrence / STFT — only a small
per-stream summary table is all-gathered once per reporting epoch (K8 `stats_gather`, 48 B/stream:
latency-bound, so the RCCL default algorithm is used and the call stays out of the per-hop loop)."""
from __future__ import annotations

from typing import Tuple

STATS_COLUMNS = ("momentary_lufs", "short_term_lufs", "max_true_peak_db", "rho_full", "rho_low", "rho_mid", "rho_high",
                 "frames_emitted", "mean_points_per_frame", "last_frame_points", "held_peak_left_db", "held_peak_right_db")


def shard_streams(total_streams: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Static contiguous partition: stream s lives on rank s // ceil(total / world).  Returns (first, count)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    per = -(-total_streams // world_size)
    first = min(rank * per, total_streams)
    return first, max(0, min(per, total_streams - first))


def gather_stats(local_stats, total_streams: int, always_collective: bool = False):
    """All-gather the per-stream summary rows.  `local_stats` is a float32 tensor [n_local, K] on this rank's
    device; returns a tensor [total_streams, K] identical on every rank (padding rows of uneven shards are
    dropped).  Works on CUDA/HIP tensors (RCCL) and CPU tensors (gloo).  A world of one rank returns its rows
    without a collective unless `always_collective` (the 1-GPU box's way to run the RCCL call itself)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return local_stats[:total_streams]
    world = dist.get_world_size()
    per = -(-total_streams // world)
    k = local_stats.shape[1]
    padded = torch.zeros((per, k), dtype=local_stats.dtype, device=local_stats.device)
    padded[: local_stats.shape[0]] = local_stats
    out = torch.empty((world * per, k), dtype=local_stats.dtype, device=local_stats.device)
    dist.all_gather_into_tensor(out, padded)
    rows = []
    for r in range(world):
        first, count = shard_streams(total_streams, r, world)
        rows.append(out[r * per: r * per + count])
    return torch.cat(rows, dim=0)
