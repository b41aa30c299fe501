"""Host-side helpers for the one-process-per-GPU layout: contiguous env/pose shards per rank and the two
small exchanges the PPO update needs (advantage statistics, flat gradient + loss statistics)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `total` independent units (envs / poses) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_adv_stats(local_sums: torch.Tensor, local_count: int, group=None):
    """all-reduce {sum(adv), sum(adv^2)} and the element count -> (mean, unbiased std) of the global advantage vector,
    i.e. what `storage.py:63-64` computes in a single process."""
    buf = torch.cat([local_sums[:2].double().reshape(2), torch.tensor([float(local_count)], dtype=torch.float64,
                                                                      device=local_sums.device)])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, group=group)
    n = buf[2]
    mean = buf[0] / n
    var = (buf[1] - n * mean * mean) / (n - 1.0)
    return mean, var.clamp_min(0).sqrt(), int(n.item())


def average_flat_gradient(grads_and_stats: torch.Tensor, group=None):
    """Sum the [total+4] buffer of `rgbm_ppo_minibatch_fwd_bwd` over ranks; returns the factor the optimiser must apply to
    the gradient part (1/world).  The statistics part {sum surrogate, sum value loss, sum KL, rows} stays a sum."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(grads_and_stats, group=group)
    return 1.0 / world
