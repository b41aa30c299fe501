"""Host-side helpers for the one-process-per-GPU layout: contiguous env/pose shards per rank and the
exchanges the PPO trainer makes — `RolloutStorage.compute_returns` (advantage statistics) and `PPO.update` (flat gradient +
loss statistics) call these, so the multi-rank arithmetic lives in one tested place."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_range(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `total` independent units (envs / poses) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def require_uniform(value: int, what: str, device, group=None) -> int:
    """Every rank must hold the same `value` (envs per rank, minibatches per epoch): the per-minibatch all-reduces of
    `PPO.update` are matched one to one across ranks, and `average_flat_gradient` weighs every rank's minibatch mean
    equally.  Raises on every rank if they differ (instead of hanging in a later collective).  Returns the world size."""
    world = world_size(group)
    if world == 1:
        return 1
    t = torch.tensor([float(value), -float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)          # max(v) and max(-v) = -min(v) in one exchange
    vmax, vmin = int(t[0].item()), int(-t[1].item())
    if vmax != vmin:
        raise ValueError(f"{what} must be equal on all ranks (min {vmin}, max {vmax}): shard the environments evenly")
    return world


def global_count(local_count: int, device, group=None) -> float:
    """Sum of the ranks' element counts (one small all-reduce and one host read; callers cache the result)."""
    if world_size(group) == 1:
        return float(local_count)
    t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return float(t.item())


def all_reduce_adv_sums(sums: torch.Tensor, group=None) -> None:
    """In place: {sum(adv), sum(adv^2)} (fp64, device) summed over ranks — with the global count this gives the mean and
    unbiased std that `storage.py:63-64` computes over all T*N advantages in a single process."""
    if world_size(group) > 1:
        dist.all_reduce(sums[:2], group=group)


def adv_mean_std(sums: torch.Tensor, count: float):
    """(mean, unbiased std) from {sum, sum of squares} and the element count — the arithmetic of `rgbm_adv_normalise`."""
    mean = sums[0] / count
    var = (sums[1] - count * mean * mean) / (count - 1.0)
    return mean, var.clamp_min(0).sqrt()


def average_flat_gradient(grads_and_stats: torch.Tensor, group=None) -> float:
    """Sum the [total+4] buffer of `rgbm_ppo_minibatch_fwd_bwd` over ranks; returns the factor the optimiser must apply to
    the gradient part (1/world).  The statistics part {sum surrogate, sum value loss, sum KL, rows} stays a sum."""
    world = world_size(group)
    if world > 1:
        dist.all_reduce(grads_and_stats, group=group)
    return 1.0 / world
