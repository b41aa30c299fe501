"""RolloutStorage — buffers and GAE.  Mirrors `/root/reference/algo/ppo/ppo/storage.py:5-86`; `compute_returns` runs the
HIP GAE kernel (`rgbm_gae` + `rgbm_adv_normalise`), with the advantage statistics all-reduced when several ranks train
one policy (each rank owns a slice of the envs)."""
from __future__ import annotations

import torch

from .. import _lib, dist_utils


class RolloutStorage:
    def __init__(self, num_envs, num_transitions_per_env, obs_shape, states_shape, actions_shape, device="cpu",
                 sampler="sequential"):
        self.device = device
        self.sampler = sampler
        T, N = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(*s, device=self.device)  # noqa: E731
        self.observations = z(T, N, *obs_shape)
        self.states = z(T, N, *states_shape)
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1).byte()
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self.num_transitions_per_env = T
        self.num_envs = N
        self.step = 0
        self._sums = None
        self._count = None           # (process group, global element count) of compute_returns

    _FIELDS = ("observations", "states", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma")

    def add_transitions(self, observations, states, actions, rewards, dones, values, actions_log_prob, mu, sigma):
        """Append one time step for all envs (same argument order as storage.py:32)."""
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        t = self.step
        column = {"rewards", "dones", "actions_log_prob"}          # stored as [N,1]
        for name, val in zip(self._FIELDS, (observations, states, actions, rewards, dones, values, actions_log_prob, mu, sigma)):
            getattr(self, name)[t].copy_(val.view(-1, 1) if name in column else val)
        self.step = t + 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam, process_group=None):
        if self.rewards.device.type != "cuda":
            raise _lib.RgbmError("RolloutStorage.compute_returns runs on the HIP GAE kernel: use a cuda device (no CPU fallback)")
        lib = _lib.load()
        T, N = self.num_transitions_per_env, self.num_envs
        if self._sums is None:
            self._sums = torch.zeros(2 + 2 * ((N + 255) // 256), dtype=torch.float64, device=self.rewards.device)
        lv = last_values.to(device=self.rewards.device, dtype=torch.float32).contiguous()
        adv = self.advantages if self.advantages.is_contiguous() else torch.empty_like(self.returns)
        _lib.check(lib.rgbm_gae(T, N, _lib.ptr(self.rewards), _lib.ptr(self.dones), _lib.ptr(self.values), _lib.ptr(lv),
                                float(gamma), float(lam), _lib.ptr(self.returns), _lib.ptr(adv), _lib.ptr(self._sums),
                                _lib.stream_ptr()), "rgbm_gae")
        # several ranks train one policy: the statistics are those of ALL ranks' advantages.  The element count is summed over
        # ranks too (shards may differ by an env, dist_utils.shard_range); it cannot change, so it is exchanged once.
        if self._count is None or self._count[0] is not process_group:
            self._count = (process_group, dist_utils.global_count(T * N, self.rewards.device, process_group))
        count = self._count[1]
        dist_utils.all_reduce_adv_sums(self._sums, process_group)
        _lib.check(lib.rgbm_adv_normalise(T * N, _lib.ptr(adv), _lib.ptr(self._sums), count, _lib.stream_ptr()),
                   "rgbm_adv_normalise")
        self.advantages = adv

    def get_statistics(self):
        """(mean trajectory length, mean reward) as storage.py:66-72: trajectories are cut at every done and at the
        last stored step, so the lengths sum to T*N and their mean is T*N / #cuts."""
        cuts = self.dones.clone()
        cuts[-1] = 1
        n_traj = cuts.ne(0).sum().clamp_min(1).float().cpu()
        total = float(self.num_transitions_per_env * self.num_envs)
        return total / n_traj, self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches):
        """Index blocks of T*N // k rows, remainder dropped (storage.py:74-86).  "sequential" yields contiguous ranges."""
        total = self.num_envs * self.num_transitions_per_env
        size = total // num_mini_batches
        if self.sampler == "sequential":
            return [range(b * size, (b + 1) * size) for b in range(total // size)]
        if self.sampler == "random":
            perm = torch.randperm(total).tolist()
            return [perm[b * size:(b + 1) * size] for b in range(total // size)]
        raise ValueError(f"unknown sampler {self.sampler!r}")
