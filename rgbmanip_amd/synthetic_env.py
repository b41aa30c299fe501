"""Synthetic stand-in for `MultiVecEnv` + `ControlInterface` on one rank.

The simulator (SAPIEN) is out of scope and absent on the GPU box, so the PPO benchmark / tests drive the trainer with this
environment: every env step costs exactly what the reference's `ControlInterface.step` costs on the estimator side — one
stereo pose estimate per env (`rl_pose.py:189-223` -> `estimate`) — with the camera move / render replaced by picking a
pre-rendered seeded view pair that is already resident in HBM.  Spaces match `rl_pose.py:35-37`
(action 12, state 75, observation 60), episodes last `max_steps` = 5 (`rl_pose.py:27,360-362`).
Observation slot t holds [camera target (7), predicted bbox centre/extent summary (4), 1] of step t; the reward is the negative
distance of the predicted handle centre to the (synthetic) ground truth plus a small action penalty.  It is a load generator with
the reference's interface and data shapes, not a model of the task.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth
from .adapose import postprocess
from .spaces import Box


class SyntheticPoseVecEnv:
    def __init__(self, num_envs, estimator_net, device, seed=0, max_steps=5, unique_views=16, rank=0):
        self.num_envs = num_envs
        self.net = estimator_net
        self.device = torch.device(device)
        self.max_steps = max_steps
        self.action_space = Box(low=-1.5, high=1.5, shape=(7 + max_steps,))
        self.state_space = Box(low=-1.5, high=1.5, shape=(max_steps * 15,))
        self.observation_space = Box(low=-1.5, high=1.5, shape=(max_steps * 12,))
        bank = synth.adapose_inputs(unique_views, seed=1000 + seed + rank)
        dev = self.device
        self.bank = {k: torch.from_numpy(v).to(dev) for k, v in bank.items()}
        self.bank["choose1"] = self.bank["choose1"].to(torch.int32)
        self.bank["choose2"] = self.bank["choose2"].to(torch.int32)
        self.unique = unique_views
        g = torch.Generator().manual_seed(seed + 17 * rank)
        self.gt_center = (torch.rand(num_envs, 3, generator=g) - 0.5).to(dev)
        self.env_ids = torch.arange(num_envs, device=dev)
        self.t = 0
        self.step_in_episode = torch.zeros(num_envs, dtype=torch.long, device=dev)
        self.obs = torch.zeros(num_envs, max_steps * 12, device=dev)
        self.state = torch.zeros(num_envs, max_steps * 15, device=dev)

    def reset(self, indices=None):
        self.obs.zero_()
        self.state.zero_()
        self.step_in_episode.zero_()
        return self.obs.clone()

    def get_state(self):
        return self.state.clone()

    def get_observation(self):
        return self.obs.clone()

    def step(self, actions, eval=False):
        n, dev = self.num_envs, self.device
        idx = (self.env_ids + self.t) % self.unique
        b = self.bank
        pred = self.net(b["img1"][idx], b["choose1"][idx], b["img2"][idx], b["choose2"][idx], b["P1"][idx], b["P2"][idx],
                        b["depths"][idx])
        bbox, ts, valid = postprocess(pred["view1_nocs"], pred["view1_depth"], pred["view1_r"], b["choose1"][idx], b["K1"][idx],
                                      b["E1"][idx])
        center = bbox.mean(dim=1).float()
        extent = (bbox.max(dim=1).values - bbox.min(dim=1).values).float().norm(dim=1, keepdim=True)
        reward = -(center - self.gt_center).norm(dim=1).clamp(max=20.0) - 0.05 * actions[:, :7].float().pow(2).sum(1)
        k = self.step_in_episode.clamp(max=self.max_steps - 1)
        feat = torch.cat([actions[:, :7].float().clamp(-1.5, 1.5), center.clamp(-1.5, 1.5), extent.clamp(max=1.5),
                          torch.ones(n, 1, device=dev)], dim=1)                         # [n,12]
        rows = torch.arange(n, device=dev)
        obs3 = self.obs.view(n, self.max_steps, 12)
        obs3[rows, k] = feat
        st3 = self.state.view(n, self.max_steps, 15)
        st3[rows, k] = torch.cat([feat, self.gt_center], dim=1)
        self.step_in_episode += 1
        done = self.step_in_episode >= self.max_steps
        if bool(done.any()):
            self.obs[done] = 0
            self.state[done] = 0
            self.step_in_episode[done] = 0
        self.t += 1
        info = {"valid_rate": valid.float().mean().reshape(1)}
        return self.obs.clone(), reward, done, info
