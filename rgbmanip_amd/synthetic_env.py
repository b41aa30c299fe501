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


# ======================================================================================================================
# Synthetic MultiVecEnv: the method surface `ControlInterface` calls on `env/my_vec_env.py:201-522`
# (reset, cam_move_to, get_image, get_observation, camera_pose, robot_pose, get_attr, close), one partition per rank.
# ======================================================================================================================
import ctypes as _C
import math as _math

from . import _lib

CAM_W, CAM_H = 640, 480
CAM_F = (CAM_H / 2) / _math.tan(0.5)          # fovy = 1 rad (base_manipulation.py:20, base_sapien_env.py:89-111)


def sample_scene(env_id: int, episode: int):
    """Seeded scene of one environment (seed 1000 + env_id): robot root [7] and the handle box [15] =
    centre (3), axis rows X (up), Y, Z (outward normal, facing the robot) (9), half extents (3), world frame."""
    rng = np.random.default_rng([1000 + int(env_id), int(episode)])
    robot = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 0.0, 1.0, 0.0, 0.0, 0.0])
    yaw = rng.uniform(-0.5, 0.5)
    c, s = np.cos(yaw), np.sin(yaw)
    X, Y, Z = np.array([0.0, 0.0, 1.0]), np.array([-s, c, 0.0]), np.array([-c, -s, 0.0])
    centre = robot[:3] + np.array([rng.uniform(0.8, 1.05), rng.uniform(-0.2, 0.2), rng.uniform(0.5, 0.9)])
    half = np.array([rng.uniform(0.06, 0.12), rng.uniform(0.02, 0.035), rng.uniform(0.02, 0.04)])
    return robot, np.concatenate([centre, X, Y, Z, half])


def box_corners(box):
    """Handle corners [N,8,3] in the reference's order (open_cabinet.py:153-158: centre (b0+b6)/2, x = b1-b0, y = b0-b2,
    z = b4-b0)."""
    c, X, Y, Z, h = box[:, 0:3], box[:, 3:6], box[:, 6:9], box[:, 9:12], box[:, 12:15]
    hx, hy, hz = h[:, 0:1] * X, h[:, 1:2] * Y, h[:, 2:3] * Z
    b0 = c - hx + hy - hz
    return torch.stack([b0, b0 + 2 * hx, b0 - 2 * hy, b0 + 2 * hx - 2 * hy, b0 + 2 * hz, b0 + 2 * hx + 2 * hz,
                        b0 + 2 * hx - 2 * hy + 2 * hz, b0 - 2 * hy + 2 * hz], dim=1)


class SyntheticMultiVecEnv:
    """Procedural stand-in for `MultiVecEnv` on one rank: `num_envs` scenes with global ids
    `env_id_offset .. env_id_offset + num_envs` (rank r owns `[r*512, (r+1)*512)`), each a robot root and an
    oriented handle box re-sampled at every `reset`.  `get_image` renders the 480x640 colour frame, handle mask,
    intrinsic and extrinsic of the hand camera with the HIP kernels of csrc/synth_env.hip; every returned array is a
    CUDA tensor.  `cam_move_to` is a reach model: targets farther than `reach` from the shoulder fail and leave the camera
    half way; it returns `[success, period]` like `merge_obs` of the reference's per-env `(bool, int)` tuples."""

    def __init__(self, num_envs: int, device, seed: int = 0, env_id_offset: int = 0, reach: float = 0.55, episodes: int = 32):
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.env_ids = np.arange(env_id_offset, env_id_offset + self.num_envs) + 100000 * int(seed)
        self.env0 = int(env_id_offset)
        self.reach = float(reach)
        self.lib = _lib.load()
        dev = self.device
        # scenes of the first `episodes` episodes of every env, sampled once on the host and kept on the device, so a reset
        # is an index operation on the GPU (no host->device copy, which would stall the stream); episode e reuses e % episodes
        self.bank_episodes = int(episodes)
        scenes = [[sample_scene(i, e) for i in self.env_ids] for e in range(self.bank_episodes)]
        self._bank_robot = torch.from_numpy(np.stack([[sc[0] for sc in row] for row in scenes])).to(dev)    # [E,N,7]
        self._bank_box = torch.from_numpy(np.stack([[sc[1] for sc in row] for row in scenes])).to(dev)      # [E,N,15]
        self.episode = np.zeros(self.num_envs, dtype=np.int64)               # host mirror of the per-env episode counter
        self._episode = torch.zeros(self.num_envs, dtype=torch.int64, device=dev)
        self._arange = torch.arange(self.num_envs, device=dev)
        self._cam = torch.zeros(self.num_envs, 7, dtype=torch.float64, device=dev)
        self._cam[:, 2] = 0.7
        self._cam[:, 3] = 1.0
        self._success = torch.zeros(self.num_envs, 1, dtype=torch.float64, device=dev)
        self._shoulder = torch.tensor([0.0, 0.0, 0.6], dtype=torch.float64, device=dev)
        self._load_scenes()

    def _load_scenes(self):
        slot = self._episode % self.bank_episodes
        self._robot = self._bank_robot[slot, self._arange]
        self._box = self._bank_box[slot, self._arange]

    # ---- my_vec_env.py:214
    def reset(self, indices=None):
        if indices is None:
            self.episode += 1
            self._episode += 1
            self._success.zero_()
        else:
            idx = np.atleast_1d(np.asarray(indices))
            self.episode[idx] += 1
            sel = torch.as_tensor(idx, device=self.device)
            self._episode[sel] += 1
            self._success[sel] = 0
        self._load_scenes()
        return None

    # ---- my_vec_env.py:382, base_manipulation.py:544
    def cam_move_to(self, pose, time=2, wait=1, planner="ik", robot_frame=False, skip_move=False, no_collision_with_front=True):
        target = torch.as_tensor(pose).to(device=self.device, dtype=torch.float64)
        if target.dim() == 1:
            target = target.expand(self.num_envs, 7)
        target = target.clone()
        if not robot_frame:
            target[:, :3] -= self._robot[:, :3]
        ok = (target[:, :3] - self._shoulder).norm(dim=1) < self.reach
        step = target[:, :3] - self._cam[:, :3]
        period = torch.ceil(step.norm(dim=1) * 400.0) + 1.0
        self._cam[:, :3] = torch.where(ok[:, None], target[:, :3], self._cam[:, :3] + 0.5 * step)
        self._cam[:, 3:] = target[:, 3:] / target[:, 3:].norm(dim=1, keepdim=True)
        return [ok, period]

    # ---- my_vec_env.py:266, base_manipulation.py:653-687
    def _scene(self):
        sc = _lib.SynthScene()
        sc.cam_pose, sc.robot_pose, sc.box = self._cam.data_ptr(), self._robot.data_ptr(), self._box.data_ptr()
        sc.fx = sc.fy = CAM_F
        sc.cx, sc.cy = CAM_W / 2, CAM_H / 2
        sc.N, sc.H, sc.W, sc.env0 = self.num_envs, CAM_H, CAM_W, self.env0
        return sc

    def get_image(self, mask="handle"):
        n, dev = self.num_envs, self.device
        K = torch.empty(n, 3, 3, dtype=torch.float64, device=dev)
        E = torch.empty(n, 4, 4, dtype=torch.float64, device=dev)
        rays = torch.empty(n, 12, dtype=torch.float64, device=dev)
        color = torch.empty(n, CAM_H, CAM_W, 3, dtype=torch.float32, device=dev)
        msk = torch.empty(n, CAM_H, CAM_W, dtype=torch.uint8, device=dev)
        sc = self._scene()
        _lib.check(self.lib.rgbm_synth_camera(_C.byref(sc), _lib.ptr(K), _lib.ptr(E), _lib.ptr(rays), _lib.stream_ptr()), "rgbm_synth_camera")
        _lib.check(self.lib.rgbm_synth_render(_C.byref(sc), _lib.ptr(rays), _lib.ptr(color), _lib.ptr(msk), _lib.stream_ptr()),
                   "rgbm_synth_render")
        return {"camera0": {"Color": color, "Mask": msk, "Intrinsic": K, "Extrinsic": E}}

    # ---- my_vec_env.py:281, open_cabinet.py:191-214
    def get_observation(self, gt=False):
        obs = {"success": self._success.clone()}
        if gt:
            obs["handle_bbox"] = box_corners(self._box)
        return obs

    # ---- my_vec_env.py:466, 482
    def camera_pose(self, robot_frame=False):
        pose = self._cam.clone()
        if not robot_frame:
            pose[:, :3] += self._robot[:, :3]
        return pose

    def robot_pose(self):
        return self._robot.clone()

    # ---- my_vec_env.py:513, 524
    def get_attr(self, attr_name, indices=None):
        if attr_name == "current_obj_config":
            return [{"name": f"synthetic_handle_{i}"} for i in self.env_ids]
        return [getattr(self, attr_name)] * self.num_envs

    def close(self):
        pass


class SyntheticManipulation:
    """Stand-in for `OpenCabinetManipulation.plan_pathway` (models/manipulation/open_cabinet.py): the grasp succeeds where the
    predicted centre is within `tol` metres of the handle centre; the outcome is what `get_observation()["success"]` reports."""

    def __init__(self, env: SyntheticMultiVecEnv, tol: float = 0.05):
        self.env, self.tol = env, tol

    def plan_pathway(self, center, direction, eval=False):
        gt = self.env._box[:, 0:3]
        self.env._success = ((torch.as_tensor(center).to(gt) - gt).norm(dim=1, keepdim=True) < self.tol).to(torch.float64)
