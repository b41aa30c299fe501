"""Device-resident view queue of the RL pose controller.

Mirrors the estimator-facing half of `ControlInterface` (`/root/reference/models/controller/rl_pose.py:14-223`): the
queues of the last `max_steps` views per env (`reset_queue`, `add_view`, `add_bbox`), the policy's observation / state
encoders (`get_observation`, `get_state`) and `get_estimation`, which picks each env's two most recent usable views and
calls the pose estimator — here without leaving the GPU: frames, masks and camera matrices stay CUDA tensors (float32
frames instead of the reference's float64 host arrays), the per-env mask extent comes from `rgbm_mask_extent`, the view
selection is index arithmetic instead of the O(max_steps * N) Python double loop copying 480x640x3 images, and the
estimator is entered through `estimate_device`.  Reward shaping, camera motion and the manipulation call (`step`,
`get_reward`, `call_manipulation`, `reset_robot`) need the simulator and stay with the reference's controller.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

CAMERA_INTRINSIC = [0.05, 100, 1, 640, 480]        # env/sapien_envs/open_cabinet.py:20 (near, far, ?, width, height)
_MUG_PERM = [0, 2, 4, 6, 1, 3, 5, 7]               # rl_pose.py:220-221


class ControlInterface:
    def __init__(self, num_envs: int, pose_estimator, max_steps: int, device=None):
        self.num_envs = int(num_envs)
        self.estimator = pose_estimator
        self.max_steps = int(max_steps)            # the reference passes cfg["controller"]["max_steps"] + 1
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.H, self.W = CAMERA_INTRINSIC[-1], CAMERA_INTRINSIC[-2]
        self.lib = _lib.load()
        self.reset_queue()

    # ------------------------------------------------------------------ rl_pose.py:89-101
    def reset_queue(self):
        T, N, H, W, dev = self.max_steps, self.num_envs, self.H, self.W, self.device
        z = lambda *shape, dtype=torch.float64: torch.zeros(*shape, dtype=dtype, device=dev)
        self.image_queue = z(T, N, H, W, 3, dtype=torch.float32)
        self.mask_queue = z(T, N, H, W, dtype=torch.uint8)
        self.bbox_queue = z(T, N, 4)
        self.pose_queue = z(T, N, 7)
        self.intrinsic_queue = z(T, N, 3, 3)
        self.extrinsic_queue = z(T, N, 4, 4)
        self.available = z(T, N)
        self.pred_bbox = z(T, N, 8, 3)
        self.gt_bbox = z(T, N, 8, 3)
        self.available_num = z(N, dtype=torch.int32)
        self.accumulate_steps = 0

    def _dev(self, x, dtype):
        return torch.as_tensor(x).to(device=self.device, dtype=dtype)

    # ------------------------------------------------------------------ rl_pose.py:118-149
    def add_view(self, image, cam_pose):
        k = self.accumulate_steps % self.max_steps
        cam = image["camera0"]
        mask = (self._dev(cam["Mask"], torch.uint8) != 0).to(torch.uint8).contiguous()
        self.image_queue[k] = self._dev(cam["Color"], torch.float32)
        self.mask_queue[k] = mask
        self.pose_queue[k] = self._dev(cam_pose, torch.float64)
        self.intrinsic_queue[k] = self._dev(cam["Intrinsic"], torch.float64)
        self.extrinsic_queue[k] = self._dev(cam["Extrinsic"], torch.float64)
        ext = torch.empty(self.num_envs, 4, dtype=torch.int32, device=self.device)
        cnt = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.rgbm_mask_extent(_lib.ptr(mask), self.num_envs, self.H, self.W, _lib.ptr(ext), _lib.ptr(cnt),
                                             _lib.stream_ptr()), "rgbm_mask_extent")
        # the reference marks EVERY env available as soon as any env's mask has a pixel (rl_pose.py:132): kept as is
        anyone = (cnt.sum() > 0).to(torch.float64)
        self.available[k] = anyone
        self.available_num += anyone.to(torch.int32)
        scale = torch.tensor([self.H, self.W, self.H, self.W], dtype=torch.float64, device=self.device)
        self.bbox_queue[k] = ext.to(torch.float64) / scale

    # ------------------------------------------------------------------ rl_pose.py:151-155
    def add_bbox(self, pred_bbox, gt_bbox):
        k = self.accumulate_steps % self.max_steps
        self.pred_bbox[k] = self._dev(pred_bbox, torch.float64)
        self.gt_bbox[k] = self._dev(gt_bbox, torch.float64)

    def _time(self):
        t = torch.zeros(self.num_envs, self.max_steps, dtype=torch.float32, device=self.device)
        t[:, self.accumulate_steps - 1] = 1.0
        return t

    # ------------------------------------------------------------------ rl_pose.py:157-171
    def get_state(self):
        centre = (self.gt_bbox[:, :, 0] + self.gt_bbox[:, :, 6]) / 2
        cur = torch.cat((self.pose_queue, self.bbox_queue, centre), dim=-1).to(torch.float32)
        return torch.cat((cur.permute(1, 0, 2).reshape(self.num_envs, -1), self._time()), dim=-1)

    # ------------------------------------------------------------------ rl_pose.py:173-187
    def get_observation(self):
        cur = torch.cat((self.pose_queue, self.bbox_queue), dim=-1).to(torch.float32)
        return torch.cat((cur.permute(1, 0, 2).reshape(self.num_envs, -1), self._time()), dim=-1)

    # ------------------------------------------------------------------ rl_pose.py:189-208
    def select_views(self):
        """Queue slots of the two views `get_estimation` hands to the estimator: the reference overwrites slot `used % 2` with
        every available view in queue order, so slot s ends up with the LAST available view whose rank has parity s."""
        avail = self.available > 0                                       # [T,N]
        rank = torch.cumsum(avail.to(torch.int64), dim=0) - 1            # rank of each available view per env
        steps = torch.arange(self.max_steps, device=self.device).view(-1, 1)
        idx, has = [], []
        for s in (0, 1):
            hit = avail & ((rank % 2) == s)
            last = torch.where(hit, steps, torch.full_like(steps, -1)).max(dim=0).values
            has.append(last >= 0)
            idx.append(last.clamp(min=0))
        return idx, has

    def _gather(self, queue, idx, has):
        env = torch.arange(self.num_envs, device=self.device)
        out = queue[idx, env]
        keep = has.view(-1, *([1] * (out.dim() - 1)))
        return torch.where(keep, out, torch.zeros_like(out))             # envs without such a view: zeros, like the reference

    # ------------------------------------------------------------------ rl_pose.py:189-223
    def get_estimation(self):
        idx, has = self.select_views()
        K = [self._gather(self.intrinsic_queue, idx[s], has[s]) for s in (0, 1)]
        E = [self._gather(self.extrinsic_queue, idx[s], has[s]) for s in (0, 1)]
        rgb = [self._gather(self.image_queue, idx[s], has[s]) for s in (0, 1)]
        mask = [self._gather(self.mask_queue, idx[s], has[s]) for s in (0, 1)]
        if hasattr(self.estimator, "estimate_device"):
            bbox = self.estimator.estimate_device(K[0], rgb[0], mask[0], E[0], rgb[1], mask[1], E[1])
        else:                                                            # numpy-in / numpy-out estimators (interface_v5.py:213)
            bbox = torch.from_numpy(np.asarray(self.estimator.estimate(
                K[0].cpu().numpy(), rgb[0].cpu().numpy(), mask[0].cpu().numpy(), E[0].cpu().numpy(), rgb[1].cpu().numpy(),
                mask[1].cpu().numpy(), E[1].cpu().numpy()))).to(self.device)
        if self.estimator.cfg["task_name"] == "mugs":
            bbox = bbox[:, _MUG_PERM]
        return bbox
