"""ActorCritic on a flat fp32 parameter vector, evaluated by the HIP policy kernels.

Mirrors `/root/reference/algo/ppo/ppo/module.py:8-107`: same constructor arguments, `act`, `act_inference`,
`evaluate`, `state_dict` key names (`log_std`, `actor.{0,2,4,6}.{weight,bias}`, `critic.{0,2,4,6}.{weight,bias}`),
orthogonal initialisation with the reference's gains, and the reference's Gaussian
(`scale_tril = diag(exp(log_std)**2)`, module.py:76-77).  Sampling noise is drawn from torch's generator
(`torch.randn`) so RNG stays with the caller; everything else runs in `rgbm_policy_forward`.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


def get_activation(act_name):
    if act_name != "elu":
        raise NotImplementedError(f"the HIP policy kernels implement the shipped cfg's ELU only (got {act_name!r})")
    return nn.ELU()


class ActorCritic:
    def __init__(self, obs_shape, states_shape, actions_shape, initial_std, model_cfg, asymmetric=False):
        if asymmetric:
            raise NotImplementedError("asymmetric critic is not used by cfg/controller/rl.yaml (asymmetric: False)")
        self.asymmetric = asymmetric
        if model_cfg is None:
            raise NotImplementedError("model_cfg=None (256x3 SELU) is not on the shipped path")
        a_h, c_h = list(model_cfg["pi_hid_sizes"]), list(model_cfg["vf_hid_sizes"])
        get_activation(model_cfg["activation"])
        if a_h != c_h or len(a_h) != 3:
            raise NotImplementedError("actor and critic must share three hidden sizes (cfg/controller/rl.yaml:30-31)")
        self.obs_dim, self.act_dim = int(obs_shape[0]), int(actions_shape[0])
        self.hidden = a_h
        dims = [self.obs_dim] + a_h
        # ---- layout of the flat vector = the reference's state_dict order ----
        self.keys = OrderedDict()
        off = 0
        self.keys["log_std"] = (off, (self.act_dim,))
        off += self.act_dim
        L = _lib.PolicyLayout()
        for k, d in enumerate(dims + [self.act_dim]):
            L.dims[k] = d
        L.log_std = 0
        for net, name, out_dim in ((0, "actor", self.act_dim), (1, "critic", 1)):
            for l in range(4):
                i, o = dims[l], (dims[l + 1] if l < 3 else out_dim)
                self.keys[f"{name}.{2 * l}.weight"] = (off, (o, i))
                L.w[net][l] = off
                off += o * i
                self.keys[f"{name}.{2 * l}.bias"] = (off, (o,))
                L.b[net][l] = off
                off += o
        L.total = off
        self.layout = L
        self.total = off
        # ---- initialisation exactly like the reference: nn.Linear defaults, then orthogonal_ with its gains ----
        actor = [nn.Linear(dims[l], dims[l + 1] if l < 3 else self.act_dim) for l in range(4)]
        critic = [nn.Linear(dims[l], dims[l + 1] if l < 3 else 1) for l in range(4)]
        log_std = np.log(initial_std) * torch.ones(self.act_dim)
        for mods, gains in ((actor, [np.sqrt(2)] * 3 + [0.01]), (critic, [np.sqrt(2)] * 3 + [1.0])):
            for m, gain in zip(mods, gains):
                torch.nn.init.orthogonal_(m.weight, gain=gain)
        flat = torch.zeros(off, dtype=torch.float32)
        flat[: self.act_dim] = log_std
        for name, mods in (("actor", actor), ("critic", critic)):
            for l, m in enumerate(mods):
                o, shp = self.keys[f"{name}.{2 * l}.weight"]
                flat[o:o + m.weight.numel()] = m.weight.detach().reshape(-1)
                o, shp = self.keys[f"{name}.{2 * l}.bias"]
                flat[o:o + m.bias.numel()] = m.bias.detach()
        self.flat = flat
        self.device = torch.device("cpu")
        self.training = True

    # ---- nn.Module-like plumbing used by PPO / RLPoseController ----
    def to(self, device):
        self.device = torch.device(device)
        self.flat = self.flat.to(self.device).contiguous()
        return self

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def parameters(self):
        return [self.flat]

    @property
    def log_std(self):
        return self.flat[: self.act_dim]

    def state_dict(self):
        return OrderedDict((k, self.flat[o:o + int(np.prod(s))].view(*s).clone()) for k, (o, s) in self.keys.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.keys if k not in sd]
        extra = [k for k in sd if k not in self.keys]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
        for k, (o, s) in self.keys.items():
            if k in sd:
                v = torch.as_tensor(sd[k]).to(device=self.flat.device, dtype=torch.float32)
                if tuple(v.shape) != tuple(s):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(s)}")
                self.flat[o:o + v.numel()] = v.reshape(-1)

    def forward(self):
        raise NotImplementedError

    # ---- the three entry points of module.py:73-107 ----
    def _run(self, mode, observations, noise=None, actions=None):
        if self.flat.device.type != "cuda":
            raise _lib.RgbmError("ActorCritic runs on the HIP policy kernels only: move it to a cuda device (no CPU fallback)")
        lib = _lib.load()
        obs = observations.to(device=self.flat.device, dtype=torch.float32).contiguous()
        n = obs.shape[0]
        dev = self.flat.device
        mu = torch.empty(n, self.act_dim, device=dev)
        logp = torch.empty(n, device=dev)
        value = torch.empty(n, 1, device=dev)
        if mode == 0:
            actions = torch.empty(n, self.act_dim, device=dev)
        elif mode == 2:
            actions = actions.to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(lib.rgbm_policy_forward(_lib.ptr(self.flat), C.byref(self.layout), n, mode, _lib.ptr(obs), _lib.ptr(noise),
                                           _lib.ptr(actions), _lib.ptr(logp), _lib.ptr(value), _lib.ptr(mu),
                                           _lib.stream_ptr()), "rgbm_policy_forward")
        return actions, logp, value, mu

    def act(self, observations, states, noise=None):
        n = observations.shape[0]
        if noise is None:
            noise = torch.randn(n, self.act_dim, device=self.flat.device)
        noise = noise.to(device=self.flat.device, dtype=torch.float32).contiguous()
        actions, logp, value, mu = self._run(0, observations, noise=noise)
        return actions, logp, value, mu, self.log_std.repeat(n, 1).detach()

    def act_inference(self, observations):
        return self._run(1, observations)[3]

    def evaluate(self, observations, states, actions, contrastive=False):
        n = observations.shape[0]
        _, logp, value, mu = self._run(2, observations, actions=actions)
        k = self.act_dim
        entropy = (0.5 * k * (1.0 + np.log(2 * np.pi)) + 2.0 * self.log_std.sum()).expand(n)
        return logp, entropy, value, mu, self.log_std.repeat(n, 1), 0
