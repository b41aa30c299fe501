"""ORACLE (test infrastructure, never shipped or measured as product).

PyTorch-CPU fp32 restatement of the reference PPO arithmetic:
  * ActorCritic.act / evaluate            /root/reference/algo/ppo/ppo/module.py:73-107
  * RolloutStorage.compute_returns (GAE)  /root/reference/algo/ppo/ppo/storage.py:50-64
  * PPO.update                            /root/reference/algo/ppo/ppo/ppo.py:449-534
The Gaussian is written in closed form: the reference builds
`MultivariateNormal(mu, scale_tril=diag(exp(log_std)**2))` (module.py:76-77), i.e. the
sampling std is exp(2*log_std).

Parity pin: `tests/test_oracle_golden.py` against `tests/golden/ppo_*.npz` produced by
`tools/make_goldens.py` from the reference classes themselves.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


def mlp(x, sd, net, n_layers=4):
    for i in range(n_layers):
        x = F.linear(x, sd[f"{net}.{2 * i}.weight"], sd[f"{net}.{2 * i}.bias"])
        if i + 1 < n_layers:
            x = F.elu(x)
    return x


def gaussian_logp(actions, mu, log_std):
    k = mu.shape[-1]
    return (-((actions - mu) ** 2) / (2.0 * torch.exp(4.0 * log_std))).sum(-1) \
        - 2.0 * log_std.sum() - 0.5 * k * LOG_2PI


def act(sd, obs, noise):
    """module.py:73-87 with the N(0,1) draw supplied: a = mu + exp(2*log_std)*eps."""
    mu = mlp(obs, sd, "actor")
    ls = sd["log_std"]
    a = mu + torch.exp(2.0 * ls) * noise
    logp = gaussian_logp(a, mu, ls)
    v = mlp(obs, sd, "critic")
    return a, logp, v, mu, ls.repeat(mu.shape[0], 1)


def evaluate(sd, obs, actions):
    """module.py:93-107."""
    mu = mlp(obs, sd, "actor")
    ls = sd["log_std"]
    k = mu.shape[-1]
    logp = gaussian_logp(actions, mu, ls)
    ent = (0.5 * k * (1.0 + LOG_2PI) + 2.0 * ls.sum()).expand(mu.shape[0])
    v = mlp(obs, sd, "critic")
    return logp, ent, v, mu, ls.repeat(mu.shape[0], 1)


def compute_returns(rewards, dones, values, last_values, gamma, lam):
    """storage.py:50-64.  rewards/values [T,N,1], dones uint8 [T,N,1], last_values [N,1]."""
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    adv = 0
    for t in reversed(range(T)):
        nv = last_values if t == T - 1 else values[t + 1]
        m = 1.0 - dones[t].float()
        delta = rewards[t] + m * gamma * nv - values[t]
        adv = delta + m * gamma * lam * adv
        returns[t] = adv + values[t]
    a = returns - values
    a = (a - a.mean()) / (a.std() + 1e-8)          # unbiased std (storage.py:64)
    return returns, a


def ppo_update(sd, roll, returns, advantages, cfg, step_size, adam_state=None, log=None):
    """ppo.py:449-534 on a recorded rollout.  `sd` holds leaf tensors updated in place.

    roll: observations [T,N,60], actions [T,N,12], values, actions_log_prob [T,N,1], mu, sigma [T,N,12].
    Returns (mean_value_loss, mean_surrogate_loss, step_size, adam_state).
    """
    names = list(sd.keys())
    params = [sd[k].detach().clone().requires_grad_(True) for k in names]
    P = dict(zip(names, params))
    if adam_state is None:
        adam_state = {"t": 0, "m": [torch.zeros_like(p) for p in params], "v": [torch.zeros_like(p) for p in params]}
    T, N = roll["observations"].shape[:2]
    nmb = cfg["num_mini_batches"]
    mb = (T * N) // nmb
    flat = {k: roll[k].reshape(T * N, -1) for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    ret = returns.reshape(T * N, 1)
    advf = advantages.reshape(T * N, 1)
    clip = cfg["clip_range"]
    mvl = msl = 0.0
    for _ in range(cfg["num_learning_epochs"]):
        for b in range(nmb):
            sl = slice(b * mb, (b + 1) * mb)
            logp, ent, v, mu, sig = evaluate(P, flat["observations"][sl], flat["actions"][sl])
            osig, omu = flat["sigma"][sl], flat["mu"][sl]
            kl = torch.sum(sig - osig + (torch.square(osig.exp()) + torch.square(omu - mu))
                           / (2.0 * torch.square(sig.exp())) - 0.5, dim=-1)
            kl_mean = kl.mean()
            if cfg.get("desired_kl") is not None and cfg.get("schedule", "adaptive") == "adaptive":      # ppo.py:480
                if kl_mean > cfg["desired_kl"] * 2.0:
                    step_size = max(cfg["min_lr"], step_size / 1.5)
                elif kl_mean < cfg["desired_kl"] / 2.0 and kl_mean > 0.0:
                    step_size = min(cfg["max_lr"], step_size * 1.5)
            ratio = torch.exp(logp - flat["actions_log_prob"][sl].squeeze(1))
            a = advf[sl].squeeze(1)
            surr = torch.max(-a * ratio, -a * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)).mean()
            tv = flat["values"][sl]
            vc = tv + (v - tv).clamp(-clip, clip)
            vl = torch.max((v - ret[sl]).pow(2), (vc - ret[sl]).pow(2)).mean()
            loss = surr + cfg["value_loss_coef"] * vl - cfg["entropy_coef"] * ent.mean()
            grads = torch.autograd.grad(loss, params, allow_unused=True)
            grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
            # clip_grad_norm_(max_norm) then Adam (beta .9/.999, eps 1e-8)
            tot = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
            coef = torch.clamp(cfg["max_grad_norm"] / (tot + 1e-6), max=1.0)
            adam_state["t"] += 1
            t = adam_state["t"]
            with torch.no_grad():
                for i, (p, g) in enumerate(zip(params, grads)):
                    g = g * coef
                    adam_state["m"][i].mul_(0.9).add_(g, alpha=0.1)
                    adam_state["v"][i].mul_(0.999).addcmul_(g, g, value=0.001)
                    bc1 = 1 - 0.9 ** t
                    bc2 = 1 - 0.999 ** t
                    denom = (adam_state["v"][i].sqrt() / math.sqrt(bc2)).add_(1e-8)
                    p.addcdiv_(adam_state["m"][i], denom, value=-step_size / bc1)
            mvl += vl.item()
            msl += surr.item()
            if log is not None:
                log.append((surr.item(), vl.item(), kl_mean.item(), step_size, tot.item()))
    n_up = cfg["num_learning_epochs"] * nmb
    for k, p in zip(names, params):
        sd[k] = p.detach()
    return mvl / n_up, msl / n_up, step_size, adam_state
