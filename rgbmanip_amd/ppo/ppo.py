"""PPO trainer for the global-scheduling policy on MI355X.

Same public surface as `/root/reference/algo/ppo/ppo/ppo.py` (`PPO(vec_env, learn_cfg)`, `run`, `update`, `save`, `load`,
`test`, `eval`, `play`, attributes `.actor_critic`, `.device`, `.storage`, `.step_size`) and the same cfg keys
(`cfg/controller/rl.yaml`), but the learn phase is four HIP launches per minibatch with no host synchronisation:
`rgbm_ppo_minibatch_fwd_bwd` (forward + clipped losses + backward + deterministic reduction), an optional RCCL all-reduce
of the flat gradient when several ranks train one policy, and `rgbm_ppo_clip_adam` (grad-norm clip, KL-adaptive learning
rate, Adam) — the reference does ~60 small launches and two `.item()` syncs per minibatch (ppo.py:472-528).

Multi-GPU: one process per GPU, each rank steps its own slice of the environments; minibatch b of a
rank is its local rows [b*n/4, (b+1)*n/4), gradients are summed over ranks and divided by the world size, the KL / loss
statistics ride in the same buffer, so every rank applies the identical update and learning-rate decision.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import time
from collections import deque

import numpy as np
import torch

from .. import _lib, dist_utils
from ..spaces import Space, concat_spaces
from .module import ActorCritic
from .storage import RolloutStorage


def concat_tensor_dict(d):
    """`utils/tools.py:166-178`."""
    parts = []
    for v in d.values():
        if isinstance(v, dict):
            parts.append(concat_tensor_dict(v))
        elif isinstance(v, torch.Tensor):
            parts.append(v)
        elif isinstance(v, np.ndarray):
            parts.append(torch.from_numpy(v))
        else:
            raise TypeError(v)
    return torch.cat(parts, dim=-1).float()


def prepare_obs(obs):
    """Tensor observations pass through; dict observations lose their "image" entry and are concatenated (ppo.py:26-34)."""
    if isinstance(obs, torch.Tensor):
        return obs, None
    image = obs.pop("image")
    return concat_tensor_dict(obs), image


class PPO:
    def __init__(self, vec_env, learn_cfg: dict, process_group=None):
        for name in ("observation_space", "state_space", "action_space"):
            if not isinstance(getattr(vec_env, name), Space):
                raise TypeError(f"vec_env.{name} must be a gym Space")
        self.observation_space = concat_spaces(vec_env.observation_space)
        self.state_space = concat_spaces(vec_env.state_space)
        self.action_space = vec_env.action_space
        lc = learn_cfg["learn"]
        self.eval_interval, self.eval_round, self.do_eval = lc["eval_interval"], lc["eval_round"], lc["eval"]
        self.device = lc["device"]
        self.asymmetric = lc["asymmetric"]
        self.desired_kl = lc["desired_kl"]
        self.lr_upper, self.lr_lower = float(lc["max_lr"]), float(lc["min_lr"])
        self.schedule = lc["schedule"]
        self.step_size = float(lc["learning_rate"])
        self.learning_rate = float(lc["learning_rate"])
        self.sampler = lc["sampler"]
        self.num_envs = vec_env.num_envs
        self.reset = lc["reset"]
        self.contrastive = lc["contrastive"]
        self.clip_param = lc["clip_range"]
        self.num_learning_epochs = lc["num_learning_epochs"]
        self.num_mini_batches = lc["num_mini_batches"]
        self.num_transitions_per_env = lc["num_transitions_per_env"]
        self.num_transitions_eval = lc["num_transitions_eval"]
        self.value_loss_coef = lc["value_loss_coef"]
        self.entropy_coef = lc["entropy_coef"]
        self.gamma, self.lam = lc["gamma"], lc["lam"]
        self.max_grad_norm = lc["max_grad_norm"]
        self.use_clipped_value_loss = lc["use_clipped_value_loss"]
        if not self.use_clipped_value_loss or self.contrastive:
            raise NotImplementedError("the HIP update implements the shipped cfg: clipped value loss, no contrastive term")
        if learn_cfg["policy"]["actor_critic_class"] != "ActorCritic":
            raise NotImplementedError(learn_cfg["policy"]["actor_critic_class"])
        self.process_group = process_group
        # equal shards are a requirement of the update (one all-reduce per minibatch, every rank's minibatch mean weighs the same)
        self.world = dist_utils.require_uniform(self.num_envs, "PPO: num_envs per rank", self.device, process_group)
        self.rank = torch.distributed.get_rank(process_group) if self.world > 1 else 0

        self.vec_env = vec_env
        self.actor_critic = ActorCritic(self.observation_space.shape, self.state_space.shape, self.action_space.shape,
                                        lc["init_noise_std"], learn_cfg["policy"], asymmetric=self.asymmetric)
        self.actor_critic.to(self.device)
        if self.world > 1:      # rank 0's initialisation is the policy every rank starts from
            torch.distributed.broadcast(self.actor_critic.flat, src=0, group=process_group)
        self.storage = RolloutStorage(self.num_envs, self.num_transitions_per_env, self.observation_space.shape,
                                      self.state_space.shape, self.action_space.shape, self.device, self.sampler)
        n = self.actor_critic.total
        dev = self.actor_critic.flat.device
        self._exp_avg = torch.zeros(n, device=dev)
        self._exp_avg_sq = torch.zeros(n, device=dev)
        self._grads = torch.zeros(n + 4, device=dev)
        self._opt_state = torch.zeros(48, dtype=torch.uint8, device=dev)
        self._partial = None
        self._write_opt_state(t=0, lr=self.step_size)

        self.log_dir = lc["log_dir"]
        self.print_log = lc["print_log"]
        self.writer = None
        if self.rank == 0:      # one writer / one checkpoint file per job, not one per rank
            try:  # TensorBoard is optional (absent in the build image)
                from torch.utils.tensorboard import SummaryWriter  # type: ignore
                self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
            except Exception:  # noqa: BLE001
                self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0.0
        self.is_testing = lc["testing"]
        self.current_learning_iteration = 0
        self.exp_name = lc["exp_name"]
        self.save_dir = lc["save_dir"]
        os.makedirs(self.save_dir, exist_ok=True)
        if learn_cfg.get("load", "") != "":
            self.load(learn_cfg["load"])

    # ------------------------------------------------------------------ optimiser state record (48 bytes on the device)
    def _write_opt_state(self, t, lr):
        rec = struct.pack("<iifffxxxxdd", int(t), 0, float(lr), 0.0, 0.0, 0.0, 0.0)
        assert len(rec) == 40
        rec = rec + b"\0" * 8
        self._opt_state.copy_(torch.frombuffer(bytearray(rec), dtype=torch.uint8))

    def _read_opt_state(self):
        raw = bytes(self._opt_state.cpu().numpy().tobytes())
        t, n_up, lr, kl, norm = struct.unpack_from("<iifff", raw, 0)
        s_surr, s_vl = struct.unpack_from("<dd", raw, 24)
        return dict(t=t, n_updates=n_up, lr=lr, last_kl=kl, last_norm=norm, sum_surr=s_surr, sum_vloss=s_vl)

    # ------------------------------------------------------------------ checkpoints (state_dict I/O, iteration from file name)
    def test(self, path):
        self.load(path)
        self.actor_critic.eval()

    def load(self, path):
        self.actor_critic.load_state_dict(torch.load(path, map_location="cpu"))
        self.current_learning_iteration = int(path.split("_")[-1].split(".")[0])
        self.actor_critic.train()

    def save(self, path):
        """Rank 0 writes (every rank holds the same parameters); the file appears atomically."""
        if self.rank != 0:
            return
        tmp = f"{path}.tmp{os.getpid()}"
        torch.save({k: v.cpu() for k, v in self.actor_critic.state_dict().items()}, tmp)
        os.replace(tmp, path)

    # ------------------------------------------------------------------ evaluation helpers (ppo.py:142-199)
    def play(self):
        obs, _ = prepare_obs(self.vec_env.reset())
        obs = obs.to(self.device)
        for _ in range(self.num_transitions_eval):
            nxt, _, _, _ = self.vec_env.step(self.actor_critic.act_inference(obs))
            obs = prepare_obs(nxt)[0].to(self.device)

    def eval(self):
        total_reward = torch.zeros((self.num_envs,), device=self.device)
        total_success = torch.zeros((self.num_envs,), device=self.device)
        for _ in range(self.eval_round):
            obs = prepare_obs(self.vec_env.reset())[0].to(self.device)
            for _ in range(self.num_transitions_eval):
                nxt, rews, _, infos = self.vec_env.step(self.actor_critic.act_inference(obs))
                obs = prepare_obs(nxt)[0].to(self.device)
                total_reward += rews.to(self.device)
                total_success += infos["successes"].to(self.device)
        reward = (total_reward.mean() / self.num_transitions_per_env / self.eval_round).item()
        success = (total_success.mean() / self.eval_round).item()
        return reward, success

    # ------------------------------------------------------------------ rollout + learn loop (ppo.py:204-306)
    def run(self, num_learning_iterations, log_interval=1, save_interval=1):
        cur_obs = prepare_obs(self.vec_env.reset())[0].to(self.device)
        cur_states = prepare_obs(self.vec_env.get_state())[0].to(self.device)
        if self.is_testing:
            return self.eval()
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        ep_reward = torch.zeros(self.num_envs, device=self.device)
        ep_len = torch.zeros(self.num_envs, device=self.device)
        # finished-episode records of one iteration stay on the device ([T, N] values + a boolean "an episode ended here" mask)
        # and reach the host in ONE copy after the rollout: the reference's per-step `.cpu()` (ppo.py:262-271) would
        # synchronise the stream 16 times per iteration.
        # The reference never clears its `reward_sum` / `episode_length` lists (ppo.py:222-223, 268-269) and extends the two
        # 100-entry deques with the CUMULATIVE lists every iteration (ppo.py:278-279): until 100 episodes have finished the
        # deques hold duplicates, which `Train/mean_reward` / `Train/mean_episode_length` average over.  Reproduced here; only
        # the last 100 entries of the cumulative lists can ever reach a maxlen-100 deque, so only those are kept.
        # (On several ranks each rank keeps the statistics of its own envs; rank 0 logs its shard.)
        T_ = self.num_transitions_per_env
        fin_rew = torch.empty(T_, self.num_envs, device=self.device) if self.print_log else None
        fin_len = torch.empty(T_, self.num_envs, device=self.device) if self.print_log else None
        fin_mask = torch.empty(T_, self.num_envs, dtype=torch.bool, device=self.device) if self.print_log else None
        cum_rew, cum_len = [], []
        self.last_fps = 0.0
        for it in range(self.current_learning_iteration, num_learning_iterations):
            start = time.time()
            ep_infos = []
            for t_step in range(self.num_transitions_per_env):
                actions, logp, values, mu, sigma = self.actor_critic.act(cur_obs, cur_states)
                nxt_obs, rews, dones, infos = self.vec_env.step(actions)
                nxt_obs = prepare_obs(nxt_obs)[0].to(self.device)
                nxt_states = prepare_obs(self.vec_env.get_state())[0].to(self.device)
                rews, dones = rews.to(self.device), dones.to(self.device)
                self.storage.add_transitions(cur_obs, cur_states, actions, rews, dones, values, logp, mu, sigma)
                cur_obs, cur_states = nxt_obs, nxt_states
                ep_infos.append(infos)
                if self.print_log:
                    ep_reward += rews.float().view(-1)
                    ep_len += 1
                    fin = dones.view(-1) > 0
                    fin_mask[t_step] = fin
                    fin_rew[t_step] = ep_reward
                    fin_len[t_step] = ep_len
                    ep_reward = torch.where(fin, torch.zeros_like(ep_reward), ep_reward)
                    ep_len = torch.where(fin, torch.zeros_like(ep_len), ep_len)
            _, _, last_values, _, _ = self.actor_critic.act(cur_obs, cur_states)     # extra stochastic act, as ppo.py:287
            torch.cuda.synchronize()
            collection_time = time.time() - start
            if self.print_log:      # same order as the reference's per-step extends: by step, then by env
                cum_rew = (cum_rew + fin_rew[fin_mask].cpu().tolist())[-100:]
                cum_len = (cum_len + fin_len[fin_mask].cpu().tolist())[-100:]
                rewbuffer.extend(cum_rew)
                lenbuffer.extend(cum_len)
            mean_trajectory_length, mean_reward = self.storage.get_statistics()
            start = time.time()
            self.storage.compute_returns(last_values[:self.num_envs], self.gamma, self.lam, process_group=self.process_group)
            mean_value_loss, mean_surrogate_loss = self.update(it)
            self.storage.clear()
            learn_time = time.time() - start
            self.last_collection_time, self.last_learn_time = collection_time, learn_time
            self.tot_timesteps += self.num_transitions_per_env * self.num_envs * self.world
            self.tot_time += collection_time + learn_time
            self.last_fps = self.num_transitions_per_env * self.num_envs * self.world / (collection_time + learn_time)
            if self.print_log and it % log_interval == 0:
                self.log(dict(it=it, num_learning_iterations=num_learning_iterations, collection_time=collection_time,
                              learn_time=learn_time, mean_value_loss=mean_value_loss, mean_surrogate_loss=mean_surrogate_loss,
                              mean_reward=float(mean_reward), mean_trajectory_length=float(mean_trajectory_length),
                              rewbuffer=rewbuffer, lenbuffer=lenbuffer, ep_infos=ep_infos))
            if it % save_interval == 0:
                self.save(os.path.join(self.save_dir, f"model_{it}.pt"))
        self.save(os.path.join(self.save_dir, f"model_{num_learning_iterations}.pt"))

    def log(self, locs, width=80, pad=35):
        fps = int(self.last_fps)
        lines = [f"Learning iteration {locs['it']}/{locs['num_learning_iterations']}",
                 f"{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)",
                 f"{'Value function loss:':>{pad}} {locs['mean_value_loss']:.4f}",
                 f"{'Surrogate loss:':>{pad}} {locs['mean_surrogate_loss']:.4f}",
                 f"{'Mean action noise std:':>{pad}} {self.actor_critic.log_std.exp().mean().item():.2f}",
                 f"{'Mean reward/step:':>{pad}} {locs['mean_reward']:.2f}",
                 f"{'Mean episode length/episode:':>{pad}} {locs['mean_trajectory_length']:.2f}",
                 f"{'Learning Rate:':>{pad}} {self.step_size}"]
        if len(locs["rewbuffer"]) > 0:
            lines.insert(5, f"{'Mean reward:':>{pad}} {np.mean(locs['rewbuffer']):.2f}")
            lines.insert(6, f"{'Mean episode length:':>{pad}} {np.mean(locs['lenbuffer']):.2f}")
        ep_scalars = self.episode_scalars(locs.get("ep_infos"))
        for key, val in ep_scalars.items():
            label = "Mean episode " + key[len("Episode/"):].replace("_train", " train") + ":"
            lines.append(f"{label:>{pad}} {val:.4f}")
        self.last_scalars = dict(ep_scalars)
        self.last_scalars.update({"Loss/value_function": locs["mean_value_loss"], "Loss/surrogate": locs["mean_surrogate_loss"],
                                  "Policy/mean_noise_std": self.actor_critic.log_std.exp().mean().item(), "Policy/lr": self.step_size,
                                  "Train2/mean_reward/step": locs["mean_reward"],
                                  "Train2/mean_episode_length/episode": locs["mean_trajectory_length"]})
        if len(locs["rewbuffer"]) > 0:
            self.last_scalars["Train/mean_reward"] = float(np.mean(locs["rewbuffer"]))
            self.last_scalars["Train/mean_episode_length"] = float(np.mean(locs["lenbuffer"]))
        if self.writer is not None:
            for key, val in self.last_scalars.items():
                self.writer.add_scalar(key, val, locs["it"])
        if self.rank == 0:
            print("\n".join(["#" * width] + lines), flush=True)

    def episode_scalars(self, ep_infos):
        """`Episode/<key>_train` = mean over the iteration's steps and envs of every entry of the env's info dict — the
        reward terms of `ControlInterface.get_reward` (ppo.py:364-384).  All means are formed on the device and fetched in
        one copy.  As in the reference, a key named "success_rate" is averaged over an EMPTY tensor (its loop sits in the
        else branch, ppo.py:368-380): both of its scalars are NaN."""
        if not ep_infos:
            return {}
        keys = list(ep_infos[0])
        plain = [k for k in keys if k != "success_rate"]
        out = {}
        if plain:
            means = torch.stack([torch.cat([torch.as_tensor(ei[k]).to(self.device).float().reshape(-1) for ei in ep_infos]).mean()
                                 for k in plain]).cpu().tolist()
            out.update({f"Episode/{k}_train": float(m) for k, m in zip(plain, means)})
        if "success_rate" in keys:
            out["Episode/worst_50.0%_success_rate_train"] = float("nan")
            out["Episode/success_rate_train"] = float("nan")
        return out

    # ------------------------------------------------------------------ the learn phase (ppo.py:449-534)
    def update(self, it):
        lib = _lib.load()
        st = self.storage
        ac = self.actor_critic
        if ac.flat.device.type != "cuda":
            raise _lib.RgbmError("PPO.update runs on the HIP kernels: use a cuda device (no CPU fallback)")
        T, N = st.num_transitions_per_env, st.num_envs
        flat = {k: getattr(st, k).reshape(T * N, -1) for k in ("observations", "actions", "values", "returns",
                                                               "actions_log_prob", "advantages", "mu", "sigma")}
        batches = st.mini_batch_generator(self.num_mini_batches)
        mb = len(batches[0])
        need = C.c_size_t()
        _lib.check(lib.rgbm_ppo_partial_floats(C.byref(ac.layout), mb, C.byref(need)), "rgbm_ppo_partial_floats")
        if self._partial is None or self._partial.numel() < need.value:
            self._partial = torch.empty(need.value, device=ac.flat.device)
        s0 = self._read_opt_state()
        adaptive = int(self.desired_kl is not None and self.schedule == "adaptive")
        stream = _lib.stream_ptr()
        for _ in range(self.num_learning_epochs):
            for idx in batches:
                if isinstance(idx, range):
                    sl = {k: v[idx.start:idx.stop] for k, v in flat.items()}
                else:
                    ii = torch.as_tensor(idx, device=ac.flat.device)
                    sl = {k: v[ii].contiguous() for k, v in flat.items()}
                _lib.check(lib.rgbm_ppo_minibatch_fwd_bwd(
                    _lib.ptr(ac.flat), C.byref(ac.layout), mb, _lib.ptr(sl["observations"]), _lib.ptr(sl["actions"]),
                    _lib.ptr(sl["actions_log_prob"]), _lib.ptr(sl["advantages"]), _lib.ptr(sl["returns"]), _lib.ptr(sl["values"]),
                    _lib.ptr(sl["mu"]), _lib.ptr(sl["sigma"]), float(self.clip_param), float(self.value_loss_coef),
                    float(self.entropy_coef), _lib.ptr(self._partial), _lib.ptr(self._grads), stream), "rgbm_ppo_minibatch_fwd_bwd")
                # sum over ranks of (gradient of the local minibatch mean | loss / KL sums); the optimiser applies 1/world
                inv_world = dist_utils.average_flat_gradient(self._grads, self.process_group)
                _lib.check(lib.rgbm_ppo_clip_adam(
                    _lib.ptr(ac.flat), _lib.ptr(self._grads), _lib.ptr(self._exp_avg), _lib.ptr(self._exp_avg_sq),
                    _lib.ptr(self._opt_state), C.byref(ac.layout), inv_world, float(self.max_grad_norm),
                    float(self.desired_kl or 0.0), self.lr_lower, self.lr_upper, adaptive, stream), "rgbm_ppo_clip_adam")
        s1 = self._read_opt_state()            # the only host sync of the learn phase
        n_up = max(s1["n_updates"] - s0["n_updates"], 1)
        self.step_size = s1["lr"]
        self.last_kl = s1["last_kl"]
        return (s1["sum_vloss"] - s0["sum_vloss"]) / n_up, (s1["sum_surr"] - s0["sum_surr"]) / n_up
