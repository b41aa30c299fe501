"""`RLManipulation` (`/root/reference/models/manipulation/rl.py:12-26`): the manipulation policy as a PPO agent over a vec-env
(SURVEY.md §8f-4).  Same constructor and methods; the agent is the HIP-backed `rgbmanip_amd.ppo.PPO`."""
from __future__ import annotations

from .ppo import PPO


class BaseManipulation:                       # models/manipulation/base_manipulation.py:6-17
    def __init__(self, env, cfg: dict, logger=None):
        self.env = env
        self.cfg = cfg
        self.logger = logger

    def plan_pathway(self, obs, eval=False):
        pass


class RLManipulation(BaseManipulation):
    def __init__(self, vec_env, cfg: dict, logger=None, process_group=None):
        super().__init__(vec_env, cfg, logger)
        self.agent = PPO(vec_env, cfg, process_group=process_group)

    def learn(self, steps, log_interval=1, save_interval=1):
        self.agent.run(steps, log_interval, save_interval)

    def plan_pathway(self, obs, eval=False):
        self.agent.play()
