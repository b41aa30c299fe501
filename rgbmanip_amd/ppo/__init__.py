"""PPO for the global-scheduling policy — same package surface as the reference's `algo/ppo/ppo/__init__.py:1-3`."""
from .storage import RolloutStorage
from .module import ActorCritic
from .ppo import PPO, prepare_obs

__all__ = ["RolloutStorage", "ActorCritic", "PPO", "prepare_obs"]
