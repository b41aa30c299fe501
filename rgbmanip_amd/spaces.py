"""Minimal stand-ins for gym.spaces (gym is not installed in the build image).  If gym is importable its classes are used,
so `isinstance(x, Space)` checks in PPO (reference `algo/ppo/ppo/ppo.py:43-48`) behave as in the reference."""
import numpy as np

try:  # pragma: no cover - gym absent in this image
    from gym.spaces import Space, Box, Dict  # type: ignore
except Exception:  # noqa: BLE001
    class Space:  # noqa: D401
        shape = None

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
            self.dtype = dtype

    class Dict(Space):
        def __init__(self, spaces=None):
            self.spaces = dict(spaces or {})


def concat_spaces(space):
    """Dict space -> flat Box; Box passes through (`utils/tools.py:150-164`)."""
    if isinstance(space, Dict):
        n = sum(concat_spaces(v).shape[0] for v in space.spaces.values())
        return Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32)
    if isinstance(space, Box):
        return space
    raise NotImplementedError(f"Unsupported observation space: {type(space)}")
