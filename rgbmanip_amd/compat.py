"""Import layer: the reference's module paths served by this package, so `train.py` runs unchanged.

`train.py` and its sibling modules import the hot-path classes by fixed module paths:

    from models.controller.rl_pose import RLPoseController                        # train.py:24
    from models.manipulation.rl import RLManipulation                            # train.py:32
    from models.pose_estimator.AdaPose.interface_v5 import AdaPoseEstimator_v5    # train.py:37, heuristic_pose.py:7-11
    from models.pose_estimator.base_estimator import BasePoseEstimator            # train.py:40, rl_pose.py:1
    from models.controller.base_controller import BaseController                  # train.py:20, rl_pose.py:3
    from algo.ppo.ppo import PPO / prepare_obs / RolloutStorage / ActorCritic     # rl_pose.py:10,494, rl.py:9, ppo.py:19-21

`install()` registers modules with exactly these names in `sys.modules`, each exposing this package's class under the
reference's name.  Python consults `sys.modules` before it searches a path, so with the reference tree on `sys.path` the
listed imports resolve here while every other reference module (simulator environments, scripted manipulation,
the other controllers) keeps loading from the reference.  Call it once before `train.py`'s own imports — first line of
`train.py`, a `sitecustomize.py`, or `python -c "import rgbmanip_amd.compat as c; c.install(); import runpy;
runpy.run_path('train.py', run_name='__main__')"`.  Without the reference tree the parent packages (`models`, `algo`,
...) are created as empty packages, so the same import lines work in isolation (tests/test_compat.py).
"""
from __future__ import annotations

import importlib
import sys
import types

# reference module path -> {name the reference imports: "our.module:attribute"}
ALIASES = {
    "models.pose_estimator.base_estimator": {"BasePoseEstimator": "rgbmanip_amd.estimator:BasePoseEstimator"},
    "models.pose_estimator.AdaPose.interface_v5": {"AdaPoseEstimator_v5": "rgbmanip_amd.estimator:AdaPoseEstimator_v5",
                                                   "BasePoseEstimator": "rgbmanip_amd.estimator:BasePoseEstimator"},
    "models.controller.base_controller": {"BaseController": "rgbmanip_amd.control_interface:BaseController"},
    "models.controller.rl_pose": {"RLPoseController": "rgbmanip_amd.control_interface:RLPoseController",
                                  "ControlInterface": "rgbmanip_amd.control_interface:ControlInterface",
                                  "CAMERA_INTRINSIC": "rgbmanip_amd.control_interface:CAMERA_INTRINSIC",
                                  "PPO": "rgbmanip_amd.ppo:PPO"},
    "models.manipulation.rl": {"RLManipulation": "rgbmanip_amd.manipulation:RLManipulation", "PPO": "rgbmanip_amd.ppo:PPO"},
    "algo.ppo.ppo": {"PPO": "rgbmanip_amd.ppo:PPO", "prepare_obs": "rgbmanip_amd.ppo:prepare_obs",
                     "ActorCritic": "rgbmanip_amd.ppo:ActorCritic", "RolloutStorage": "rgbmanip_amd.ppo:RolloutStorage"},
    "algo.ppo.ppo.ppo": {"PPO": "rgbmanip_amd.ppo.ppo:PPO", "prepare_obs": "rgbmanip_amd.ppo.ppo:prepare_obs"},
    "algo.ppo.ppo.module": {"ActorCritic": "rgbmanip_amd.ppo.module:ActorCritic"},
    "algo.ppo.ppo.storage": {"RolloutStorage": "rgbmanip_amd.ppo.storage:RolloutStorage"},
}
_PACKAGES = {"algo.ppo.ppo"}           # aliases that are packages in the reference (they have submodules)
_installed: dict[str, types.ModuleType | None] = {}      # name -> module that was there before install()


def _resolve(spec: str):
    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr)


def _ensure_parent(name: str) -> types.ModuleType:
    """The package `name`, imported from the reference tree if it is on sys.path, else an empty stand-in package."""
    if name in sys.modules:
        return sys.modules[name]
    if "." in name:
        _ensure_parent(name.rsplit(".", 1)[0])
    try:
        return importlib.import_module(name)
    except ImportError:
        pkg = types.ModuleType(name)
        pkg.__path__ = []            # a package (so that `import a.b.c` accepts it), with nothing to search
        pkg.__rgbm_compat__ = True
        _installed.setdefault(name, None)
        sys.modules[name] = pkg
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            setattr(sys.modules[parent], leaf, pkg)
        return pkg


def install(overwrite: bool = True) -> list[str]:
    """Register the alias modules; returns the names registered.  `overwrite=False` leaves already imported reference
    modules of the same name in place."""
    done = []
    for name in sorted(ALIASES, key=lambda n: n.count(".")):       # parents before children
        if name in sys.modules and not overwrite and not getattr(sys.modules[name], "__rgbm_compat__", False):
            continue
        parent_name, leaf = name.rsplit(".", 1)
        parent = _ensure_parent(parent_name)
        mod = types.ModuleType(name)
        mod.__doc__ = f"rgbmanip_amd.compat alias of the reference module `{name}`"
        mod.__rgbm_compat__ = True
        if name in _PACKAGES:
            mod.__path__ = []
        for attr, spec in ALIASES[name].items():
            setattr(mod, attr, _resolve(spec))
        mod.__all__ = list(ALIASES[name])
        _installed.setdefault(name, sys.modules.get(name))
        sys.modules[name] = mod
        setattr(parent, leaf, mod)
        done.append(name)
    return done


def uninstall() -> None:
    """Undo install(): put back whatever was registered before (tests)."""
    for name in sorted(_installed, key=lambda n: -n.count(".")):
        prev = _installed[name]
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            p = sys.modules.get(parent)
            if p is not None and getattr(p, leaf, None) is not prev:
                if prev is None:
                    if hasattr(p, leaf):
                        delattr(p, leaf)
                else:
                    setattr(p, leaf, prev)
    _installed.clear()
