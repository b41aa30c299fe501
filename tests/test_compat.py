"""CPU: the import layer (`rgbmanip_amd.compat.install`) serves the reference's module paths, so the import lines of
`train.py` / `rl_pose.py` / `rl.py` / `ppo.py` resolve to this package and its classes construct from the reference's
cfg dicts (SURVEY.md §8b "drops into train.py unchanged")."""
import sys

import numpy as np
import pytest
import torch

from rgbmanip_amd import compat, config

# the reference's own import statements for the hot-path classes (file:line in the comment)
REFERENCE_IMPORT_LINES = [
    "from models.controller.base_controller import BaseController",                     # train.py:20, rl_pose.py:3
    "from models.controller.rl_pose import RLPoseController",                           # train.py:24
    "from models.manipulation.rl import RLManipulation",                                # train.py:32
    "from models.pose_estimator.AdaPose.interface_v5 import AdaPoseEstimator_v5",       # train.py:37, heuristic_pose.py:7-11
    "from models.pose_estimator.base_estimator import BasePoseEstimator",               # train.py:40, rl_pose.py:1
    "from algo.ppo.ppo import PPO",                                                     # rl_pose.py:10, models/manipulation/rl.py:9
    "from algo.ppo.ppo import prepare_obs",                                             # rl_pose.py:494
    "from algo.ppo.ppo import RolloutStorage",                                          # algo/ppo/ppo/ppo.py:19
    "from algo.ppo.ppo import ActorCritic",                                             # algo/ppo/ppo/ppo.py:20
    "from .storage import RolloutStorage",                                              # algo/ppo/ppo/__init__.py:1 (relative: checked below)
]


@pytest.fixture()
def installed():
    names = compat.install()
    yield names
    compat.uninstall()


def test_reference_import_lines_resolve_to_this_package(installed):
    ns = {}
    for line in REFERENCE_IMPORT_LINES:
        if line.startswith("from ."):
            continue
        exec(line, ns)
    import rgbmanip_amd.control_interface as ci
    import rgbmanip_amd.estimator as est
    import rgbmanip_amd.manipulation as man
    import rgbmanip_amd.ppo as ppo
    assert ns["RLPoseController"] is ci.RLPoseController and issubclass(ns["RLPoseController"], ns["BaseController"])
    assert ns["AdaPoseEstimator_v5"] is est.AdaPoseEstimator_v5 and issubclass(ns["AdaPoseEstimator_v5"], ns["BasePoseEstimator"])
    assert ns["RLManipulation"] is man.RLManipulation
    assert ns["PPO"] is ppo.PPO and ns["prepare_obs"] is ppo.prepare_obs
    assert ns["RolloutStorage"] is ppo.RolloutStorage and ns["ActorCritic"] is ppo.ActorCritic
    # the package form the reference itself uses (algo/ppo/ppo/__init__.py:1-3: .storage / .module / .ppo submodules)
    import algo.ppo.ppo.module as m
    import algo.ppo.ppo.ppo as p
    import algo.ppo.ppo.storage as s
    assert s.RolloutStorage is ppo.RolloutStorage and m.ActorCritic is ppo.ActorCritic and p.PPO is ppo.PPO
    from models.controller.rl_pose import CAMERA_INTRINSIC, ControlInterface      # rl_pose.py:4 constant, :14 class
    assert CAMERA_INTRINSIC == [0.05, 100, 1, 640, 480] and ControlInterface is ci.ControlInterface
    assert "models.controller.rl_pose" in installed and "algo.ppo.ppo" in installed


def test_uninstall_restores_sys_modules():
    before = set(sys.modules)
    compat.install()
    assert "models.controller.rl_pose" in sys.modules
    compat.uninstall()
    assert not ({"models", "algo", "models.controller.rl_pose", "algo.ppo.ppo"} & (set(sys.modules) - before))


class _Env:
    """The attributes PPO / ControlInterface read from the vec-env at construction time (env/my_vec_env.py)."""
    def __init__(self, n):
        from rgbmanip_amd.spaces import Box
        self.num_envs = n
        self.observation_space = Box(-1.5, 1.5, (60,))
        self.state_space = Box(-1.5, 1.5, (75,))
        self.action_space = Box(-1.5, 1.5, (12,))


def test_classes_construct_from_reference_cfg_dicts(installed, tmp_path):
    """train.py:238-240 builds the estimator from cfg["pose_estimator"] and rl_pose.py:478 builds PPO from cfg["controller"];
    here the same constructor calls on the CPU, as far as they go without a device (the network itself needs the GPU)."""
    from algo.ppo.ppo import PPO, ActorCritic, RolloutStorage
    from models.pose_estimator.AdaPose.interface_v5 import AdaPoseEstimator_v5
    from models.pose_estimator.base_estimator import BasePoseEstimator
    cfg = config.rl_cfg(task="cabinet", device="cpu", log_dir=str(tmp_path / "log"), save_dir=str(tmp_path / "save"), print_log=False)
    ppo = PPO(_Env(8), cfg)                                      # cfg/controller/rl.yaml keys, verbatim
    assert isinstance(ppo.actor_critic, ActorCritic) and isinstance(ppo.storage, RolloutStorage)
    assert ppo.actor_critic.total == 36985 and ppo.storage.observations.shape == (16, 8, 60)
    assert ppo.world == 1 and ppo.rank == 0
    ppo.save(str(tmp_path / "save" / "model_3.pt"))
    ppo.load(str(tmp_path / "save" / "model_3.pt"))
    assert ppo.current_learning_iteration == 3
    with pytest.raises(TypeError):                               # non-gym spaces are rejected like ppo.py:43-48
        bad = _Env(8)
        bad.action_space = (12,)
        PPO(bad, cfg)
    # the estimator: cfg/pose_estimator/adapose_cabinet.yaml keys; construction needs the device, so it must say so loudly
    ecfg = dict(config.ADAPOSE_CFGS["adapose_cabinet"], load=False)
    assert set(ecfg) >= {"name", "task_name", "load", "checkpoint_path", "img_size", "use_depth", "n_pts", "direct_regression", "real_world"}
    if not torch.cuda.is_available():
        with pytest.raises(Exception) as ei:
            AdaPoseEstimator_v5(None, ecfg, None)
        assert "GPU" in str(ei.value) or "cuda" in str(ei.value).lower() or "HIP" in str(ei.value)
    est = AdaPoseEstimator_v5.__new__(AdaPoseEstimator_v5)
    BasePoseEstimator.__init__(est, None, ecfg, None)
    assert est.cfg["task_name"] == "one_door_cabinet" and est.env is None


def test_episode_scalars_follow_reference_keys():
    """PPO.log's `Episode/<key>_train` scalars (ppo.py:364-384): mean over steps and envs of every info entry."""
    from rgbmanip_amd.ppo import PPO
    ppo = PPO.__new__(PPO)
    ppo.device = "cpu"
    infos = [{"REW:move": torch.tensor([1.0, 3.0]), "LOSS:far": torch.tensor([0.5, 0.5])},
             {"REW:move": torch.tensor([5.0, 7.0]), "LOSS:far": torch.tensor([1.5, 2.5])}]
    sc = ppo.episode_scalars(infos)
    assert sc == {"Episode/REW:move_train": 4.0, "Episode/LOSS:far_train": 1.25}
    sc = ppo.episode_scalars([{"success_rate": torch.tensor([1.0])}])
    assert set(sc) == {"Episode/worst_50.0%_success_rate_train", "Episode/success_rate_train"} and all(np.isnan(v) for v in sc.values())
    assert ppo.episode_scalars([]) == {}
