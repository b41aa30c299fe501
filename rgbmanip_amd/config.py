"""Default configs with the reference's keys and values (`cfg/controller/rl.yaml`, `cfg/pose_estimator/adapose_*.yaml`).
Hydra is not required: the classes consume plain dicts exactly as `train.py:414-415` hands them over."""
import copy

RL_CONTROLLER_CFG = {
    "name": "rl",
    "controller": {"max_steps": 4, "action_type": "pose", "pose_min": [-0.3, -0.3, 0.4], "pose_max": [0.3, 0.3, 1.0], "early_stop": 4},
    "reward": {"diff_coef": -0.5, "move_success_coef": 8.0, "move_period_coef": -0.0, "far_coef": -2.5, "ori_coef": 0.25,
               "xyz_lookat_coef": -0.05, "bbox_coef": -1.0, "bbox_boundary_coef": -1.0, "have_bbox_coef": 2.0, "center_coef": 12.0,
               "open_coef": 8.0, "view_coef": 0.5, "view_norm_coef": -0.3, "success_coef": 0.0},
    "policy": {"actor_critic_class": "ActorCritic", "pi_hid_sizes": [96, 96, 32], "vf_hid_sizes": [96, 96, 32], "activation": "elu"},
    "learn": {
        "exp_name": "PPO", "reset": True, "num_transitions_per_env": 16, "num_transitions_eval": 512, "num_learning_epochs": 8,
        "num_mini_batches": 4, "clip_range": 0.2, "gamma": 0.98, "lam": 0.98, "init_noise_std": 0.6, "value_loss_coef": 1.0,
        "entropy_coef": 0.0, "learning_rate": 0.00001, "max_grad_norm": 1.0, "use_clipped_value_loss": True,
        "schedule": "adaptive", "desired_kl": 0.016, "max_lr": 0.005, "min_lr": 0.0002, "device": "cuda", "sampler": "sequential",
        "log_dir": "logs/ppo_controller", "save_dir": "saves/ppo_controller", "testing": False, "eval_interval": 64,
        "eval_round": 16, "eval": False, "print_log": True, "contrastive": False, "contrastive_m": 0.99, "asymmetric": False,
    },
    "load": "",
}


def adapose_cfg(task_name="one_door_cabinet", checkpoint_path="downloads/pose_estimator/one_door_cabinet.pth", load=True):
    return {"name": "adapose_v5", "task_name": task_name, "load": load, "checkpoint_path": checkpoint_path, "img_size": 224,
            "use_depth": True, "n_pts": 1024, "direct_regression": True, "real_world": False}


ADAPOSE_CFGS = {
    "adapose_cabinet": adapose_cfg("one_door_cabinet", "downloads/pose_estimator/one_door_cabinet.pth"),
    "adapose_drawer": adapose_cfg("one_drawer_cabinet", "downloads/pose_estimator/one_drawer_cabinet.pth"),
    "adapose_mug": adapose_cfg("mugs", "downloads/pose_estimator/mugs.pth"),
    "adapose_pot": adapose_cfg("pots", "downloads/pose_estimator/pots.pth"),
}


def rl_cfg(task="cabinet", **learn_overrides):
    """cfg/controller/rl.yaml merged with the task name ControlInterface reads (cfg/task/*.yaml `name`)."""
    cfg = copy.deepcopy(RL_CONTROLLER_CFG)
    cfg["learn"].update(learn_overrides)
    cfg["task"] = {"name": task}
    return cfg
