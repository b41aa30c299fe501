"""-m gpu: the controller step on the device — action decode, reward, grasp frame, the synthetic
camera and the full `ControlInterface.step` loop — against goldens recorded from the reference class, the oracle and the
numpy twin of the renderer."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rgbmanip_amd import _lib, synth  # noqa: E402


def _canon(q):
    from oracle.control_ref import canonical_quat
    return canonical_quat(q)


def test_lookat_quat_matches_reference_golden(golden_dir):
    """rgbm_lookat_quat on directions covering every per-row branch of utils/transform.py:50-99, against the reference's
    own output up to the eigenvector sign (the reference's sign is LAPACK's; the device's is canonical)."""
    g = np.load(os.path.join(golden_dir, "control_step.npz"))
    lib = _lib.load()
    d = torch.from_numpy(g["lookat_dir"]).cuda()
    q = torch.empty(d.shape[0], 4, dtype=torch.float64, device="cuda")
    _lib.check(lib.rgbm_lookat_quat(_lib.ptr(d), d.shape[0], 0, _lib.ptr(q), _lib.stream_ptr()), "lookat")
    q = q.cpu().numpy()
    np.testing.assert_allclose(q, _canon(g["lookat_quat"]), rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-14)
    # whole-batch-zero branch (transform.py:69): identity
    z = torch.zeros(2, 3, dtype=torch.float64, device="cuda")
    qz = torch.empty(2, 4, dtype=torch.float64, device="cuda")
    _lib.check(lib.rgbm_lookat_quat(_lib.ptr(z), 2, 1, _lib.ptr(qz), _lib.stream_ptr()), "lookat")
    np.testing.assert_array_equal(qz.cpu().numpy(), np.tile([1.0, 0, 0, 0], (2, 1)))


def test_control_step_matches_reference_golden(golden_dir):
    """The device ControlInterface driven through the episodes the reference class was recorded on
    (tests/golden/control_step.npz): observations / states / dones exact, the camera target equal up to quaternion sign,
    every reward term within 1e-12 (REW:diff, which depends on that sign through |cam_pose - target|, is checked against the
    value recomputed from the golden target with the canonical sign), grasp frames equal, env call flags equal."""
    from control_util import drive_steps
    from rgbmanip_amd.control_interface import ControlInterface, REWARD_KEYS
    g = np.load(os.path.join(golden_dir, "control_step.npz"))
    runs = drive_steps(ControlInterface, REWARD_KEYS)
    for task, rec in runs.items():
        np.testing.assert_array_equal(rec["obs"], g[task + "_obs"])
        np.testing.assert_array_equal(rec["state"], g[task + "_state"])
        np.testing.assert_array_equal(rec["done"], g[task + "_done"])
        tgt = g[task + "_target"].copy()
        tgt[..., 3:] = _canon(tgt[..., 3:])
        np.testing.assert_allclose(rec["target"], tgt, rtol=0, atol=1e-12)
        terms, gterms = rec["terms"], g[task + "_terms"]
        keep = [i for i, k in enumerate(REWARD_KEYS) if k != "REW:diff"]
        np.testing.assert_allclose(terms[:, keep], gterms[:, keep], rtol=0, atol=1e-12)
        # REW:diff with the canonical target sign: cam_pose is the replay env's pose of that step (t = step + 1 (+1 per reset))
        env, t = synth.ReplayVecEnv(3, 4), 0
        exp_diff = []
        env.get_image()                                            # reset_robot
        for step in range(terms.shape[0]):
            if step > 0 and g[task + "_done"][step - 1].any():
                env.get_image()                                    # reset -> reset_robot
            env.get_image()
            exp_diff.append(np.clip(np.linalg.norm(env.camera_pose() - tgt[step], axis=-1), -2, 2) * synth.CONTROL_REWARD_CFG["diff_coef"])
        np.testing.assert_allclose(terms[:, 0], np.stack(exp_diff), rtol=0, atol=1e-12)
        np.testing.assert_allclose(rec["reward"] - terms[:, 0], g[task + "_reward"] - gterms[:, 0], rtol=0, atol=1e-11)
        env, man = rec["env"], rec["manipulation"]
        assert env.resets == int(g[task + "_resets"])
        flags = np.array([[m["skip_move"], m["no_collision_with_front"], m["robot_frame"]] for m in env.moves])
        np.testing.assert_array_equal(flags, g[task + "_move_flags"])
        np.testing.assert_array_equal(np.array([[m["time"], m["wait"]] for m in env.moves]), g[task + "_move_tw"])
        moves = np.stack([np.broadcast_to(m["pose"], (3, 7)) for m in env.moves])
        gm = g[task + "_move_pose"].copy()
        gm[..., 3:] = _canon(gm[..., 3:])
        np.testing.assert_allclose(moves, gm, rtol=0, atol=1e-12)
        if task == "pots":
            np.testing.assert_allclose(np.stack([c[0] for c in man.calls]), g["pots_manip_center"], rtol=0, atol=1e-14)
            np.testing.assert_allclose(np.stack([c[1] for c in man.calls]), g["pots_manip_direction"], rtol=0, atol=1e-14)
            assert [c[2] for c in man.calls] == list(g["pots_manip_eval"])


def test_control_save_data_matches_reference_golden(golden_dir, tmp_path, monkeypatch):
    """The device ControlInterface's eval-time dataset export against the files the reference class wrote
    (tests/golden/control_save.npz): same paths, shapes, float64 dtype, sums and strided samples."""
    from control_util import check_saved_dataset
    from rgbmanip_amd.control_interface import ControlInterface
    check_saved_dataset(ControlInterface, golden_dir, tmp_path, monkeypatch)


def test_control_reward_float32_terms_bit_exact(golden_dir):
    """The float32 terms of the reward (view-norm penalty, xyz-lookat, move success, bbox boundary) follow numpy's float32
    arithmetic exactly; the float64 terms that involve no transcendental function are bit-exact too."""
    from control_util import drive_steps
    from rgbmanip_amd.control_interface import ControlInterface, REWARD_KEYS
    g = np.load(os.path.join(golden_dir, "control_step.npz"))
    rec = drive_steps(ControlInterface, REWARD_KEYS)["cabinet"]
    for name in ("REW:view_norm_penalty", "REW:xyz_lookat", "REW:move_success", "REW:bbox_boundary_penalty", "REW:have_bbox",
                 "REW:far", "REW:bbox_penalty", "REW:move_period", "REW:view_rew", "LOSS:center_diff"):
        i = REWARD_KEYS.index(name)
        np.testing.assert_array_equal(rec["terms"][:, i], g["cabinet_terms"][:, i], err_msg=name)


def _synth_inputs(N, episode=0):
    from oracle import control_ref as cr
    from rgbmanip_amd import synthetic_env as se
    robots, boxes = (np.stack(a) for a in zip(*[se.sample_scene(i, episode) for i in range(N)]))
    rng = np.random.default_rng(5)
    cam = np.zeros((N, 7))
    cam[:, :3] = rng.uniform([-0.3, -0.3, 0.4], [0.3, 0.3, 1.0], (N, 3))
    heading = np.concatenate([np.ones((N, 1)), rng.normal(0, 0.15, (N, 2))], axis=1)
    heading[:, 2] -= 0.1
    cam[:, 3:] = cr.canonical_quat(cr.lookat_quat(heading)) * 1.7          # un-normalised on purpose: the kernel normalises
    return robots, boxes, cam


def test_synth_camera_bit_exact_against_numpy_twin():
    """rgbm_synth_camera + rgbm_synth_render vs oracle/synth_env_ref.py: K, E, colour frame and mask identical bit for bit."""
    from oracle import synth_env_ref as sr
    from rgbmanip_amd import synthetic_env as se
    N = 4
    robots, boxes, cam = _synth_inputs(N)
    env = se.SyntheticMultiVecEnv(N, "cuda", seed=0)
    env._robot.copy_(torch.from_numpy(robots)); env._box.copy_(torch.from_numpy(boxes)); env._cam.copy_(torch.from_numpy(cam))
    img = env.get_image()["camera0"]
    f = se.CAM_F
    K, E, rays = sr.camera_ref(cam, robots, boxes, f, f, 320.0, 240.0)
    color, mask = sr.render_ref(rays, boxes, f, f, 320.0, 240.0, 480, 640, env0=0)
    np.testing.assert_array_equal(img["Intrinsic"].cpu().numpy(), K)
    np.testing.assert_array_equal(img["Extrinsic"].cpu().numpy(), E)
    np.testing.assert_array_equal(img["Mask"].cpu().numpy(), mask)
    np.testing.assert_array_equal(img["Color"].cpu().numpy(), color)
    assert mask.reshape(N, -1).sum(1).min() > 0                     # every handle is in view in this set-up
    np.testing.assert_array_equal(env.get_observation(gt=True)["handle_bbox"].cpu().numpy(), sr.box_corners(boxes))


class _NumpyEnvView:
    """A SyntheticMultiVecEnv seen through numpy arrays, i.e. what the reference's host code would receive."""

    def __init__(self, env):
        self.env, self.num_envs = env, env.num_envs

    def cam_move_to(self, pose, **kw):
        ok, period = self.env.cam_move_to(np.asarray(pose), **kw)
        return [ok.cpu().numpy(), period.cpu().numpy()]

    def get_image(self):
        return {"camera0": {k: v.cpu().numpy() for k, v in self.env.get_image()["camera0"].items()}}

    def camera_pose(self, robot_frame=False):
        return self.env.camera_pose(robot_frame).cpu().numpy()

    def robot_pose(self):
        return self.env.robot_pose().cpu().numpy()

    def get_observation(self, gt=False):
        return {k: v.cpu().numpy() for k, v in self.env.get_observation(gt).items()}

    def reset(self, indices=None):
        return self.env.reset(indices)


class _ReplayEstimator:
    def __init__(self, boxes):
        self.cfg, self.boxes, self.i = {"task_name": "cabinet"}, boxes, 0

    def estimate(self, *a):
        self.i += 1
        return self.boxes[self.i - 1]


def test_control_step_on_synthetic_env_matches_oracle(monkeypatch):
    """Full loop on the GPU — SyntheticMultiVecEnv render -> device queues -> rgbm_prepare_inputs -> AdaPose (HIP) ->
    post-processing -> rgbm_control_reward — against the oracle's ControlInterfaceRef fed the numpy view of an identical
    env and the boxes the device estimator produced: rewards / observations / dones agree, masks are non-trivial."""
    from oracle import control_ref as cr
    from oracle.control_ref import ControlInterfaceRef, REWARD_KEYS as RK
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface, REWARD_KEYS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    assert RK == REWARD_KEYS
    N, steps = 4, 6
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=1)
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="fp32")
    boxes = []

    class Rec:
        cfg = est.cfg

        def estimate_device(self, *a):
            b = est.estimate_device(*a)
            boxes.append(b.cpu().numpy())
            return b
    ccfg = synth.control_cfg("cabinet", 0.0)
    env = se.SyntheticMultiVecEnv(N, "cuda", seed=3)
    ci = ControlInterface(env, Rec(), se.SyntheticManipulation(env), ccfg)
    acts = [synth.control_actions(N, s, 9) * 0.3 for s in range(steps)]
    dev = [ci.step(torch.from_numpy(a).cuda()) for a in acts]
    assert len(boxes) == steps
    avail_px = ci.mask_queue.view(ci.max_steps, N, -1).sum(-1)
    assert int((avail_px > 0).sum()) >= N                          # handles are seen
    env2 = se.SyntheticMultiVecEnv(N, "cuda", seed=3)
    # the oracle inherits LAPACK's eigenvector sign for the target quaternion, which the env hands back as the camera pose
    # (hence as observation); give it the device's canonical sign so the two runs see the same numbers
    raw_lookat = cr.lookat_quat
    monkeypatch.setattr(cr, "lookat_quat", lambda d: cr.canonical_quat(raw_lookat(d)))
    ref = ControlInterfaceRef(_NumpyEnvView(env2), _ReplayEstimator(boxes), None, ccfg)
    for s, a in enumerate(acts):
        obs, rew, done, info = ref.step(a)
        dobs, drew, ddone, dinfo = dev[s]
        np.testing.assert_array_equal(dobs.cpu().numpy(), obs)
        np.testing.assert_array_equal(ddone.cpu().numpy(), done)
        for k in REWARD_KEYS:
            np.testing.assert_allclose(dinfo[k].cpu().numpy(), np.asarray(info[k], dtype=np.float64), rtol=0, atol=1e-9, err_msg=k)
        assert np.isfinite(drew.cpu().numpy()).all()


def test_ppo_trains_on_device_control_interface():
    """cfg/controller/rl.yaml end to end on one GPU: PPO.run over ControlInterface(SyntheticMultiVecEnv, HIP estimator)."""
    from test_gpu_ppo import CFG
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    from rgbmanip_amd.ppo import PPO
    N = 8
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=1)
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="bf16")
    env = se.SyntheticMultiVecEnv(N, "cuda", seed=1)
    ci = ControlInterface(env, est, se.SyntheticManipulation(env), synth.control_cfg("cabinet"))
    ppo = PPO(ci, CFG)
    before = ppo.actor_critic.flat.clone()
    ppo.run(1, log_interval=1, save_interval=10 ** 9)
    assert torch.isfinite(ppo.actor_critic.flat).all()
    assert not torch.equal(before, ppo.actor_critic.flat)
    assert env.episode.min() >= 3                                   # 16 transitions = 4 episodes of 4 steps


def test_rl_manipulation_wrapper_learns_and_plays():
    """models/manipulation/rl.py: RLManipulation(vec_env, cfg, logger).learn / plan_pathway over the device controller env."""
    import copy
    from test_gpu_ppo import CFG
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    from rgbmanip_amd.manipulation import RLManipulation
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device")
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="bf16")
    env = se.SyntheticMultiVecEnv(4, "cuda", seed=2)
    ci = ControlInterface(env, est, se.SyntheticManipulation(env), synth.control_cfg("cabinet"))
    pcfg = copy.deepcopy(CFG)
    pcfg["learn"].update(num_transitions_per_env=4, num_transitions_eval=3, num_learning_epochs=1)
    man = RLManipulation(ci, pcfg, None)
    man.learn(1, log_interval=1, save_interval=10 ** 9)
    steps_before = env.episode.copy()
    man.plan_pathway(None, eval=True)
    assert torch.isfinite(man.agent.actor_critic.flat).all()
    assert (env.episode > steps_before).all()                      # play() resets and steps the env


def test_rl_pose_controller_train_and_run(tmp_path, monkeypatch):
    """rl_pose.py:464-516: RLPoseController.train_controller (PPO over the device ControlInterface) and run (deterministic
    roll-out to the end of the episode, last estimate handed to the manipulation planner; its eval steps export the
    third-stage dataset, rl_pose.py:446-447, here into a scratch directory)."""
    import copy
    monkeypatch.chdir(tmp_path)
    from test_gpu_ppo import CFG
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import RLPoseController
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    ecfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device")
    est = AdaPoseEstimator_v5(None, ecfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="bf16")
    env = se.SyntheticMultiVecEnv(4, "cuda", seed=5)
    calls = []

    class Plan:
        def plan_pathway(self, center, direction, eval):
            calls.append((center.cpu().numpy(), direction.cpu().numpy(), eval))
    cfg = copy.deepcopy(CFG)
    cfg["learn"].update(num_transitions_per_env=4, num_learning_epochs=1)
    cfg.update(synth.control_cfg("cabinet"))
    ctl = RLPoseController(env, est, Plan(), cfg, None)
    ctl.train_controller(1, log_interval=1, save_interval=10 ** 9)
    est_box = ctl.run(eval=True)
    assert est_box.shape == (4, 8, 3) and len(calls) == 1 and calls[0][2] is True
    assert np.isfinite(calls[0][0]).all() and np.isfinite(calls[0][1]).all()
    assert ctl.control_interface.accumulate_steps == ctl.control_interface.max_steps          # ran to the end of the episode
    saved = sorted(os.listdir(os.path.join("saves", "third_stage")))
    assert saved == sorted(c["name"] for c in env.get_attr("current_obj_config"))              # one directory per object, 8 files each
    assert len(os.listdir(os.path.join("saves", "third_stage", saved[0], "1"))) == 8


def test_fp16_estimator_in_the_control_loop():
    """configs[4]'s storage type through the whole device loop: ControlInterface.step over the synthetic env with an fp16
    estimator stays finite and agrees with the fp32 estimator's boxes to fp16 accuracy on the same frames."""
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=1)
    sd = synth.adapose_state_dict(seed=0, prefix="module.")
    boxes = {}
    for dt in ("fp32", "fp16"):
        est = AdaPoseEstimator_v5(None, cfg, None, state_dict=sd, dtype=dt)
        env = se.SyntheticMultiVecEnv(4, "cuda", seed=3)
        ci = ControlInterface(env, est, se.SyntheticManipulation(env), synth.control_cfg("cabinet"))
        out = []
        for s in range(3):
            obs, rew, done, info = ci.step(torch.from_numpy(synth.control_actions(4, s, 9) * 0.3).cuda())
            assert torch.isfinite(rew).all()
            out.append(ci.pred_bbox[ci.accumulate_steps - 1].cpu().numpy())
        boxes[dt] = np.stack(out)
    scale = np.abs(boxes["fp32"]).max()
    assert np.abs(boxes["fp16"] - boxes["fp32"]).max() / scale < 2e-2


class _SubsetEnvView:
    """A strided subset of a full-size SyntheticMultiVecEnv as numpy arrays (what the reference's host code would see for
    those envs).  The full env has to move ALL its cameras to stay identical to the device run's env, so every
    `cam_move_to` replays the full-size pose the device run issued at that call, with the subset's rows replaced by the
    poses the oracle computed for them."""

    def __init__(self, env, sub, recorded_poses):
        self.env, self.sub, self.rec, self.calls = env, torch.as_tensor(sub, device="cuda"), recorded_poses, 0
        self.num_envs = len(sub)

    def cam_move_to(self, pose, **kw):
        full = self.rec[self.calls].clone()
        self.calls += 1
        p = torch.as_tensor(np.asarray(pose), dtype=torch.float64, device="cuda")
        if full.dim() == 2 and p.dim() == 2:
            full[self.sub] = p
        else:
            assert p.dim() == 1 and torch.allclose(full.reshape(-1)[:7], p, rtol=0, atol=1e-12)      # a broadcast pose (reset_robot)
        ok, period = self.env.cam_move_to(full, **kw)
        return [ok[self.sub].cpu().numpy(), period[self.sub].cpu().numpy()]

    def get_image(self):
        return {"camera0": {k: v[self.sub].cpu().numpy() for k, v in self.env.get_image()["camera0"].items()}}

    def camera_pose(self, robot_frame=False):
        return self.env.camera_pose(robot_frame)[self.sub].cpu().numpy()

    def robot_pose(self):
        return self.env.robot_pose()[self.sub].cpu().numpy()

    def get_observation(self, gt=False):
        return {k: v[self.sub].cpu().numpy() for k, v in self.env.get_observation(gt).items()}


def test_control_step_and_ppo_iteration_at_512_envs(monkeypatch):
    """BASELINE configs[2] at its own size: `ControlInterface.step` over 512 synthetic envs with the HIP estimator in the loop
    (bf16x3: the mode inside the 1e-4 gate), checked on a strided subset of envs against the oracle — `ControlInterfaceRef`
    for observations / dones / the 17 reward terms, and the CPU estimator pipeline (oracle prepare_model_input with the same
    subset hash -> oracle network -> oracle post-processing) for the boxes — and then one full `PPO.run` iteration
    (16 transitions x 512 envs, 32 optimiser steps) with the bf16 estimator the config names."""
    from oracle import adapose_ref, control_ref as cr, postproc_ref
    from oracle.control_ref import ControlInterfaceRef
    from test_gpu_ppo import CFG
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface, REWARD_KEYS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5, DEFAULT_BBOX
    from rgbmanip_amd.ppo import PPO
    N, steps, sub, seed = 512, 2, [0, 73, 255, 511], 1
    ecfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=seed)
    sd = synth.adapose_state_dict(seed=0, prefix="module.")
    est = AdaPoseEstimator_v5(None, ecfg, None, state_dict=sd, dtype="bf16x3")
    boxes, poses = [], []

    class Rec:
        cfg = est.cfg

        def estimate_device_indexed(self, *a):
            b = est.estimate_device_indexed(*a)
            boxes.append(b.cpu().numpy())
            return b
    ccfg = synth.control_cfg("cabinet", 0.0)
    env = se.SyntheticMultiVecEnv(N, "cuda", seed=3)
    move = env.cam_move_to

    def rec_move(pose, **kw):
        poses.append(torch.as_tensor(pose).to(device="cuda", dtype=torch.float64).clone())
        return move(pose, **kw)
    env.cam_move_to = rec_move
    ci = ControlInterface(env, Rec(), se.SyntheticManipulation(env), ccfg)
    acts = [synth.control_actions(N, s, 9) * 0.3 for s in range(steps)]
    dev = [ci.step(torch.from_numpy(a).cuda()) for a in acts]
    assert len(boxes) == steps and boxes[0].shape == (N, 8, 3)
    assert float((ci.mask_queue.view(ci.max_steps, N, -1).sum(-1) > 0).float().mean()) > 0.3      # handles are seen

    tsd = adapose_ref.to_torch_sd(synth.adapose_state_dict(seed=0))
    own = []

    class OracleEstimator:
        """CPU pipeline on the subset's frames; hands the DEVICE boxes on (so the reward terms see identical numbers) and
        keeps its own for the comparison below."""
        cfg = {"task_name": "cabinet"}

        def __init__(self):
            self.i = 0

        def estimate(self, K, rgb1, m1, E1, rgb2, m2, E2):
            out = np.repeat(DEFAULT_BBOX[None], len(sub), axis=0)
            rows, ins = [], []
            for j, e in enumerate(sub):
                a = postproc_ref.prepare_model_input(rgb1[j], m1[j], K[j], 224, rng=("hash", seed, e))
                b = postproc_ref.prepare_model_input(rgb2[j], m2[j], K[j], 224, rng=("hash", seed + 1, e))
                if a[0] is None or b[0] is None:
                    continue
                P1, P2 = np.eye(4), np.eye(4)
                P1[:3] = a[3] @ E1[j][:3]
                P2[:3] = b[3] @ E2[j][:3]
                rows.append(j)
                ins.append((a, b, P1.astype(np.float32), P2.astype(np.float32)))
            if rows:
                t = lambda xs, dt: torch.from_numpy(np.stack(xs)).to(dt)  # noqa: E731
                dep = (torch.arange(24, dtype=torch.float32) * 0.1 + 0.1)[None].repeat(len(rows), 1)
                o = adapose_ref.adapose_forward(tsd, t([i[0][0] for i in ins], torch.float32), t([i[0][1] for i in ins], torch.int64),
                                                t([i[1][0] for i in ins], torch.float32), t([i[1][1] for i in ins], torch.int64),
                                                t([i[2] for i in ins], torch.float32), t([i[3] for i in ins], torch.float32), dep)
                for q, j in enumerate(rows):
                    out[j] = postproc_ref.bbox_world(o["view1_nocs"][q].numpy(), o["view1_depth"][q].numpy(), o["view1_r"][q].numpy(),
                                                     ins[q][0][1], ins[q][0][3], E1[j])
            own.append(out)
            self.i += 1
            return boxes[self.i - 1][sub]

    env2 = se.SyntheticMultiVecEnv(N, "cuda", seed=3)
    raw_lookat = cr.lookat_quat
    monkeypatch.setattr(cr, "lookat_quat", lambda d: cr.canonical_quat(raw_lookat(d)))
    # the reference's quat_to_axis scrambles its batch (transform.py:234: row i holds elements 3i..3i+2 of [A.., B.., C..] over
    # ALL envs), so REW:ori_rew of an env depends on the other envs' cameras: the subset oracle gets the rows the FULL batch gives
    raw_q2a = cr.quat_to_axis0
    monkeypatch.setattr(cr, "quat_to_axis0", lambda q: raw_q2a(env2.camera_pose(robot_frame=True).cpu().numpy()[:, 3:])[sub])
    ref = ControlInterfaceRef(_SubsetEnvView(env2, sub, poses), OracleEstimator(), None, ccfg)
    for s, a in enumerate(acts):
        obs, rew, done, info = ref.step(a[sub])
        dobs, drew, ddone, dinfo = dev[s]
        np.testing.assert_array_equal(dobs[sub].cpu().numpy(), obs)
        np.testing.assert_array_equal(ddone[sub].cpu().numpy(), done)
        for k in REWARD_KEYS:
            np.testing.assert_allclose(dinfo[k][sub].cpu().numpy(), np.asarray(info[k], dtype=np.float64), rtol=0, atol=1e-9, err_msg=k)
        # the estimator in the loop against the CPU pipeline on the same frames (fp32 oracle vs bf16x3 device)
        got, exp = boxes[s][sub], own[s]
        real = np.abs(exp - DEFAULT_BBOX).max(axis=(1, 2)) > 1e-9
        assert np.array_equal(real, np.abs(got - DEFAULT_BBOX).max(axis=(1, 2)) > 1e-9)          # same samples fell back to the +10 cube
        assert real.any()
        err = np.abs(got[real] - exp[real]).max() / np.abs(exp[real]).max()
        print(f"step {s}: device boxes vs CPU oracle pipeline on envs {sub}: rel err {err:.2e}")
        assert err < 1e-3, (s, err)

    # one learning iteration of the trainer at the config's size (bf16 estimator, as configs[2] names it)
    est16 = AdaPoseEstimator_v5(None, ecfg, None, state_dict=sd, dtype="bf16")
    env3 = se.SyntheticMultiVecEnv(N, "cuda", seed=1)
    ppo = PPO(ControlInterface(env3, est16, se.SyntheticManipulation(env3), ccfg), CFG)
    before = ppo.actor_critic.flat.clone()
    ppo.run(1, log_interval=1, save_interval=10 ** 9)
    assert torch.isfinite(ppo.actor_critic.flat).all() and not torch.equal(before, ppo.actor_critic.flat)
    assert ppo.storage.observations.shape == (16, N, 60) and ppo.last_fps > 0
    assert env3.episode.min() >= 3                                   # 16 transitions = 4 episodes of 4 steps, on every env
