"""Seeded synthetic weights and inputs for the AdaPose / PPO hot path.

There are no checkpoints in the container (reference `install.sh:1-16` downloads them), so
parity fixtures, GPU tests and bench.py all use weights generated here from numpy PCG64
streams keyed by the state_dict entry name.  The schema (names, shapes) is the one of the
reference module `StereoPoseNet_with_depth(n_cat=1, nv_pts=1024, regress_pose=True)`
(`models/pose_estimator/AdaPose/lib/network_v5.py:301-376`, `lib/pspnet.py:33-126`) and
of `ActorCritic` (`algo/ppo/ppo/module.py:8-67`); SURVEY.md Appendix A lists it.

The distributions mimic the reference's own initialisers so activations stay well scaled:
backbone Conv2d ~ N(0, sqrt(2/(k*k*out))) (`pspnet.py:45-48`), everything else PyTorch's
default U(+-1/sqrt(fan_in)); BatchNorm3d gets non-trivial affine/running stats so that
BN folding is really exercised.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

IMG = 224
N_PTS = 1024
N_DEPTH = 24
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)


# --------------------------------------------------------------------------------------
# AdaPose schema
# --------------------------------------------------------------------------------------
def adapose_schema():
    """List of (name, shape, kind) in the reference's state_dict order."""
    s = []
    fe = "img_extractor.feats."
    s.append((fe + "conv1.weight", (64, 3, 7, 7), "bb"))
    inpl = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for b in range(blocks):
            cin = inpl if b == 0 else planes
            s.append((f"{fe}layer{li}.{b}.conv1.weight", (planes, cin, 3, 3), "bb"))
            s.append((f"{fe}layer{li}.{b}.conv2.weight", (planes, planes, 3, 3), "bb"))
            if b == 0 and cin != planes:
                s.append((f"{fe}layer{li}.{b}.downsample.0.weight", (planes, cin, 1, 1), "bb"))
        inpl = planes
    for i in range(4):
        s.append((f"img_extractor.psp.stages.{i}.1.weight", (128, 512, 1, 1), "w"))
    for nm, cin, cout in (("up_1", 1024, 256), ("up_2", 256, 64), ("up_3", 64, 64)):
        s.append((f"img_extractor.{nm}.conv.0.weight", (cout, cin, 3, 3), "w"))
        s.append((f"img_extractor.{nm}.conv.0.bias", (cout,), "b:%d" % (cin * 9)))
        s.append((f"img_extractor.{nm}.conv.1.weight", (1,), "prelu"))
    s.append(("img_extractor.final.weight", (32, 64, 1, 1), "w"))
    s.append(("img_extractor.final.bias", (32,), "b:64"))
    s.append(("instance_color.0.weight", (64, 32, 1), "w"))
    s.append(("instance_color.0.bias", (64,), "b:32"))
    cr = "cost_regularization."

    def bn(prefix, c):
        return [(prefix + ".bn.weight", (c,), "bn_g"), (prefix + ".bn.bias", (c,), "bn_b"),
                (prefix + ".bn.running_mean", (c,), "bn_m"), (prefix + ".bn.running_var", (c,), "bn_v"),
                (prefix + ".bn.num_batches_tracked", (), "nbt")]

    for nm, cin, cout in (("conv0", 32, 8), ("conv1", 8, 16), ("conv2", 16, 16), ("conv3", 16, 32),
                          ("conv4", 32, 32), ("conv5", 32, 64), ("conv6", 64, 64)):
        s.append((f"{cr}{nm}.conv.weight", (cout, cin, 3, 3, 3), "w"))
        s += bn(cr + nm, cout)
    for nm, cin, cout in (("conv7", 64, 32), ("conv9", 32, 16), ("conv11", 16, 8)):
        # ConvTranspose3d weight is [in, out, k, k, k]; PyTorch's fan_in uses dim 1
        s.append((f"{cr}{nm}.conv.weight", (cin, cout, 3, 3, 3), "wT"))
        s += bn(cr + nm, cout)
    s.append((cr + "prob.weight", (1, 8, 3, 3, 3), "w"))

    def mlp(prefix, dims, idx, conv1d=True):
        out = []
        for i, (a, b) in zip(idx, zip(dims[:-1], dims[1:])):
            out.append((f"{prefix}.{i}.weight", (b, a, 1) if conv1d else (b, a), "w"))
            out.append((f"{prefix}.{i}.bias", (b,), "b:%d" % a))
        return out

    s += mlp("nocs_head", (64, 128, 64, 3), (0, 2, 4))
    s += mlp("nocs_pts_mlp", (3, 32, 64), (0, 2))
    s += mlp("pose_mlp1", (96, 128, 128), (0, 2))
    s += mlp("pose_mlp2", (256, 256, 256), (0, 2))
    s += mlp("rotation_estimator", (256, 256, 128, 6), (0, 2, 4), conv1d=False)
    s += mlp("translation_estimator", (256, 256, 128, 3), (0, 2, 4), conv1d=False)
    s += mlp("size_estimator", (256, 256, 128, 3), (0, 2, 4), conv1d=False)
    return s


def adapose_state_dict(seed: int = 0, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Synthetic checkpoint: name -> fp32 ndarray (int64 for num_batches_tracked)."""
    sd = OrderedDict()
    for name, shape, kind in adapose_schema():
        g = _rng(name, seed)
        if kind == "bb":
            n = shape[2] * shape[3] * shape[0]
            v = g.normal(0.0, np.sqrt(2.0 / n), size=shape)
        elif kind == "w":
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = g.uniform(-bound, bound, size=shape)
        elif kind == "wT":
            fan_in = int(shape[1] * np.prod(shape[2:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = g.uniform(-bound, bound, size=shape)
        elif kind.startswith("b:"):
            bound = 1.0 / np.sqrt(int(kind[2:]))
            v = g.uniform(-bound, bound, size=shape)
        elif kind == "prelu":
            v = np.full(shape, 0.25)
        elif kind == "bn_g":
            v = g.uniform(0.5, 1.5, size=shape)
        elif kind == "bn_b":
            v = g.normal(0.0, 0.1, size=shape)
        elif kind == "bn_m":
            v = g.normal(0.0, 0.1, size=shape)
        elif kind == "bn_v":
            v = g.uniform(0.5, 1.5, size=shape)
        elif kind == "nbt":
            sd[prefix + name] = np.array(0, dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[prefix + name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


# --------------------------------------------------------------------------------------
# Synthetic estimator inputs
# --------------------------------------------------------------------------------------
def _lookat_extrinsic(eye, target, up=np.array([0.0, 0.0, 1.0])):
    """World->camera 4x4 (OpenCV convention: +z forward, +x right, +y down)."""
    f = target - eye
    f = f / np.linalg.norm(f)
    r = np.cross(f, up)
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f], axis=0)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E


def adapose_inputs(B: int, seed: int = 0, img: int = IMG, n_pts: int = N_PTS):
    """Network-level inputs for B poses (stereo pairs).

    Returns dict of float32/int64 ndarrays:
      img1,img2 [B,3,img,img] normalised; choose1,choose2 [B,n_pts] int64;
      P1,P2 [B,4,4] float32 (K'.E[:3] padded); depths [B,24] float32;
      K1 [B,3,3] float64 cropped intrinsics of view 1; E1,E2 [B,4,4] float64.
    """
    g = np.random.default_rng(1234567 + seed)
    yy, xx = np.meshgrid(np.arange(img), np.arange(img), indexing="ij")
    imgs = np.empty((B, 2, 3, img, img), dtype=np.float32)
    choose = np.empty((B, 2, n_pts), dtype=np.int64)
    P = np.empty((B, 2, 4, 4), dtype=np.float32)
    Kc = np.empty((B, 2, 3, 3), dtype=np.float64)
    E = np.empty((B, 2, 4, 4), dtype=np.float64)
    fx = fy = 240.0 / np.tan(0.5)
    for b in range(B):
        target = g.uniform(-0.05, 0.05, size=3) + np.array([0.0, 0.0, 0.5])
        base_dir = g.normal(size=3)
        base_dir[2] = abs(base_dir[2]) * 0.3
        base_dir /= np.linalg.norm(base_dir)
        dist = g.uniform(0.55, 0.9)
        eye1 = target + base_dir * dist
        side = np.cross(base_dir, np.array([0.0, 0.0, 1.0]))
        side /= np.linalg.norm(side)
        eye2 = eye1 + side * g.uniform(0.15, 0.35) + np.array([0, 0, g.uniform(-0.05, 0.05)])
        for v, eye in enumerate((eye1, eye2)):
            rgb = g.random((3, img, img))
            smooth = np.zeros((img, img))
            for _ in range(4):
                fxy = g.uniform(0.5, 3.0, size=2) * 2 * np.pi / img
                ph = g.uniform(0, 2 * np.pi)
                smooth += np.cos(fxy[0] * xx + fxy[1] * yy + ph)
            rgb = np.clip(0.5 * rgb + 0.5 * (0.5 + 0.125 * smooth)[None], 0.0, 1.0)
            rgb = (rgb - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]
            imgs[b, v] = rgb.astype(np.float32)
            cy, cx = g.uniform(80, 144, size=2) * img / 224.0
            ay, ax = g.uniform(30, 90, size=2) * img / 224.0
            mask = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1.0
            idx = np.flatnonzero(mask.ravel())
            idx = idx[g.permutation(idx.size)]
            if idx.size >= n_pts:
                ch = np.sort(idx[:n_pts])
            else:
                ch = np.pad(np.sort(idx), (0, n_pts - idx.size), "wrap")
            choose[b, v] = ch
            # crop window of the 480x640 frame -> cropped intrinsics (interface_v5.py:153-168)
            w = float(g.integers(5, 12) * 40)  # 200..440
            ratio = img / w
            K = np.eye(3)
            K[0, 0] = fx * ratio
            K[1, 1] = fy * ratio
            # the look-at target sits on the optical axis; keep the cropped principal
            # point near the crop centre so the handle projects inside both crops
            K[0, 2] = img / 2 + g.uniform(-20, 20)
            K[1, 2] = img / 2 + g.uniform(-20, 20)
            Ex = _lookat_extrinsic(eye, target)
            Kc[b, v] = K
            E[b, v] = Ex
            Pm = np.eye(4)
            Pm[:3, :] = K @ Ex[:3, :]
            P[b, v] = Pm.astype(np.float32)
    depths = np.tile(np.arange(0.1, 0.1 * (N_DEPTH - 0.5) + 0.1, 0.1, dtype=np.float32)[None], (B, 1))
    return {
        "img1": np.ascontiguousarray(imgs[:, 0]), "img2": np.ascontiguousarray(imgs[:, 1]),
        "choose1": np.ascontiguousarray(choose[:, 0]), "choose2": np.ascontiguousarray(choose[:, 1]),
        "P1": np.ascontiguousarray(P[:, 0]), "P2": np.ascontiguousarray(P[:, 1]),
        "depths": depths,
        "K1": np.ascontiguousarray(Kc[:, 0]), "K2": np.ascontiguousarray(Kc[:, 1]),
        "E1": np.ascontiguousarray(E[:, 0]), "E2": np.ascontiguousarray(E[:, 1]),
    }


# --------------------------------------------------------------------------------------
# PPO
# --------------------------------------------------------------------------------------
def policy_state_dict(seed: int = 0, obs=60, act=12, hid=(96, 96, 32), init_std=0.6):
    """ActorCritic state_dict in the reference's key order (`module.py:24-54`)."""
    sd = OrderedDict()
    sd["log_std"] = np.full((act,), np.log(init_std), dtype=np.float32)
    for net, out_dim, gains in (("actor", act, [np.sqrt(2)] * len(hid) + [0.01]),
                                ("critic", 1, [np.sqrt(2)] * len(hid) + [1.0])):
        dims = (obs,) + tuple(hid) + (out_dim,)
        for li, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            g = _rng(f"{net}.{2 * li}", seed)
            m = g.normal(size=(max(a, b), min(a, b)))
            q, r = np.linalg.qr(m)
            q = q * np.sign(np.diag(r))[None]
            w = q if b >= a else q.T
            sd[f"{net}.{2 * li}.weight"] = np.ascontiguousarray(gains[li] * w[:b, :a], dtype=np.float32)
            bound = 1.0 / np.sqrt(a)
            sd[f"{net}.{2 * li}.bias"] = g.uniform(-bound, bound, size=(b,)).astype(np.float32)
    return sd


def ppo_rollout(T: int, N: int, seed: int = 0, obs=60, states=75, act=12):
    """A recorded rollout with plausible statistics (rewards ~ reference reward scale)."""
    g = np.random.default_rng(777 + seed)
    r = {
        "observations": g.uniform(-1.0, 1.0, size=(T, N, obs)).astype(np.float32),
        "states": g.uniform(-1.0, 1.0, size=(T, N, states)).astype(np.float32),
        "actions": g.normal(0, 0.5, size=(T, N, act)).astype(np.float32),
        "rewards": g.normal(2.0, 3.0, size=(T, N, 1)).astype(np.float32),
        "dones": (g.random((T, N, 1)) < 0.2).astype(np.uint8),
        "values": g.normal(1.0, 2.0, size=(T, N, 1)).astype(np.float32),
        "last_values": g.normal(1.0, 2.0, size=(N, 1)).astype(np.float32),
    }
    return r


# --------------------------------------------------------------------------- ControlInterface view stream
def control_view(num_envs: int, step: int, seed: int = 0, H: int = 480, W: int = 640):
    """One seeded `MultiVecEnv.get_image()` result + camera pose + GT box for the view-queue tests and goldens.

    Colour frames are a per-(step, env) constant plus a faint ramp so a captured frame identifies its origin; masks are
    rectangles, with env 1 empty at step 2 and every env empty at step 4 (the reference's global `if p_env.shape[0]` branch)."""
    rng = np.random.default_rng(seed * 1000 + step)
    color = np.empty((num_envs, H, W, 3), dtype=np.float32)
    ramp = (np.arange(W, dtype=np.float32) / (8.0 * W))[None, :, None]
    mask = np.zeros((num_envs, H, W), dtype=bool)
    for e in range(num_envs):
        color[e] = (step * 16 + e + 1) / 256.0 + ramp
        if step == 4 or (step == 2 and e == 1):
            continue
        r0, c0 = int(rng.integers(20, 300)), int(rng.integers(20, 400))
        mask[e, r0:r0 + int(rng.integers(30, 150)), c0:c0 + int(rng.integers(30, 200))] = True
    K = np.tile(np.array([[439.31, 0, 320.0], [0, 439.31, 240.0], [0, 0, 1.0]]), (num_envs, 1, 1)) + rng.normal(0, 0.1, (num_envs, 3, 3))
    E = np.tile(np.eye(4), (num_envs, 1, 1))
    E[:, :3, :] += rng.normal(0, 0.2, (num_envs, 3, 4))
    pose = rng.normal(0, 0.5, (num_envs, 7))
    gt = rng.normal(0, 0.3, (num_envs, 8, 3))
    image = {"camera0": {"Color": color, "Mask": mask, "Intrinsic": K, "Extrinsic": E}}
    return image, pose, gt


CONTROL_REWARD_CFG = {            # cfg/controller/rl.yaml:11-26, verbatim
    "diff_coef": -0.5, "move_success_coef": 8.0, "move_period_coef": -0.0, "far_coef": -2.5, "ori_coef": 0.25,
    "xyz_lookat_coef": -0.05, "bbox_coef": -1.0, "bbox_boundary_coef": -1.0, "have_bbox_coef": 2.0, "center_coef": 12.0,
    "open_coef": 8.0, "view_coef": 0.5, "view_norm_coef": -0.3, "success_coef": 0.0,
}


def control_cfg(task: str = "cabinet", success_coef: float = 0.0):
    """The slice of the merged yaml config `ControlInterface` reads (cfg/controller/rl.yaml:3-26, cfg/task/*.yaml name)."""
    reward = dict(CONTROL_REWARD_CFG)
    reward["success_coef"] = success_coef
    return {"controller": {"max_steps": 4, "action_type": "pose", "pose_min": [-0.3, -0.3, 0.4], "pose_max": [0.3, 0.3, 1.0],
                           "early_stop": 4},
            "reward": reward, "task": {"name": task}}


def control_actions(num_envs: int, step: int, seed: int = 0, act: int = 12):
    """Seeded policy actions (float32, as `ActorCritic.act` returns them) for the ControlInterface.step goldens; row 0 of
    step 3 carries a large z offset so the pose clip of rl_pose.py:406 is exercised."""
    a = np.random.default_rng(seed * 1000 + 700 + step).normal(0, 0.6, (num_envs, act)).astype(np.float32)
    if step == 3:
        a[0, 2] = 2.5
    return a


class ReplayVecEnv:
    """Deterministic numpy stand-in for the slice of `MultiVecEnv` that `ControlInterface` calls (`env/my_vec_env.py:214,
    266, 281, 382, 466, 482`): every answer is a seeded function of the call count, so the reference class (golden
    generator), the oracle and the device implementation can be driven through identical episodes.  Calls are recorded."""

    def __init__(self, num_envs: int, seed: int = 0):
        self.num_envs, self.seed = num_envs, seed
        self.t = 0
        self.cur = None
        self.moves, self.resets = [], 0
        self._robot = np.random.default_rng(seed * 1000 + 900).normal(0, 0.2, (num_envs, 7))

    def cam_move_to(self, pose, time=2, wait=1, planner="ik", robot_frame=False, skip_move=False, no_collision_with_front=True):
        rng = np.random.default_rng(self.seed * 1000 + 500 + self.t)
        pose = pose.detach().cpu().numpy() if hasattr(pose, "detach") else pose
        self.moves.append(dict(pose=np.array(pose, dtype=np.float64), time=time, wait=wait, planner=planner,
                               robot_frame=robot_frame, skip_move=skip_move, no_collision_with_front=no_collision_with_front))
        return [(rng.random(self.num_envs) > 0.25), rng.integers(1, 2000, self.num_envs)]     # merge_obs of (bool, int) tuples

    def get_image(self, mask="handle"):
        self.cur = control_view(self.num_envs, self.t, self.seed)
        self.t += 1
        return self.cur[0]

    def camera_pose(self, robot_frame=False):
        return self.cur[1]

    def robot_pose(self):
        return self._robot

    def get_observation(self, gt=False):
        rng = np.random.default_rng(self.seed * 1000 + 300 + self.t)
        return {"handle_bbox": self.cur[2], "success": (rng.random((self.num_envs, 1)) > 0.5).astype(np.float64)}

    def reset(self, indices=None):
        self.resets += 1
        return None

    def get_attr(self, attr_name, indices=None):
        """`SubprocVecEnv.get_attr` (`env/my_vec_env.py:466`): one entry per env.  Only `current_obj_config` (read by the
        eval-time dataset export, `rl_pose.py:58`) is known: two object names that change with every reset."""
        if attr_name != "current_obj_config":
            raise AttributeError(attr_name)
        return [{"name": f"obj{(e + self.resets) % 2}"} for e in range(self.num_envs)]


# --------------------------------------------------------------------------- similarity alignment cases
def align_case(case: int, P: int = 1024):
    """Seeded (nocs, camera points) pair for the Umeyama-RANSAC tests: points = s R nocs + t + noise, a fraction of outliers.
    case 3 is pure noise (the reference returns None), case 4 has mirrored points so the SVD's reflection fix is taken."""
    rng = np.random.default_rng(4200 + case)
    nocs = rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32) * np.array([0.9, 0.5, 0.3], dtype=np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    s, t = rng.uniform(0.15, 0.5), rng.uniform(-0.3, 0.3, 3) + np.array([0, 0, 0.9])
    pts = s * nocs.astype(np.float64) @ q.T + t + rng.normal(0, 0.002, (P, 3))
    out_frac = [0.0, 0.2, 0.45, 1.0, 0.1][case % 5]
    bad = rng.random(P) < out_frac
    pts[bad] = rng.uniform(-0.5, 0.5, (int(bad.sum()), 3)) + np.array([0, 0, 0.9])
    if case % 5 == 4:
        pts[:, 0] = -pts[:, 0]
    return nocs, pts


def pnp_case(case: int, P: int = 1024):
    """Seeded two-view case for the `use_depth: False` (NOCS matches -> triangulation -> EPnP-RANSAC) branch of predict
    (interface_v5.py:340-346): an object of scale s at pose (R_o, t_o) in the world, seen by two look-at cameras; per view the
    predicted NOCS of P surface points (+ noise) and their pixels in the 480x640 frame.  View 2 re-observes most of view 1's
    points (so that mutual nearest neighbours in NOCS space exist) in another order; a fraction of view 1's pixels are outliers.
    case 3 has no common points (no match -> default bbox).  Returns a dict incl. the ground truth (scale, R, t) in camera 1."""
    rng = np.random.default_rng(7300 + case)
    dims = np.array([0.9, 0.5, 0.3], dtype=np.float32)
    nocs = (rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32) * dims)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    s = float(rng.uniform(0.25, 0.5))
    t_o = rng.uniform(-0.05, 0.05, 3) + np.array([0.0, 0.0, 0.4])
    world = s * nocs.astype(np.float64) @ q.T + t_o
    eye1 = t_o + np.array([0.75, 0.1, 0.25]) * rng.uniform(0.9, 1.1)
    eye2 = eye1 + np.array([-0.1, 0.3, 0.05]) * rng.uniform(0.8, 1.2)
    E1, E2 = _lookat_extrinsic(eye1, t_o), _lookat_extrinsic(eye2, t_o)
    f = 240.0 / np.tan(0.5)
    K = np.array([[f, 0, 320.0], [0, f, 240.0], [0, 0, 1.0]])

    def project(E, X):
        c = X @ E[:3, :3].T + E[:3, 3]
        uv = c @ K.T
        return uv[:, :2] / uv[:, 2:3]
    noise = [0.0005, 0.001, 0.002, 0.001, 0.001][case % 5]
    pix = [0.2, 0.4, 0.8, 0.4, 0.4][case % 5]
    nocs1 = (nocs + rng.normal(0, noise, nocs.shape)).astype(np.float32)
    pts1 = (project(E1, world) + rng.normal(0, pix, (P, 2))).astype(np.float32)
    bad = rng.random(P) < [0.0, 0.1, 0.25, 0.1, 0.05][case % 5]
    pts1[bad] = rng.uniform([100, 60], [540, 420], (int(bad.sum()), 2)).astype(np.float32)
    perm = rng.permutation(P)
    if case % 5 == 3:
        nocs_b = (rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32) * dims) + np.float32(2.0)      # nothing in common with view 1
        world_b = s * nocs_b.astype(np.float64) @ q.T + t_o
    else:
        nocs_b, world_b = nocs[perm], world[perm]
        fresh = rng.random(P) < 0.3                      # 30 % of view 2 are other surface points
        nocs_b = np.where(fresh[:, None], rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32) * dims, nocs_b).astype(np.float32)
        world_b = np.where(fresh[:, None], s * nocs_b.astype(np.float64) @ q.T + t_o, world_b)
    nocs2 = (nocs_b + rng.normal(0, noise, nocs_b.shape)).astype(np.float32)
    pts2 = (project(E2, world_b) + rng.normal(0, pix, (P, 2))).astype(np.float32)
    R_gt = E1[:3, :3] @ q
    t_gt = E1[:3, :3] @ t_o + E1[:3, 3]
    return dict(nocs1=nocs1, pts1=pts1, nocs2=nocs2, pts2=pts2, K=K, E1=E1, E2=E2, scale=s, R=R_gt, t=t_gt)


# --------------------------------------------------------------------------- a deterministic vec env for PPO.run
class StubVecEnv:
    """A closed-form stand-in for `MultiVecEnv` with the interface `PPO.run` uses (`reset`, `step`, `get_state`, `num_envs`, the three
    spaces): smooth dynamics in float32 on the CPU, per-env episode lengths, a reward, and an info dict with reward terms and the
    specially handled "success_rate" key.  The SAME class drives the reference's `PPO.run` when the golden file is generated
    (tools/make_goldens.py::gen_ppo_run) and the product's in tests/test_gpu_ppo.py, so both see identical environment arithmetic;
    every action it receives is recorded in `action_log`."""

    def __init__(self, num_envs, box, seed=0):
        import torch
        g = np.random.default_rng(4242 + seed)
        self.num_envs = num_envs
        self.observation_space, self.state_space, self.action_space = box(-1.5, 1.5, (60,)), box(-1.5, 1.5, (75,)), box(-1.5, 1.5, (12,))
        self.Wa = torch.from_numpy((g.normal(size=(12, 75)) * 0.3).astype(np.float32))
        self.Ws = torch.from_numpy((g.normal(size=(75, 75)) * (0.5 / np.sqrt(75.0))).astype(np.float32))
        self.Po = torch.from_numpy((g.normal(size=(75, 60)) * (1.0 / np.sqrt(75.0))).astype(np.float32))
        self.s0 = torch.from_numpy(g.uniform(-0.5, 0.5, size=(num_envs, 75)).astype(np.float32))
        self.ep_len = torch.tensor([5 + (e % 7) for e in range(num_envs)], dtype=torch.int64)
        self.s = self.s0.clone()
        self.cnt = torch.zeros(num_envs, dtype=torch.int64)
        self.action_log = []

    def _obs(self):
        import torch
        return torch.tanh(self.s @ self.Po)

    def reset(self):
        self.s = self.s0.clone()
        self.cnt.zero_()
        return self._obs()

    def get_state(self):
        return self.s.clone()

    def step(self, actions):
        import torch
        a = actions.detach().to("cpu", torch.float32)
        self.action_log.append(a.clone())
        a = a.clamp(-1.5, 1.5)
        self.s = 0.85 * self.s + 0.15 * torch.tanh(a @ self.Wa + self.s @ self.Ws)
        self.cnt += 1
        dist = (self.s[:, :3] ** 2).sum(1)
        rew = 1.0 - dist + 0.05 * a.mean(1)
        done = self.cnt >= self.ep_len
        infos = {"REW:dist": dist.clone(), "REW:act": a.abs().mean(1), "successes": (rew > 0.8).float(),
                 "success_rate": (dist < 0.2).float()}
        self.s = torch.where(done[:, None], self.s0, self.s)
        self.cnt = torch.where(done, torch.zeros_like(self.cnt), self.cnt)
        return self._obs(), rew, done.to(torch.int64), infos
