"""CPU restatement (numpy, float64) of the synthetic camera kernels — TEST INFRASTRUCTURE ONLY.

There is no reference code behind this file: the reference renders with SAPIEN (env/sapien_envs/base_manipulation.py:653-687),
which is out of scope and absent on the GPU box.  rgbmanip_amd/csrc/synth_env.hip is the build's own stand-in producing the
`MultiVecEnv.get_image()` record; this module restates its arithmetic expression by expression so the two agree bit for bit
(both avoid fused multiply-adds and use IEEE division / sqrt), and adds the geometric self-checks the tests use
(`project_points`: the mask must be the silhouette of the ground-truth box under K and E).
"""
import numpy as np


def camera_ref(cam_pose, robot_pose, box, fx, fy, cx, cy):
    """-> K [N,3,3], E [N,4,4], rays [N,12] (synth_camera_kernel)."""
    cp, rp, bx = (np.asarray(a, dtype=np.float64) for a in (cam_pose, robot_pose, box))
    N = cp.shape[0]
    qn = np.sqrt(cp[:, 3] * cp[:, 3] + cp[:, 4] * cp[:, 4] + cp[:, 5] * cp[:, 5] + cp[:, 6] * cp[:, 6])
    w, x, y, z = cp[:, 3] / qn, cp[:, 4] / qn, cp[:, 5] / qn, cp[:, 6] / qn
    f = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], axis=1)
    l = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], axis=1)
    u = np.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], axis=1)
    p = rp[:, :3] + cp[:, :3]
    R = np.stack([-l, -u, f], axis=1)                                    # rows: right, down, forward
    E = np.zeros((N, 4, 4))
    E[:, :3, :3] = R
    E[:, :3, 3] = -((R[:, :, 0] * p[:, None, 0] + R[:, :, 1] * p[:, None, 1]) + R[:, :, 2] * p[:, None, 2])
    E[:, 3, 3] = 1
    K = np.zeros((N, 3, 3))
    K[:, 0, 0], K[:, 0, 2], K[:, 1, 1], K[:, 1, 2], K[:, 2, 2] = fx, cx, fy, cy, 1
    rays = np.zeros((N, 12))
    c = bx[:, 0:3]
    for a in range(3):
        ax = bx[:, 3 + 3 * a: 6 + 3 * a]
        rays[:, a] = (ax[:, 0] * (p[:, 0] - c[:, 0]) + ax[:, 1] * (p[:, 1] - c[:, 1])) + ax[:, 2] * (p[:, 2] - c[:, 2])
        for k in range(3):
            rays[:, 3 + a * 3 + k] = (ax[:, 0] * R[:, k, 0] + ax[:, 1] * R[:, k, 1]) + ax[:, 2] * R[:, k, 2]
    return K, E, rays


_BASE = np.array([[0.85, 0.30, 0.25], [0.25, 0.80, 0.35], [0.30, 0.40, 0.90]])


def render_ref(rays, box, fx, fy, cx, cy, H, W, env0=0):
    """-> color [N,H,W,3] float32, mask [N,H,W] uint8 (synth_render_kernel)."""
    rays, box = np.asarray(rays, dtype=np.float64), np.asarray(box, dtype=np.float64)
    N = rays.shape[0]
    color = np.zeros((N, H, W, 3), dtype=np.float32)
    mask = np.zeros((N, H, W), dtype=np.uint8)
    uu, vv = np.meshgrid(np.arange(W), np.arange(H))
    xn, yn = (uu.astype(np.float64) - cx) / fx, (vv.astype(np.float64) - cy) / fy
    for e in range(N):
        h = box[e, 12:15]
        tmin = np.full((H, W), -np.inf); tmax = np.full((H, W), np.inf)
        face = np.zeros((H, W), dtype=np.int64)
        o, d = [], []
        with np.errstate(divide="ignore", invalid="ignore"):
            for a in range(3):
                oa = rays[e, a]
                da = (rays[e, 3 + a * 3] * xn + rays[e, 4 + a * 3] * yn) + rays[e, 5 + a * 3]
                t1, t2 = (-h[a] - oa) / da, (h[a] - oa) / da
                nz = da != 0.0
                inside = abs(oa) <= h[a]
                tn = np.where(nz, np.minimum(t1, t2), -np.inf if inside else np.inf)
                tf = np.where(nz, np.maximum(t1, t2), np.inf)
                upd = tn > tmin
                tmin = np.where(upd, tn, tmin); face = np.where(upd, a, face)
                tmax = np.minimum(tmax, tf)
                o.append(oa); d.append(da)
        hit = (tmax >= tmin) & (tmin > 0.0)
        a1 = np.where(face == 0, 1, 0); a2 = np.where(face == 2, 1, 2)
        dstack = np.stack(d); ovec = np.array(o)
        d1 = np.take_along_axis(dstack, a1[None], 0)[0]; d2 = np.take_along_axis(dstack, a2[None], 0)[0]
        tm = np.where(hit, tmin, 0.0)
        p1, p2 = ovec[a1] + tm * d1, ovec[a2] + tm * d2
        cell = np.floor(p1 * 50.0).astype(np.int64) + np.floor(p2 * 50.0).astype(np.int64)
        shade = np.where(cell & 1, 1.0, 0.6)
        fg = (_BASE[face] * shade[..., None]).astype(np.float32)
        g = env0 + e
        m0 = (uu * 7 + vv * 3 + g * 31) % 97; m1 = (uu * 2 + vv * 5 + g * 17) % 89; m2 = ((uu >> 3) + (vv >> 3) + g) % 13
        bg = np.stack([0.20 + 0.5 * (m0 / 97.0), 0.25 + 0.4 * (m1 / 89.0), 0.15 + 0.6 * (m2 / 13.0)], axis=-1).astype(np.float32)
        color[e] = np.where(hit[..., None], fg, bg)
        mask[e] = hit
    return color, mask


def box_corners(box):
    """Ground-truth corners [N,8,3] in the order the reference's handle_bbox uses (open_cabinet.py:153-158:
    centre = (b0+b6)/2, x = b1-b0, y = b0-b2, z = b4-b0)."""
    box = np.asarray(box, dtype=np.float64)
    c, X, Y, Z, h = box[:, 0:3], box[:, 3:6], box[:, 6:9], box[:, 9:12], box[:, 12:15]
    hx, hy, hz = h[:, 0:1] * X, h[:, 1:2] * Y, h[:, 2:3] * Z
    b0 = c - hx + hy - hz
    return np.stack([b0, b0 + 2 * hx, b0 - 2 * hy, b0 + 2 * hx - 2 * hy, b0 + 2 * hz, b0 + 2 * hx + 2 * hz,
                     b0 + 2 * hx - 2 * hy + 2 * hz, b0 - 2 * hy + 2 * hz], axis=1)


def project_points(K, E, pts):
    """World points [N,M,3] -> pixel (u, v) [N,M,2] and camera depth [N,M]."""
    cam = np.einsum("nij,nmj->nmi", E[:, :3, :3], pts) + E[:, None, :3, 3]
    uv = np.einsum("nij,nmj->nmi", K, cam / cam[..., 2:3])
    return uv[..., :2], cam[..., 2]
