"""CPU restatement (numpy, float64) of the `direct_regression: False`, `use_depth: False` tail of AdaPoseEstimator_v5.predict —
TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference implements this branch with OpenCV calls (`cv2.triangulatePoints`, `cv2.solvePnPRansac(flags=
SOLVEPNP_EPNP, reprojectionError=3.0)`, `cv2.solvePnPRefineVVS`, `cv2.Rodrigues`); OpenCV is not installed in the build image and
not vendored by the reference (requirements.txt: opencv-python, unpinned), so nothing here could be checked against cv2 outputs.
What is restated is (a) the reference's own numpy code around those calls, line by line, and (b) the published algorithms behind
them as OpenCV 4.x implements them (modules/calib3d/src: triangulate.cpp, solvepnp.cpp, ptsetreg.cpp, epnp.cpp):
  * depth_estimation_from_nocs_matches            lib/utils.py:121-195   (mutual NOCS nearest neighbours, 0.01 distance gate, epipolar
                                                                          gate through F = K^-T [t]x R K^-1 with a float32 [t]x,
                                                                          DLT triangulation, compute_scale of the left points)
  * compute_scale                                  lib/utils.py:76-96
  * estimatePnPRansac                              lib/align.py:104-115
  * the glue                                       interface_v5.py:340-374
  * triangulate_points: per point the 4x4 DLT system [x P3 - P1; y P3 - P2] of both views, null vector by SVD
  * epnp: Lepetit / Moreno-Noguer / Fua, the four-control-point formulation, three beta approximations + 5 Gauss-Newton steps
  * RANSAC: 5-point EPnP models, squared reprojection error <= 3^2, adaptive iteration count (confidence 0.99, at most 100),
    final EPnP over the inliers.  OpenCV draws the subsets from its own MWC generator (cv::theRNG()), whose state depends on every
    earlier OpenCV call of the process; here subsets come from the package's seeded hash (`mix32`), like the Umeyama branch.
  * refine_vvs: virtual visual servoing (Marchand et al.), lambda = 1, at most 20 iterations, 1e-6 on the residual change;
    it runs over ALL points from the RANSAC pose, so the end result is the least-squares pose whatever subsets RANSAC drew
    (as long as the start is inside the basin).
The device kernel (csrc/pnp.hip) is tested against this file under the shared hash stream.
"""
import numpy as np

from .postproc_ref import DEFAULT_BBOX, compute_scale, mix32

RANSAC_ITERS, RANSAC_CONF, REPROJ_ERR, MODEL_POINTS = 100, 0.99, 3.0, 5


# ------------------------------------------------------------------------------------------------ lib/utils.py:121-195
def triangulate_points(P1, P2, x1, x2):
    """cv2.triangulatePoints: P1, P2 [3,4]; x1, x2 [2,N] -> homogeneous [4,N] (unit-norm null vectors of the DLT systems)."""
    n = x1.shape[1]
    X = np.zeros((4, n))
    for i in range(n):
        A = np.stack([x1[0, i] * P1[2] - P1[0], x1[1, i] * P1[2] - P1[1], x2[0, i] * P2[2] - P2[0], x2[1, i] * P2[2] - P2[1]])
        X[:, i] = np.linalg.svd(A)[2][3]
    return X


def nocs_matches(left_pts2d, left_nocs, right_pts2d, right_nocs, left_pose, right_pose, K):
    """utils.py:124-180: indices of the mutually nearest, close (< 0.01) and epipolar-consistent (< 1.0) NOCS matches."""
    dis = np.linalg.norm(left_nocs[:, None, :] - right_nocs[None, :, :], axis=-1)
    l2r = np.argmin(dis, axis=1)
    r2l = np.argmin(dis, axis=0)
    left_id = np.arange(left_nocs.shape[0])
    ok = r2l[l2r] == left_id
    ml = left_id[ok]
    mr = l2r[ml]
    ok = dis[ml, mr] < 0.01
    ml, mr = ml[ok], mr[ok]
    rel = left_pose @ np.linalg.inv(right_pose)
    R1, t1 = rel[:3, :3], rel[:3, 3]
    tx = np.zeros((3, 3)).astype(np.float32)               # float32 skew matrix, as in the reference
    tx[0, 1], tx[1, 0], tx[0, 2], tx[2, 0], tx[1, 2], tx[2, 1] = -t1[2], t1[2], t1[1], -t1[1], -t1[0], t1[0]
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R1 @ Kinv
    lh = np.ones((3, len(ml)))
    rh = np.ones((3, len(ml)))
    lh[:2] = left_pts2d[ml].T
    rh[:2] = right_pts2d[mr].T
    epi = np.abs(np.einsum("in,ij,jn->n", lh, F, rh))
    ok = epi < 1.0
    return ml[ok], mr[ok]


def left_scale_from_matches(left_pts2d, left_nocs, left_proj, left_pose, right_pts2d, right_nocs, right_proj, right_pose, K):
    """utils.py:121-195 -> (left_scale, n_matches); left_scale is NaN without a valid pair (np.median of an empty list)."""
    ml, mr = nocs_matches(left_pts2d, left_nocs, right_pts2d, right_nocs, left_pose, right_pose, K)
    if len(ml) == 0:
        return float("nan"), 0
    X = triangulate_points(left_proj[:3], right_proj[:3], left_pts2d[ml].T.astype(np.float64), right_pts2d[mr].T.astype(np.float64))
    X = X / X[3]
    left_pts = (left_pose @ X)[:3].T
    with np.errstate(all="ignore"):
        return float(compute_scale(left_pts, left_nocs[ml])), len(ml)


# ------------------------------------------------------------------------------------------------ EPnP (OpenCV epnp.cpp)
def _solve_ls(A, b):
    return np.linalg.lstsq(A, b, rcond=None)[0]


def epnp(pw, uv, K):
    """pw [n,3] object points, uv [n,2] pixels, K [3,3] -> (R [3,3], t [3]) of the best of the three beta approximations."""
    n = pw.shape[0]
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    # control points: centroid + principal directions scaled by sqrt(eigenvalue / n)
    cws = np.zeros((4, 3))
    cws[0] = pw.mean(axis=0)
    pw0 = pw - cws[0]
    dc, uct = np.linalg.eigh(pw0.T @ pw0)                   # ascending; OpenCV's SVD gives them descending
    dc, uct = dc[::-1], uct[:, ::-1].T
    for i in range(1, 4):
        cws[i] = cws[0] + np.sqrt(max(dc[i - 1], 0.0) / n) * uct[i - 1]
    CC = (cws[1:] - cws[0]).T                               # columns = control axes
    al = np.zeros((n, 4))
    al[:, 1:] = (np.linalg.inv(CC) @ (pw - cws[0]).T).T
    al[:, 0] = 1.0 - al[:, 1:].sum(axis=1)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - uv[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - uv[:, 1])
    w, V = np.linalg.eigh(M.T @ M)                           # ascending: V[:, 0] is OpenCV's ut[11]
    v = [V[:, k] for k in range(4)]                          # v[0] = smallest eigenvalue
    pairs = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
    dv = [[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in pairs] for i in range(4)]
    L = np.zeros((6, 10))
    for r in range(6):
        d = [dv[i][r] for i in range(4)]
        L[r] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2],
                2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3], d[3] @ d[3]]
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for (a, b) in pairs])

    def approx1():
        b4 = _solve_ls(L[:, [0, 1, 3, 6]], rho)
        s = -1.0 if b4[0] < 0 else 1.0
        b0 = np.sqrt(s * b4[0])
        return np.array([b0, s * b4[1] / b0, s * b4[2] / b0, s * b4[3] / b0])

    def approx2():
        b3 = _solve_ls(L[:, [0, 1, 2]], rho)
        if b3[0] < 0:
            b = [np.sqrt(-b3[0]), np.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
        else:
            b = [np.sqrt(b3[0]), np.sqrt(b3[2]) if b3[2] > 0 else 0.0]
        if b3[1] < 0:
            b[0] = -b[0]
        return np.array([b[0], b[1], 0.0, 0.0])

    def approx3():
        b5 = _solve_ls(L[:, [0, 1, 2, 3, 4]], rho)
        if b5[0] < 0:
            b = [np.sqrt(-b5[0]), np.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
        else:
            b = [np.sqrt(b5[0]), np.sqrt(b5[2]) if b5[2] > 0 else 0.0]
        if b5[1] < 0:
            b[0] = -b[0]
        return np.array([b[0], b[1], b5[3] / b[0], 0.0])

    def gauss_newton(b):
        b = b.copy()
        for _ in range(5):
            A = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                          L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                          L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                          L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], axis=1)
            bb = rho - (L[:, 0] * b[0] * b[0] + L[:, 1] * b[0] * b[1] + L[:, 2] * b[1] * b[1] + L[:, 3] * b[0] * b[2] +
                        L[:, 4] * b[1] * b[2] + L[:, 5] * b[2] * b[2] + L[:, 6] * b[0] * b[3] + L[:, 7] * b[1] * b[3] +
                        L[:, 8] * b[2] * b[3] + L[:, 9] * b[3] * b[3])
            b = b + _solve_ls(A, bb)
        return b

    def R_and_t(b):
        ccs = np.zeros((4, 3))
        for i in range(4):
            for j in range(4):
                ccs[j] += b[i] * v[i][3 * j:3 * j + 3]
        pcs = al @ ccs
        if pcs[0, 2] < 0.0:
            ccs, pcs = -ccs, -pcs
        pc0, pw_0 = pcs.mean(axis=0), pw.mean(axis=0)
        ABt = (pcs - pc0).T @ (pw - pw_0)
        U, _, Vt = np.linalg.svd(ABt)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw_0
        Xc = pw @ R.T + t
        ue, ve = uc + fu * Xc[:, 0] / Xc[:, 2], vc + fv * Xc[:, 1] / Xc[:, 2]
        err = np.mean(np.sqrt((uv[:, 0] - ue) ** 2 + (uv[:, 1] - ve) ** 2))
        return err, R, t

    with np.errstate(all="ignore"):
        cands = [R_and_t(gauss_newton(f())) for f in (approx1, approx2, approx3)]
    errs = [c[0] if np.isfinite(c[0]) else np.inf for c in cands]
    k = int(np.argmin(errs))
    return cands[k][1], cands[k][2]


def reproj_sq_err(pw, uv, K, R, t):
    Xc = pw @ R.T + t
    with np.errstate(all="ignore"):
        u = K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2]
        v = K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]
    return (uv[:, 0] - u) ** 2 + (uv[:, 1] - v) ** 2


def hash_subset(seed, pose, it, n):
    """five DISTINCT indices of iteration `it` of pose `pose`: draws mix32(seed, pose * 128 + it, k) % n for k = 0, 1, ... and keeps
    the first five that differ (csrc/pnp.hip draws the same way)."""
    out, k = [], 0
    while len(out) < MODEL_POINTS:
        idx = int(mix32(seed, pose * 128 + it, np.array([k]))[0] % np.uint32(n))
        k += 1
        if idx not in out:
            out.append(idx)
    return np.array(out)


def solve_pnp_ransac(pw, uv, K, seed=0, pose=0):
    """cv2.solvePnPRansac(flags=SOLVEPNP_EPNP, reprojectionError=3.0): (ok, R, t, inlier mask)."""
    n = pw.shape[0]
    thr = REPROJ_ERR * REPROJ_ERR
    best, best_mask, niters, it = 0, None, RANSAC_ITERS, 0
    while it < niters:
        sub = hash_subset(seed, pose, it, n)
        it += 1
        R, t = epnp(pw[sub], uv[sub], K)
        if not (np.isfinite(R).all() and np.isfinite(t).all()):
            continue
        mask = reproj_sq_err(pw, uv, K, R, t) <= thr
        good = int(mask.sum())
        if good > max(best, MODEL_POINTS - 1):
            best, best_mask = good, mask
            ep = (n - good) / n
            num, den = max(1.0 - RANSAC_CONF, np.finfo(float).tiny), 1.0 - (1.0 - ep) ** MODEL_POINTS
            if den < np.finfo(float).tiny:
                niters = 0
            else:
                num, den = np.log(num), np.log(den)
                niters = niters if (den >= 0 or -num >= niters * (-den)) else int(np.rint(num / den))
    if best_mask is None:
        return False, None, None, None
    R, t = epnp(pw[best_mask], uv[best_mask], K)
    return True, R, t, best_mask


def rodrigues_to_R(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def refine_vvs(pw, uv, K, R, t, iters=20, eps=1e-6, lam=1.0):
    """cv2.solvePnPRefineVVS: Gauss-Newton on SE(3) over the normalised image coordinates, update cMo <- exp(-dq)^-1 ... written as
    the left-multiplied inverse exponential of the velocity twist dq = -lambda pinv(L) e."""
    xd = (uv[:, 0] - K[0, 2]) / K[0, 0]
    yd = (uv[:, 1] - K[1, 2]) / K[1, 1]
    prev = np.inf
    for _ in range(iters):
        Xc = pw @ R.T + t
        Z = Xc[:, 2]
        x, y = Xc[:, 0] / Z, Xc[:, 1] / Z
        e = np.empty(2 * len(x))
        e[0::2], e[1::2] = x - xd, y - yd
        err = np.sqrt(e @ e / len(x))
        if abs(err - prev) < eps:
            break
        prev = err
        Lm = np.zeros((2 * len(x), 6))
        Lm[0::2] = np.stack([-1 / Z, 0 * Z, x / Z, x * y, -(1 + x * x), y], axis=1)
        Lm[1::2] = np.stack([0 * Z, -1 / Z, y / Z, 1 + y * y, -x * y, -x], axis=1)
        dq = -lam * (np.linalg.pinv(Lm) @ e)
        # the camera moves by the twist dq for one unit of time: cMo <- exp(dq)^-1 cMo
        w = dq[3:]
        Rw = rodrigues_to_R(w)
        th = np.linalg.norm(w)
        if th < 1e-12:
            Vm = np.eye(3)
        else:
            k = w / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            Vm = np.eye(3) + (1 - np.cos(th)) / th * Kx + (1 - np.sin(th) / th) * (Kx @ Kx)
        dt = Vm @ dq[:3]
        R = Rw.T @ R
        t = Rw.T @ (t - dt)
    return R, t


# ------------------------------------------------------------------------------------------------ interface_v5.py:340-374
def pnp_bbox_world(nocs1, pts2d1, nocs2, pts2d2, K, E1, E2, seed=0, pose=0):
    """The `use_depth: False` branch of predict for one pose: NOCS matches -> left scale -> EPnP-RANSAC + VVS on (nocs * scale,
    pixels of view 1, the ORIGINAL intrinsics) -> box in the world frame.  Returns (bbox [8,3], dict of intermediates)."""
    P1, P2 = np.eye(4), np.eye(4)
    P1[:3], P2[:3] = K @ E1[:3], K @ E2[:3]
    scale, nm = left_scale_from_matches(pts2d1, nocs1, P1, E1, pts2d2, nocs2, P2, E2, K)
    info = {"scale": scale, "n_matches": nm}
    if not np.isfinite(scale):
        return DEFAULT_BBOX.copy(), info
    pw = (nocs1.astype(np.float32) * np.float32(scale)).astype(np.float64)      # align.py:105 on float32 nocs
    uv = pts2d1.astype(np.float32).astype(np.float64)
    ok, R, t, mask = solve_pnp_ransac(pw, uv, K, seed, pose)
    info["ransac_ok"] = ok
    if not ok:
        return DEFAULT_BBOX.copy(), info
    info["n_inliers"] = int(mask.sum())
    R, t = refine_vvs(pw, uv, K, R, t)
    info.update(R=R, t=t)
    from .align_ref import bbox_from_srt
    return bbox_from_srt(nocs1, scale, R, t, E1), info
