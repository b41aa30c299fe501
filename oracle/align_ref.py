"""CPU restatement (numpy, float64) of the `direct_regression: False`, `use_depth: True` tail of AdaPoseEstimator_v5.predict —
TEST INFRASTRUCTURE ONLY.

Follows /root/reference/models/pose_estimator/AdaPose/lib/align.py:
  estimateSimilarityUmeyama :10-41, estimateSimilarityTransform (RANSAC, 128 iterations of 5-point samples) :44-104,
and /root/reference/models/pose_estimator/AdaPose/interface_v5.py:322-339 (back-projection of the predicted depth),
:348-374 (bbox from scale / rotation / translation, world transform, default bbox).
The reference draws the 5-point samples from the global `np.random`; `similarity_ransac` takes the sampler as an argument:
`numpy_sampler()` reproduces the reference stream (pinned by tests/golden/align.npz, recorded from the reference function
under np.random.seed), `hash_sampler(seed, pose)` is the reproducible stream the device kernel uses (csrc/align.hip).
The third branch of predict (`use_depth: False`: cv2.solvePnPRansac) needs OpenCV and is not restated.
"""
import numpy as np

from .postproc_ref import BBOX_SIGNS, DEFAULT_BBOX, mix32

MAX_ITER, CONFIDENCE = 128, 0.99


def umeyama(SourceHom, TargetHom):
    """align.py:10-41.  Raises RuntimeError on NaN input like the reference."""
    SourceCentroid = np.mean(SourceHom[:3, :], axis=1)
    TargetCentroid = np.mean(TargetHom[:3, :], axis=1)
    n = SourceHom.shape[1]
    CenteredSource = SourceHom[:3, :] - SourceCentroid[:, None]
    CenteredTarget = TargetHom[:3, :] - TargetCentroid[:, None]
    Cov = np.matmul(CenteredTarget, CenteredSource.T) / n
    if np.isnan(Cov).any():
        raise RuntimeError("There are NANs in the input.")
    U, D, Vh = np.linalg.svd(Cov, full_matrices=True)
    if (np.linalg.det(U) * np.linalg.det(Vh)) < 0.0:
        D[-1] = -D[-1]
        U[:, -1] = -U[:, -1]
    Rotation = np.matmul(U, Vh)
    varP = np.var(SourceHom[:3, :], axis=1).sum()
    Scale = 1 / varP * np.sum(D)
    Translation = TargetHom[:3, :].mean(axis=1) - SourceHom[:3, :].mean(axis=1).dot(Scale * Rotation.T)
    T = np.identity(4)
    T[:3, :3] = Scale * Rotation
    T[:3, 3] = Translation
    return Scale, Rotation, Translation, T


def numpy_sampler():
    return lambda i, n: np.random.randint(n, size=5)


def hash_sampler(seed, pose):
    """Sample k of iteration i of pose b: mix32(seed, b * 128 + i, k) % n  (csrc/align.hip)."""
    return lambda i, n: (mix32(seed, pose * MAX_ITER + i, np.arange(5)) % np.uint32(n)).astype(np.int64)


def similarity_ransac(source, target, sampler=None):
    """align.py:44-104 -> (Scale, Rotation, Translation, OutTransform) or (None,)*4."""
    sampler = sampler or numpy_sampler()
    n = source.shape[0]
    SourceHom = np.transpose(np.hstack([source, np.ones([n, 1])]))
    TargetHom = np.transpose(np.hstack([target, np.ones([n, 1])]))
    centred = SourceHom[:3, :] - np.mean(SourceHom[:3, :], axis=1)[:, None]
    InlierT = 2 * np.amax(np.linalg.norm(centred, axis=0)) / 10.0
    best_ratio, best_idx = 0, np.arange(n)
    for i in range(MAX_ITER):
        idx = sampler(i, n)
        Scale, _, _, T = umeyama(SourceHom[:, idx], TargetHom[:, idx])
        Diff = TargetHom - np.matmul(T, SourceHom)
        inl = np.where(np.linalg.norm(Diff[:3, :], axis=0) < Scale * InlierT)[0]
        ratio = inl.shape[0] / n
        if ratio > best_ratio:
            best_ratio, best_idx = ratio, inl
        if (1 - (1 - best_ratio ** 5) ** i) > CONFIDENCE:
            break
    if best_ratio < 0.1:
        return None, None, None, None
    return umeyama(SourceHom[:, best_idx], TargetHom[:, best_idx])


def backproject(depth, choose, K, img_size):
    """interface_v5.py:322-337: camera-space points of the chosen pixels from the predicted depth."""
    choose = np.asarray(choose)
    x = (choose % img_size)[:, None]
    y = (choose // img_size)[:, None]
    pt2 = np.asarray(depth).flatten()[:, None]
    pt0 = (x - K[0, 2]) * pt2 / K[0, 0]
    pt1 = (y - K[1, 2]) * pt2 / K[1, 1]
    return np.concatenate((pt0, pt1, pt2), axis=1)


def bbox_from_srt(nocs, ts, tr, tt, E1):
    """interface_v5.py:348-374 for one pose."""
    if ts is None:
        return DEFAULT_BBOX.copy()
    with np.errstate(all="ignore"):
        half = np.max(np.abs(nocs), axis=0)
        size = 2 * half * ts
        bbox = (BBOX_SIGNS * (size[None, :] / 2)).T
        sRT = np.eye(4).astype(np.float32)
        sRT[:3, :3] = tr
        sRT[:3, 3] = np.asarray(tt).flatten()
        homo = np.vstack([bbox, np.ones((1, 8), dtype=np.float32)])
        nb = sRT @ homo
        nb = nb[:3] / nb[3]
        try:
            ex_inv = np.linalg.inv(E1)
        except np.linalg.LinAlgError:
            return DEFAULT_BBOX.copy()
        if np.isfinite(ex_inv).all() and np.isfinite(nb).all():
            return (ex_inv[:3, :3] @ nb + ex_inv[:3, 3:4]).T
        return DEFAULT_BBOX.copy()


def bbox_world_ransac(nocs, depth, choose, K, E1, img_size=224, sampler=None):
    pts = backproject(depth, choose, K, img_size)
    try:
        ts, tr, tt, _ = similarity_ransac(nocs, pts, sampler)
    except RuntimeError:
        return DEFAULT_BBOX.copy(), (None, None, None)          # the reference raises here; the device marks the pose invalid
    return bbox_from_srt(nocs, ts, tr, tt, E1), (ts, tr, tt)
