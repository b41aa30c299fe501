"""-m gpu: the Umeyama-RANSAC tail of predict (`direct_regression: False`, `use_depth: True`) on the device
against the oracle fed the same hash sample stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rgbmanip_amd import synth  # noqa: E402


def _case(b, P=1024, S=224, outliers=0.2, noise=0.002, mirror=False):
    """nocs / depth / choose / K such that the back-projected points are a similarity image of nocs (+ noise, outliers)."""
    rng = np.random.default_rng(900 + b)
    K = np.array([[430.0 + 20 * rng.random(), 0, 110 + 5 * rng.random()], [0, 425.0 + 20 * rng.random(), 112 + 5 * rng.random()], [0, 0, 1]])
    choose = np.sort(rng.choice(S * S, P, replace=False)).astype(np.int32)
    depth = rng.uniform(0.5, 1.2, P).astype(np.float32)
    x, y, z = choose % S, choose // S, depth.astype(np.float64)
    pts = np.stack([(x - K[0, 2]) * z / K[0, 0], (y - K[1, 2]) * z / K[1, 1], z], axis=1)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    s, t = rng.uniform(0.8, 2.0), pts.mean(0)
    nocs = ((pts - t) @ q) / s + rng.normal(0, noise, (P, 3))            # pts = s q nocs + t
    bad = rng.random(P) < outliers
    nocs[bad] = rng.uniform(-0.5, 0.5, (int(bad.sum()), 3))
    if mirror:
        nocs[:, 0] = -nocs[:, 0]
    E = np.eye(4)
    E[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    E[:3, 3] = rng.normal(0, 0.3, 3)
    return nocs.astype(np.float32), depth, choose, K, E


def test_umeyama_ransac_matches_oracle():
    from oracle import align_ref as ar
    from rgbmanip_amd.adapose import postprocess_ransac
    cfgs = [dict(), dict(outliers=0.0, noise=0.0), dict(outliers=0.45), dict(outliers=1.0), dict(mirror=True, outliers=0.05),
            dict(P=200), dict(P=37, outliers=0.1)]
    seed = 17
    for P in (1024, 200, 37):
        group = [(b, c) for b, c in enumerate(cfgs) if c.get("P", 1024) == P]
        data = [_case(b, **c) for b, c in group]
        dev = lambda i, dt: torch.from_numpy(np.stack([d[i] for d in data])).to("cuda", dt)
        bbox, srt, valid = postprocess_ransac(dev(0, torch.float32), dev(1, torch.float32), dev(2, torch.int32), dev(3, torch.float64),
                                              dev(4, torch.float64), img_size=224, seed=seed)
        bbox, srt, valid = bbox.cpu().numpy(), srt.cpu().numpy(), valid.cpu().numpy()
        for j, ((b, c), d) in enumerate(zip(group, data)):
            want, (s, R, t) = ar.bbox_world_ransac(d[0], d[1], d[2], d[3], d[4], 224, ar.hash_sampler(seed, j))
            if s is None:
                assert valid[j] == 0 and np.isnan(srt[j, 0]), (b, c)
                np.testing.assert_array_equal(bbox[j], ar.DEFAULT_BBOX)
                continue
            assert valid[j] == 1, (b, c)
            np.testing.assert_allclose(srt[j, 0], s, rtol=1e-10, err_msg=str(c))
            np.testing.assert_allclose(srt[j, 1:10].reshape(3, 3), R, rtol=0, atol=1e-10, err_msg=str(c))
            np.testing.assert_allclose(srt[j, 10:], t, rtol=0, atol=1e-10, err_msg=str(c))
            np.testing.assert_allclose(bbox[j], want, rtol=0, atol=1e-9, err_msg=str(c))
    # sanity: the clean case recovers the planted similarity
    nocs, depth, choose, K, E = _case(1, outliers=0.0, noise=0.0)
    _, (s, R, t) = ar.bbox_world_ransac(nocs, depth, choose, K, E, 224, ar.hash_sampler(seed, 0))
    assert abs(np.linalg.det(R) - 1) < 1e-9 and 0.8 <= s <= 2.0


def test_umeyama_ransac_nan_input_gives_default_bbox():
    from oracle import align_ref as ar
    from rgbmanip_amd.adapose import postprocess_ransac
    nocs, depth, choose, K, E = _case(3)
    nocs = np.stack([nocs, nocs]); nocs[1, 5, 1] = np.nan
    t = lambda a, dt: torch.from_numpy(a).to("cuda", dt)
    bbox, srt, valid = postprocess_ransac(t(nocs, torch.float32), t(np.stack([depth] * 2), torch.float32), t(np.stack([choose] * 2), torch.int32),
                                          t(np.stack([K] * 2), torch.float64), t(np.stack([E] * 2), torch.float64), seed=1)
    assert valid.cpu().tolist() == [1, 0]
    np.testing.assert_array_equal(bbox[1].cpu().numpy(), ar.DEFAULT_BBOX)


def test_estimator_ransac_branch_end_to_end():
    """AdaPoseEstimator_v5 with `direct_regression: False`: network (HIP) -> Umeyama-RANSAC tail (HIP) equals the oracle tail
    applied to the same network outputs."""
    from oracle import align_ref as ar
    from rgbmanip_amd.adapose import postprocess_ransac
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, direct_regression=False, use_depth=True, hip_ransac_seed=5)
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="fp32")
    inp = synth.adapose_inputs(2, seed=3)
    pred = est.estimator(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
    got = est._bbox_tail(pred, inp["choose1"], inp["K1"], inp["E1"]).cpu().numpy()
    for b in range(2):
        want, _ = ar.bbox_world_ransac(pred["view1_nocs"][b].cpu().numpy(), pred["view1_depth"][b].cpu().numpy(), inp["choose1"][b],
                                       inp["K1"][b], inp["E1"][b], 224, ar.hash_sampler(5, b))
        np.testing.assert_allclose(got[b], want, rtol=0, atol=1e-8)
    # (`use_depth: False`, the PnP branch, is covered by tests/test_pnp.py)
