"""-m gpu: end-to-end parity of the HIP AdaPose forward (through the C ABI) against
 (a) the committed golden vectors generated from the reference module itself and
 (b) the CPU oracle on the same seeded inputs, including intermediates for bisecting."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rgbmanip_amd import synth  # noqa: E402

RTOL_FP32 = 1e-4            # north_star: 1e-4 relative, fp32
OUT_KEYS = ["view1_nocs", "view2_nocs", "view1_depth", "view2_depth", "view1_r", "view2_r", "view1_t", "view2_t",
            "view1_s", "view2_s"]


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.fixture(scope="module")
def inputs():
    return synth.adapose_inputs(2, seed=0)


@pytest.fixture(scope="module")
def oracle_taps(inputs):
    from oracle import adapose_ref
    sd = adapose_ref.to_torch_sd(synth.adapose_state_dict(seed=0))
    t = {k: torch.from_numpy(v) for k, v in inputs.items()}
    taps = {}
    out = adapose_ref.adapose_forward(sd, t["img1"], t["choose1"], t["img2"], t["choose2"], t["P1"], t["P2"], t["depths"],
                                      taps=taps)
    return out, taps


def _net(dtype, **kw):
    from rgbmanip_amd.adapose import AdaPoseNet
    return AdaPoseNet(synth.adapose_state_dict(seed=0, prefix="module."), dtype=dtype, **kw)


class _generic_kernels_only:
    """Within the block every conv launch takes the generic tiles (rgbm_set_tuning ws_min_rows = 2^30): since round 4 the persistent
    kernels start at 1024 GEMM rows in every storage type, so two batch sizes on either side of a threshold sum in different
    orders (1e-6 .. 1e-5 apart) — tests that compare batches of different sizes BIT FOR BIT (or to 1e-6) pin the selection."""
    def __enter__(self):
        from rgbmanip_amd import _lib
        _lib.check(_lib.load().rgbm_set_tuning(b"ws_min_rows", 1 << 30))

    def __exit__(self, *a):
        from rgbmanip_amd import _lib
        _lib.check(_lib.load().rgbm_set_tuning(b"ws_min_rows", 0))


def _net_sd(sd, dtype, **kw):
    from rgbmanip_amd.adapose import AdaPoseNet
    return AdaPoseNet(sd, dtype=dtype, **kw)


def _run(net, inp, **kw):
    out = net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"], **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("cost_impl", [2, 1, 0])
def test_fp32_intermediates_vs_oracle(inputs, oracle_taps, cost_impl):
    """Bisecting aid: PSPNet stages and cost-volume stages of view 1 against the oracle (fp32), for each
    cost-volume implementation (2 = halo-tiled + fused warp [default], 1 = halo-tiled, 0 = generic igemm)."""
    _, taps = oracle_taps
    net = _net("fp32", cost_impl=cost_impl)
    B, V = 2, 4
    _run(net, inputs, stop_after=1)

    def nhwc(name, C, H):
        x = net.fetch(B, name, V * H * H * C).view(V, H, H, C)[:B]      # views 0..B-1 = view 1 of each pose
        return x.permute(0, 3, 1, 2).cpu().numpy()
    errs = {}
    errs["conv1"] = _rel(nhwc("conv1", 64, 112), taps["v1_conv1"].numpy())
    errs["layer4"] = _rel(nhwc("layer4", 512, 28), taps["v1_layer4"].numpy())
    errs["cat"] = _rel(nhwc("cat", 1024, 28), taps["v1_psp"].numpy())
    errs["u1"] = _rel(nhwc("u1", 256, 56), taps["v1_up_1"].numpy())
    errs["u2"] = _rel(nhwc("u2", 64, 112), taps["v1_up_2"].numpy())
    errs["u3"] = _rel(nhwc("u3", 64, 224), taps["v1_up_3"].numpy())
    errs["feat"] = _rel(nhwc("feat", 32, 224), taps["feat1"].numpy())
    print("fp32 PSPNet stage errors:", errs)
    for k, e in errs.items():
        assert e < RTOL_FP32, (k, errs)
    # cost volume of the first chunk (all 4 views fit one chunk)
    _run(net, inputs, stop_after=2)

    def ndhwc(name, C, D, H):
        x = net.fetch(B, name, V * D * H * H * C).view(V, D, H, H, C)[:B]
        return x.permute(0, 4, 1, 2, 3).cpu().numpy()
    errs = {}
    if cost_impl != 2:
        errs["vol"] = _rel(ndhwc("vol", 32, 24, 224), taps["fused1"].numpy())
    errs["c0"] = _rel(ndhwc("c0", 8, 24, 224), taps["v1_c0"].numpy())
    errs["c2"] = _rel(ndhwc("c2", 16, 12, 112), taps["v1_c2"].numpy())
    errs["c4"] = _rel(ndhwc("c4", 32, 6, 56), taps["v1_c4"].numpy())
    errs["c6"] = _rel(ndhwc("c6", 64, 3, 28), taps["v1_c6"].numpy())
    errs["u7"] = _rel(ndhwc("u7", 32, 6, 56), taps["v1_u7"].numpy())
    errs["u9"] = _rel(ndhwc("u9", 16, 12, 112), taps["v1_u9"].numpy())
    if cost_impl == 0:
        errs["u11"] = _rel(ndhwc("u11", 8, 24, 224), taps["v1_u11"].numpy())
    else:
        # the halo-tiled conv11 writes its 8 sub-pixel classes as 8 dense volumes [cls][view][12][112][112][8]
        cm = net.fetch(B, "u11", V * 24 * 224 * 224 * 8).view(2, 2, 2, V, 12, 112, 112, 8)[:, :, :, :B]
        full = cm.permute(3, 7, 4, 0, 5, 1, 6, 2).reshape(B, 8, 24, 224, 224)      # [v, c, (qd,pd), (qh,ph), (qw,pw)]
        errs["u11"] = _rel(full.cpu().numpy(), taps["v1_u11"].numpy())
    prob = net.fetch(B, "prob", V * 1024 * 24).view(V, 1024, 24)[:B].permute(0, 2, 1).cpu().numpy()
    errs["prob"] = _rel(prob, taps["v1_prob"].numpy())
    print("fp32 cost-volume stage errors:", errs)
    for k, e in errs.items():
        assert e < RTOL_FP32, (k, errs)


def test_fp32_matches_reference_golden(inputs, golden_dir):
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    out = _run(_net("fp32"), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print("fp32 vs reference golden:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
        assert errs[k] < RTOL_FP32, (k, errs)


def test_bf16x3_matches_reference_golden(inputs, golden_dir):
    """The split-pair mode (RGBM_BF16X3: values as bf16 hi + lo, products hi*hi + lo*hi + hi*lo on the bf16 matrix pipe, fp32
    accumulate) must meet north_star's fp32 gate — it is the mode that is benched as "passes 1e-4".  CPU emulation of the
    arithmetic through the oracle (tools/split_emulation.py) predicts a worst output error of 2.2e-5."""
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    out = _run(_net("bf16x3"), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print("bf16x3 vs reference golden:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
        assert errs[k] < RTOL_FP32, (k, errs)


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_per_sample_batchnorm_matches_reference_train_mode_golden(inputs, golden_dir, dtype):
    """norm_mode = 1 (per-sample BatchNorm3d statistics: the reference as shipped, train mode at batch 1 with Dropout2d as identity)
    against the reference's own outputs in that mode (tests/golden/adapose_b2_trainbn.npz), inside the fp32 gate — batched here
    (B = 2), one pose per call there: every view is normalised with its own volume's statistics either way."""
    g = np.load(os.path.join(golden_dir, "adapose_b2_trainbn.npz"))
    out = _run(_net(dtype, norm_mode=1), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print(f"{dtype} norm_mode=1 vs reference train-mode golden:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
        assert errs[k] < RTOL_FP32, (k, errs)


def test_bf16x3_intermediates_vs_oracle(inputs, oracle_taps):
    """Stage taps of the split-pair mode against the oracle: every stage inside the fp32 gate."""
    _, taps = oracle_taps
    # the `conv1` tap exists only on the unfused stem path (the fused one: test_stem_*), the whole `u9` volume only with the dense decoder
    net = _net("bf16x3", options={"stem": 0, "sparse_dec": 0})
    B, V = 2, 4
    _run(net, inputs, stop_after=1)

    def nhwc(name, C, H):
        return net.fetch(B, name, V * H * H * C).view(V, H, H, C)[:B].permute(0, 3, 1, 2).cpu().numpy()
    errs = {"conv1": _rel(nhwc("conv1", 64, 112), taps["v1_conv1"].numpy()), "layer4": _rel(nhwc("layer4", 512, 28), taps["v1_layer4"].numpy()),
            "cat": _rel(nhwc("cat", 1024, 28), taps["v1_psp"].numpy()), "u1": _rel(nhwc("u1", 256, 56), taps["v1_up_1"].numpy()),
            "u2": _rel(nhwc("u2", 64, 112), taps["v1_up_2"].numpy()), "feat": _rel(nhwc("feat", 32, 224), taps["feat1"].numpy())}
    _run(net, inputs, stop_after=2)

    def ndhwc(name, C, D, H):
        return net.fetch(B, name, V * D * H * H * C).view(V, D, H, H, C)[:B].permute(0, 4, 1, 2, 3).cpu().numpy()
    errs["c0"] = _rel(ndhwc("c0", 8, 24, 224), taps["v1_c0"].numpy())
    errs["c4"] = _rel(ndhwc("c4", 32, 6, 56), taps["v1_c4"].numpy())
    errs["u9"] = _rel(ndhwc("u9", 16, 12, 112), taps["v1_u9"].numpy())
    prob = net.fetch(B, "prob", V * 1024 * 24).view(V, 1024, 24)[:B].permute(0, 2, 1).cpu().numpy()
    errs["prob"] = _rel(prob, taps["v1_prob"].numpy())
    print("bf16x3 stage errors:", errs)
    for k, e in errs.items():
        assert e < RTOL_FP32, (k, errs)


@pytest.mark.parametrize("dtype", ["bf16x3", "fp32"])
def test_pf96_tap_after_a_full_forward(inputs, oracle_taps, dtype):
    """`pf96` ([view][point][32 depth-fused feature channels | 64 nocs_pts_mlp channels]) fetched AFTER a whole forward: a split-pair
    net has converted the buffer in place to hi / lo pairs for the pose MLP by then, and the tap has to undo that (round-3 advisor:
    it used to reinterpret the pair bits as fp32)."""
    _, taps = oracle_taps
    net = _net(dtype)
    _run(net, inputs)
    pf = net.fetch(2, "pf96", 4 * 1024 * 96).view(4, 1024, 96)[:2].cpu().numpy()
    assert np.isfinite(pf).all()
    e = _rel(np.transpose(pf[:, :, :32], (0, 2, 1)), taps["v1_fg"].numpy())
    print(dtype, "pf96[:32] vs oracle fg", e)
    assert e < RTOL_FP32, e


@pytest.mark.parametrize("dtype", ["bf16", "bf16x3"])
def test_3d_taps_are_refused_under_sparse_cost_regularisation(inputs, dtype):
    """With the default (sparse_dec = 2) c0 .. u9 are written only inside the chosen pixels' dependency cones: fetching them must
    fail loudly instead of returning partly stale tensors; with sparse_dec = 0 the same call works."""
    from rgbmanip_amd import _lib
    net = _net(dtype)
    _run(net, inputs, stop_after=2)
    for name in ("c0", "c2", "c4", "c6", "u7", "u9"):
        with pytest.raises(_lib.RgbmError, match="sparse_dec"):
            net.fetch(2, name, 4 * 24 * 224 * 224 * 8)
    assert net.fetch(2, "prob", 4 * 1024 * 24).numel() == 4 * 1024 * 24
    dense = _net(dtype, options={"sparse_dec": 0})
    _run(dense, inputs, stop_after=2)
    assert dense.fetch(2, "c0", 4 * 24 * 224 * 224 * 8).numel() == 4 * 24 * 224 * 224 * 8


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16", "fp32"])
def test_graph_replay_is_bit_identical_to_eager(dtype):
    """`AdaPoseNet(graph=True)` (rgbm_adapose_forward_graph: the forward of a batch size captured once into a hipGraph, static
    buffers, one hipGraphLaunch per call) against the eager launch sequence at B = 1, 2 and 8, alternating batch sizes and inputs:
    every output bit-identical, the capture happens once per batch size, and a changed option drops the stale graph."""
    eager = _net(dtype)
    graph = _net(dtype, graph=True)
    nodes = {}
    for rnd, (B, seed) in enumerate([(1, 3), (8, 4), (1, 5), (2, 6), (8, 7), (2, 6)]):
        inp = synth.adapose_inputs(B, seed=seed)
        a = _run(eager, inp)
        b = _run(graph, inp)
        assert graph.last_graph_nodes > 50, graph.last_graph_nodes      # a whole forward: ~150 kernel launches + copies
        nodes.setdefault(B, graph.last_graph_nodes)
        assert nodes[B] == graph.last_graph_nodes
        for k in OUT_KEYS:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{dtype} round {rnd} B={B} {k}")
    print(dtype, "graph nodes per batch size:", nodes)
    # an option change invalidates the captured graphs (the dense decoder launches other kernels)
    from rgbmanip_amd import _lib
    if dtype != "fp32":
        _lib.check(graph.lib.rgbm_adapose_set_option(graph._h, b"sparse_dec", 0))
        _lib.check(eager.lib.rgbm_adapose_set_option(eager._h, b"sparse_dec", 0))
        inp = synth.adapose_inputs(2, seed=9)
        a, b = _run(eager, inp), _run(graph, inp)
        for k in OUT_KEYS:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{dtype} after option change {k}")


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16", "fp32"])
def test_view1_only_heads_equal_the_full_forward(dtype):
    """Option view2_heads = 0 (what `AdaPoseEstimator_v5` runs: the box is built from view1_nocs / view1_depth / view1_r alone,
    interface_v5.py:318-374): the probability volume, point heads and pose regression of the view-2 crops are skipped.  The five
    view-1 outputs must equal the full forward's BIT FOR BIT (B = 3: a chunked cost volume with a ragged last chunk as well), the
    five view-2 outputs must be NaN (not stale numbers)."""
    inp = synth.adapose_inputs(3, seed=12)
    with _generic_kernels_only():          # 3 instead of 6 views in the heads: keep both runs on the same kernels
        for kw in ({}, {"max_chunk_views": 4}):
            full = _run(_net(dtype, **kw), inp)
            v1 = _run(_net(dtype, options={"view2_heads": 0}, **kw), inp)
            for k in OUT_KEYS:
                if k.startswith("view1"):
                    np.testing.assert_array_equal(v1[k], full[k], err_msg=f"{dtype} {kw} {k}")
                else:
                    assert np.isnan(v1[k]).all(), (dtype, kw, k)
                    assert np.isfinite(full[k]).all()
    # default kernel selection: half the head views may cross a dispatch threshold — same function, another summation order
    full, v1 = _run(_net(dtype), inp), _run(_net(dtype, options={"view2_heads": 0}), inp)
    for k in OUT_KEYS:
        if k.startswith("view1"):
            assert _rel(v1[k], full[k]) < (2e-3 if dtype == "bf16" else 1e-5), (dtype, k, _rel(v1[k], full[k]))


def test_estimator_skips_view2_heads_unless_the_tail_needs_them():
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    sd = synth.adapose_state_dict(seed=0, prefix="module.")
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False)
    assert AdaPoseEstimator_v5(None, cfg, None, state_dict=sd, dtype="bf16").view2_heads is False
    assert AdaPoseEstimator_v5(None, dict(cfg, direct_regression=False, use_depth=True), None, state_dict=sd, dtype="bf16").view2_heads is False
    assert AdaPoseEstimator_v5(None, dict(cfg, direct_regression=False, use_depth=False), None, state_dict=sd, dtype="bf16").view2_heads is True
    assert AdaPoseEstimator_v5(None, dict(cfg, hip_view2_heads=True), None, state_dict=sd, dtype="bf16").view2_heads is True
    # hip_options: rgbm_adapose_set_option keys through the cfg (round 5: sweep_f16 = 0 keeps a bf16 net's feature map in bf16); an unknown key fails loudly
    est = AdaPoseEstimator_v5(None, dict(cfg, hip_options={"sweep_f16": 0}), None, state_dict=sd, dtype="bf16")
    assert est.estimator.options["sweep_f16"] == 0 and est.estimator.options["view2_heads"] == 0
    with pytest.raises(Exception):
        AdaPoseEstimator_v5(None, dict(cfg, hip_options={"no_such_option": 1}), None, state_dict=sd, dtype="bf16")
    # a SHARED net keeps its own options: a cfg that asks for something else is refused, one that names what the net has is accepted
    assert AdaPoseEstimator_v5(None, dict(cfg, hip_options={"sweep_f16": 0}), None, dtype="bf16", net=est.estimator).estimator is est.estimator
    with pytest.raises(ValueError, match="shared net"):
        AdaPoseEstimator_v5(None, dict(cfg, hip_options={"sweep_f16": 1}), None, dtype="bf16", net=est.estimator)


def test_f16_feature_map_only_for_weights_that_fit_f16():
    """sweep_f16 (bf16 nets: f16 feature map, f16 conv0 / `final` weights) is the default, but only for checkpoints whose folded conv0 and
    `final` weights are representable in IEEE f16 (round-5 advice): scaled beyond +-65504 — or down into the f16 subnormals — the network
    falls back to the all-bf16 form by itself instead of saturating silently."""
    inp = synth.adapose_inputs(1, seed=3)

    def feat_dtype(sd):
        net = _net_sd(sd, "bf16")
        _run(net, inp, stop_after=1)
        f = net.fetch(1, "feat", 2 * 224 * 224 * 32)            # fp32 copies of the stored values
        assert torch.isfinite(f).all() and float(f.abs().max()) > 0
        # a bf16 value has 8 significand bits (low 16 bits of its fp32 form are zero), an f16 value up to 11
        return torch.float16 if bool((f.view(torch.int32) & 0xFFFF).any()) else torch.bfloat16

    sd = synth.adapose_state_dict(seed=0)
    assert feat_dtype(sd) == torch.float16
    big = dict(sd)
    big["cost_regularization.conv0.conv.weight"] = sd["cost_regularization.conv0.conv.weight"] * 1e6
    assert feat_dtype(big) == torch.bfloat16
    tiny = dict(sd)
    tiny["img_extractor.final.weight"] = sd["img_extractor.final.weight"] * 1e-6
    assert feat_dtype(tiny) == torch.bfloat16


def test_fp32_batch_invariance_and_chunking():
    """B=3 with a chunked cost volume (4 views per chunk, ragged last chunk) equals per-pose results."""
    inp3 = synth.adapose_inputs(3, seed=5)
    out3 = _run(_net("fp32", max_chunk_views=4), inp3)
    net1 = _net("fp32")
    for b in range(3):
        one = {k: v[b:b + 1] for k, v in inp3.items()}
        o1 = _run(net1, one)
        for k in OUT_KEYS:
            assert _rel(out3[k][b:b + 1], o1[k]) < 1e-5, (b, k)


def test_bf16_sweep_f16_off_close_to_golden(inputs, golden_dir):
    """option sweep_f16 = 0: the all-bf16 plane sweep of rounds 1-4 (bf16 feature map, fp32 blend, one workgroup per tile) stays inside the bf16 gates."""
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    out = _run(_net("bf16", options={"sweep_f16": 0}), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print("bf16, sweep_f16 = 0, vs reference golden:", errs)
    gate = {"nocs": 2.7e-2, "depth": 1.0e-2, "r": 6.0e-3, "t": 3.5e-3, "s": 1.2e-3}
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all() and errs[k] < gate[k.split("_")[1]], (k, errs)


@pytest.mark.parametrize("cost_impl", [3, 2, 0])
def test_bf16_close_to_golden(inputs, golden_dir, cost_impl):
    """bf16 storage / fp32 accumulate: the throughput mode.  The bound is what 8-bit mantissas allow through a
    ~60-layer un-normalised network; the measured errors are printed and recorded in DESIGN.md."""
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    out = _run(_net("bf16", cost_impl=cost_impl), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print(f"bf16 (cost_impl={cost_impl}) vs reference golden:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
    # gates at 2x the measured errors (nocs 1.34e-2, depth 4.9e-3, R 2.97e-3, t 1.72e-3, s 5.7e-4): a regression of the
    # bf16 path by a factor of two fails here
    gate = {"nocs": 2.7e-2, "depth": 1.0e-2, "r": 6.0e-3, "t": 3.5e-3, "s": 1.2e-3}
    for k in OUT_KEYS:
        assert errs[k] < gate[k.split("_")[1]], (k, errs)


@pytest.mark.parametrize("cost_impl", [3, 0])
def test_fp16_close_to_golden(inputs, golden_dir, cost_impl):
    """fp16 storage (saturating stores) / fp32 accumulate through the generic kernels (BASELINE configs[4] names fp16): 11-bit
    mantissas, so it must sit well inside the bf16 bounds and stay finite (layer4 activations reach 650)."""
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    out = _run(_net("fp16", cost_impl=cost_impl), inputs)
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    print(f"fp16 (cost_impl={cost_impl}) vs reference golden:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
    # gates at 2x the measured errors (nocs 1.2e-3, depth 8.3e-4, R 5.8e-5, t 1.5e-4, s 3.5e-5)
    gate = {"nocs": 2.4e-3, "depth": 1.7e-3, "r": 1.2e-4, "t": 3.0e-4, "s": 7.0e-5}
    for k in OUT_KEYS:
        assert errs[k] < gate[k.split("_")[1]], (k, errs)


@pytest.mark.parametrize("blend", ["f16_features", "dot2", "bf16_f32", "bf16_scalar_f32"])
def test_bf16_sweep_conv0_vs_tile_conv0(inputs, oracle_taps, blend):
    """cost_impl 3 (depth-sweeping conv0, conv0_sweep.hip) vs 2 (halo-tile conv0) in a bf16 net.
    f16_features (the default since round 5, option sweep_f16 = 1): `final` writes the feature map as f16 and the sweep blends it with
    packed f16 FMAs, while the halo-tile conv0 of the comparison reads the bf16 feature map — the two differ by the features' rounding
    (8 against 11 bits), and the sweep must be the one closer to the fp32 oracle.
    The other variants (sweep_f16 = 0): same bf16 features and conv weights as the tile kernel, fp32 accumulation in a different order.
    With the fp32 blends (debug flags 0 / 2097152) c0 may differ by one bf16 rounding at most; the dot2 blend (v_perm + v_dot2_f32_bf16,
    debug flag 4194304: the round-4 default) also rounds the four bilinear weights of a voxel to bf16 — measured 1.5 bf16 steps at most,
    the mean difference 1.9e-3; every variant stays close to the oracle."""
    from rgbmanip_amd import _lib
    _, taps = oracle_taps
    c0 = {}
    _lib.check(_lib.load().rgbm_debug_flags({"dot2": 1 << 22, "bf16_f32": 0, "bf16_scalar_f32": 1 << 21, "f16_features": 0}[blend]))
    try:
        for ci in (2, 3):
            # the whole c0 volume is compared: no tile skipping
            net = _net("bf16", cost_impl=ci, options={"sparse_dec": 0, "sweep_f16": int(blend == "f16_features")})
            _run(net, inputs, stop_after=2)
            c0[ci] = net.fetch(2, "c0", 4 * 24 * 224 * 224 * 8).view(4, 24, 224, 224, 8).cpu().numpy()
    finally:
        _lib.check(_lib.load().rgbm_debug_flags(0))
    scale = np.abs(c0[2]).max()
    dmax, dmean = np.abs(c0[3] - c0[2]).max() / scale, np.abs(c0[3] - c0[2]).mean() / np.abs(c0[2]).mean()
    print(blend, "sweep vs tile conv0: max", dmax, "mean", dmean)
    assert dmax < {"dot2": 1.3e-2, "f16_features": 2.0e-2}.get(blend, 8e-3)
    assert dmean < {"dot2": 2.5e-3, "f16_features": 4.5e-3}.get(blend, 1e-3)      # measured: dot2 1.9e-3, f16 features against bf16 ones 3.1e-3
    ref = taps["v1_c0"].numpy()                                   # [2,8,24,224,224]
    rel = {}
    for ci in (2, 3):
        got = np.transpose(c0[ci][:2], (0, 4, 1, 2, 3))
        rel[ci] = _rel(got, ref)
        assert rel[ci] < 2e-2, ci
    print(blend, "c0 vs the fp32 oracle: tile", rel[2], "sweep", rel[3])
    if blend == "f16_features":
        assert rel[3] < rel[2]      # more mantissa in the features: closer to fp32 than the all-bf16 conv0


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_persistent_sweep_is_bit_identical_to_the_one_tile_sweep(inputs, dtype):
    """conv0_sweep_persistent_kernel (one workgroup per CU walks its tiles, the two roles stream across tile boundaries; the default of bf16
    nets, and since round 6 of fp16 nets with their fp32-accumulating v_fma_mix blend) against the one-workgroup-per-tile kernel with the
    same arithmetic (debug flag 268435456): the whole c0 volume, and the network outputs with the sparse cost regularisation's tile list,
    bit for bit."""
    from rgbmanip_amd import _lib
    lib = _lib.load()

    def run(flag, sparse_dec):
        _lib.check(lib.rgbm_debug_flags(flag))
        try:
            net = _net(dtype, options={"sparse_dec": sparse_dec})
            if sparse_dec == 0:
                _run(net, inputs, stop_after=2)
                return net.fetch(2, "c0", 4 * 24 * 224 * 224 * 8).cpu().numpy()
            return _run(net, inputs)
        finally:
            _lib.check(lib.rgbm_debug_flags(0))
    np.testing.assert_array_equal(run(0, 0), run(1 << 28, 0))
    a, b = run(0, 2), run(1 << 28, 2)
    for k in OUT_KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "bf16x3"])
def test_prob_sparse_kernels_agree(dtype):
    """The sparse tail's round-6 point kernel (prob_sparse2_kernel: neighbourhoods staged once in LDS, parity-specialised code, both depth
    parities on the 16 MFMA rows) against the round-3..5 kernel (debug flag 536870912) on the same network: same fp32 arithmetic per voxel
    in another summation order, so prob / depth agree to fp32 rounding, and everything downstream with them.  Inputs with chosen pixels on
    the crop's border rows / columns (zero padding of every axis) and wrap-padded duplicates."""
    from rgbmanip_amd import _lib
    lib = _lib.load()
    inp = {k: v.copy() for k, v in synth.adapose_inputs(3, seed=4).items()}
    g = np.random.default_rng(1)
    border = np.concatenate([np.arange(224), 223 * 224 + np.arange(224), np.arange(224) * 224, np.arange(224) * 224 + 223])
    for k in ("choose1", "choose2"):
        inp[k][0] = np.sort(g.choice(border, 1024))                          # corners, edges, duplicates
        inp[k][1] = np.sort(np.concatenate([g.permutation(224 * 224)[:1000], g.permutation(224 * 224)[:24]]))
    outs = {}
    for flag in (0, 1 << 29):
        _lib.check(lib.rgbm_debug_flags(flag))
        try:
            net = _net(dtype)
            outs[flag] = _run(net, inp)
            prob = net.fetch(3, "prob", 6 * 1024 * 24).view(6, 1024, 24).cpu().numpy()
            assert np.isfinite(prob).all() and np.allclose(prob.sum(-1), 1.0, atol=1e-5)
            outs[flag]["prob"] = prob
        finally:
            _lib.check(lib.rgbm_debug_flags(0))
    a, b = outs[0], outs[1 << 29]
    assert np.abs(a["prob"] - b["prob"]).max() < 2e-6
    for k in OUT_KEYS:
        assert _rel(a[k], b[k]) < 1e-5, k


@pytest.mark.parametrize("dtype", ["bf16", "bf16x3"])
def test_halo_tile_conv0_runtime_switch(inputs, golden_dir, dtype):
    """debug flag 4096 swaps the depth-sweeping conv0 kernels (hand-counted inline-asm gathers) for the halo-tile conv0 at run
    time, in every storage type: the way out if a compiler update ever disturbs the sweep kernels.  Same gates as the default path."""
    from rgbmanip_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "adapose_b2.npz"))
    _lib.check(lib.rgbm_debug_flags(4096))
    try:
        out = _run(_net(dtype), inputs)
    finally:
        _lib.check(lib.rgbm_debug_flags(0))
    errs = {k: _rel(out[k], g[k]) for k in OUT_KEYS}
    gate = {"nocs": 2.7e-2, "depth": 1.0e-2, "r": 6.0e-3, "t": 3.5e-3, "s": 1.2e-3}
    for k in OUT_KEYS:
        assert np.isfinite(out[k]).all(), k
        assert errs[k] < (RTOL_FP32 if dtype == "bf16x3" else gate[k.split("_")[1]]), (k, errs)


def test_bf16_fused_final_conv(inputs, oracle_taps):
    """up_3 + final in one launch (the store waves of conv_igemm_ws64_kernel multiply the staged bf16 tile by the 1x1
    weights) against the two-launch path: same bf16 operands, same fp32 accumulation, so the feature maps must agree to
    one bf16 rounding; and both must stay close to the oracle's feature map."""
    _, taps = oracle_taps
    feats = {}
    for ff in (0, 1):
        net = _net("bf16", options={"fuse_final": ff})
        _run(net, inputs, stop_after=1)
        feats[ff] = net.fetch(2, "feat", 4 * 224 * 224 * 32).view(4, 224, 224, 32).cpu().numpy()
    scale = np.abs(feats[0]).max()
    assert np.isfinite(feats[1]).all()
    assert np.abs(feats[1] - feats[0]).max() / scale < 8e-3
    assert np.abs(feats[1] - feats[0]).mean() / np.abs(feats[0]).mean() < 1e-3
    ref = taps["feat1"].numpy()                                        # [2,32,224,224]
    for ff in (0, 1):
        got = np.transpose(feats[ff][:2], (0, 3, 1, 2))
        assert _rel(got, ref) < 3e-2, ff


def test_bf16_sparse_tail_vs_dense_tail(inputs, oracle_taps):
    """prob_sparse.hip (conv11 + skip + prob conv + softmax + depth only where prob is gathered; u11 kept in fp32) against
    the dense conv11 + gathering prob kernel it replaces (u11 rounded to bf16), same bf16 u9 / c0 / weights; and both
    against the oracle's probability volume samples."""
    _, taps = oracle_taps
    got = {}
    for st in (0, 1):
        net = _net("bf16", cost_impl=3, sparse_tail=st)
        out = _run(net, inputs, stop_after=2)
        prob = net.fetch(2, "prob", 4 * 1024 * 24).view(4, 1024, 24).cpu().numpy()
        got[st] = prob
    assert np.isfinite(got[1]).all()
    np.testing.assert_allclose(got[1].sum(-1), 1.0, rtol=0, atol=1e-5)
    assert np.abs(got[1] - got[0]).max() < 2e-2                      # one bf16 rounding of u11 apart
    ref = np.transpose(taps["v1_prob"].numpy(), (0, 2, 1))           # [2,1024,24]
    e_dense = np.abs(got[0][:2] - ref).max()
    e_sparse = np.abs(got[1][:2] - ref).max()
    print("prob error vs oracle: dense tail", e_dense, "sparse tail", e_sparse)
    assert e_sparse < 3e-2 and e_sparse < e_dense + 5e-3


def test_bf16x3_sparse_tail_vs_dense_tail(inputs, oracle_taps):
    """The split-pair instantiation of prob_sparse.hip (u9 / c0 as bf16 hi + lo chunks, three MFMAs per product, u11 kept in
    fp32) against the dense conv11 + gathering prob kernel on the same tensors, and both against the oracle: all inside the
    fp32 gate."""
    _, taps = oracle_taps
    got = {}
    for st in (0, 1):
        net = _net("bf16x3", sparse_tail=st)
        _run(net, inputs, stop_after=2)
        got[st] = net.fetch(2, "prob", 4 * 1024 * 24).view(4, 1024, 24).cpu().numpy()
    assert np.isfinite(got[1]).all()
    np.testing.assert_allclose(got[1].sum(-1), 1.0, rtol=0, atol=1e-5)
    ref = np.transpose(taps["v1_prob"].numpy(), (0, 2, 1))           # [2,1024,24]
    e_dense, e_sparse = _rel(got[0][:2], ref), _rel(got[1][:2], ref)
    print("bf16x3 prob error vs oracle: dense tail", e_dense, "sparse tail", e_sparse, "sparse vs dense", _rel(got[1], got[0]))
    assert e_sparse < RTOL_FP32 and e_dense < RTOL_FP32 and _rel(got[1], got[0]) < 5e-5


def test_nan_projection_stays_per_sample(inputs):
    """A degenerate pair (singular projection) must poison only its own pose."""
    inp = {k: v.copy() for k, v in inputs.items()}
    inp["P2"][1] = 0.0
    inp["P2"][1, 3, 3] = 1.0
    net = _net("fp32")
    out = _run(net, inp)
    ref = _run(net, inputs)
    assert _rel(out["view1_depth"][0], ref["view1_depth"][0]) < 1e-6
    assert _rel(out["view1_r"][0], ref["view1_r"][0]) < 1e-6


def test_estimate_end_to_end_vs_oracle(inputs, oracle_taps):
    """network + device post-processing -> world bbox, against oracle network + numpy post-processing."""
    from oracle import postproc_ref
    from rgbmanip_amd.adapose import postprocess
    oout, _ = oracle_taps
    net = _net("fp32")
    out = net(inputs["img1"], inputs["choose1"], inputs["img2"], inputs["choose2"], inputs["P1"], inputs["P2"], inputs["depths"])
    bbox, ts, valid = postprocess(out["view1_nocs"], out["view1_depth"], out["view1_r"], inputs["choose1"], inputs["K1"], inputs["E1"])
    torch.cuda.synchronize()
    for b in range(2):
        exp = postproc_ref.bbox_world(oout["view1_nocs"][b].numpy(), oout["view1_depth"][b].numpy(), oout["view1_r"][b].numpy(),
                                      inputs["choose1"][b], inputs["K1"][b], inputs["E1"][b])
        assert _rel(bbox[b].cpu().numpy(), exp) < 1e-3, b


def test_estimator_plugin_estimate_vs_oracle_pipeline():
    """`AdaPoseEstimator_v5.estimate` (crop/resize/sample on the host, batched HIP net + post-processing) against the
    oracle pipeline; an empty mask must yield the +10 default cube without disturbing its neighbours."""
    from oracle import adapose_ref, postproc_ref
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5, DEFAULT_BBOX
    g = np.random.default_rng(3)
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="host")      # the mode that consumes the reference's RNG stream
    sd = synth.adapose_state_dict(seed=0)
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=sd, dtype="fp32")
    n = 3
    yy, xx = np.mgrid[0:480, 0:640]
    K = np.tile(np.array([[439.31, 0, 320.0], [0, 439.31, 240.0], [0, 0, 1.0]])[None], (n, 1, 1))
    base = synth.adapose_inputs(n, seed=9)
    rgb1 = np.clip(0.5 + 0.25 * np.cos(xx / 37.0)[None, :, :, None] + 0.2 * g.random((n, 480, 640, 3)), 0, 1)
    rgb2 = np.clip(0.5 + 0.25 * np.sin(yy / 29.0)[None, :, :, None] + 0.2 * g.random((n, 480, 640, 3)), 0, 1)
    m1 = np.stack([((yy - 240) / 60.0) ** 2 + ((xx - 300 - 10 * i) / 90.0) ** 2 <= 1 for i in range(n)])
    m2 = np.stack([((yy - 250) / 70.0) ** 2 + ((xx - 340 + 10 * i) / 80.0) ** 2 <= 1 for i in range(n)])
    m2[1] = False                                            # sample 1: empty second mask -> default bbox
    E1, E2 = base["E1"], base["E2"]
    est.rng = np.random.default_rng(11)
    out = est.estimate(K, rgb1, m1, E1, rgb2, m2, E2)
    assert out.shape == (n, 8, 3) and np.allclose(out[1], DEFAULT_BBOX)
    # oracle pipeline with the same sampling stream
    rng = np.random.default_rng(11)
    tsd = adapose_ref.to_torch_sd(sd)
    for i in range(n):
        a = postproc_ref.prepare_model_input(rgb1[i], m1[i], K[i], 224, rng=rng)     # same RNG consumption as estimate()
        b = postproc_ref.prepare_model_input(rgb2[i], m2[i], K[i], 224, rng=rng)
        if a[0] is None or b[0] is None:
            continue
        P1, P2 = np.eye(4), np.eye(4)
        P1[:3] = a[3] @ E1[i][:3]
        P2[:3] = b[3] @ E2[i][:3]
        dep = torch.arange(24, dtype=torch.float32)[None] * 0.1 + 0.1
        o = adapose_ref.adapose_forward(tsd, torch.from_numpy(a[0]).float()[None], torch.from_numpy(a[1])[None],
                                        torch.from_numpy(b[0]).float()[None], torch.from_numpy(b[1])[None],
                                        torch.from_numpy(P1).float()[None], torch.from_numpy(P2).float()[None], dep)
        exp = postproc_ref.bbox_world(o["view1_nocs"][0].numpy(), o["view1_depth"][0].numpy(), o["view1_r"][0].numpy(), a[1], a[3], E1[i])
        assert _rel(out[i], exp) < 1e-3, i


def _prepare_case(seed=3):
    """Five 480x640 frames: big ellipse (subset), small blob near the border (wrap-pad, clamped window), empty mask, two
    corner blobs (empty resized mask), thin line (few resized pixels)."""
    rng = np.random.default_rng(seed)
    N, H, W = 5, 480, 640
    rgb = rng.random((N, H, W, 3), dtype=np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.zeros((N, H, W), dtype=np.uint8)
    mask[0] = (((yy - 250) / 120.0) ** 2 + ((xx - 300) / 170.0) ** 2) < 1.0
    mask[1] = (((yy - 12) / 9.0) ** 2 + ((xx - 630) / 7.0) ** 2) < 1.0       # window clamped at the top-right corner
    mask[3][440:480, 600:640] = 1
    mask[3][0:30, 0:25] = 1                      # bbox spans the frame -> centred 440 window misses both blobs -> empty resized mask
    mask[4][200:203, 50:400] = 1
    K = np.tile(np.array([[439.31, 0, 320.0], [0, 439.31, 240.0], [0, 0, 1.0]]), (N, 1, 1))
    K[:, 0, 2] += np.arange(N) * 1.5
    return rgb, mask, K


def test_prepare_inputs_bit_exact_vs_oracle():
    """Device-side prepare_model_input (rgbm_prepare_inputs) against the numpy restatement of
    interface_v5.py:58-170: windows, chosen indices, cropped intrinsics and pts2d exactly; images bit for bit."""
    from oracle import postproc_ref
    from rgbmanip_amd.adapose import prepare_inputs
    rgb, mask, K = _prepare_case()
    seed = 77
    out = prepare_inputs(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(), torch.from_numpy(K).cuda(), 224, 1024,
                         seed, want_pts2d=True)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    n_subset = 0
    for f in range(rgb.shape[0]):
        view, choose, pts2d, Kn = postproc_ref.prepare_model_input(rgb[f], mask[f], K[f], 224, rng=("hash", seed, f))
        if view is None:
            assert got["valid"][f] == 0, f
            assert np.isfinite(got["img"][f]).all()
            continue
        assert got["valid"][f] == 1, f
        assert np.array_equal(got["choose"][f], choose.astype(np.int32)), f
        assert np.array_equal(got["Kcrop"][f], Kn), f
        assert np.array_equal(got["pts2d"][f], pts2d.astype(np.float32)), f
        assert np.array_equal(got["img"][f], view.astype(np.float32)), (f, np.abs(got["img"][f] - view).max())
        n_subset += int(mask[f].sum() > 0 and len(np.unique(choose)) == 1024)
    assert got["valid"].tolist() == [1, 1, 0, 0, 1]         # frame 2: empty mask; frame 3: empty *resized* mask
    assert n_subset >= 1                      # at least one frame exercised the random-subset branch


def test_estimate_device_prepare_matches_host_prepare():
    """The estimator plugin with cfg hip_prepare='device' (frames uploaded once, everything else on the GPU) against the
    host-numpy preparation with the same hash subset: same crops bit for bit, so the same boxes."""
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    rgb, mask, K = _prepare_case(seed=5)
    N = rgb.shape[0]
    rgb2 = np.ascontiguousarray(rgb[:, :, ::-1])          # a second "view": mirrored frames and masks
    mask2 = np.ascontiguousarray(mask[:, :, ::-1])
    inp = synth.adapose_inputs(N, seed=2)
    E1, E2 = inp["E1"].astype(np.float64), inp["E2"].astype(np.float64)
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False)
    sd = synth.adapose_state_dict(seed=0, prefix="module.")
    host = AdaPoseEstimator_v5(None, dict(cfg, hip_prepare="host", hip_prepare_seed=9), None, state_dict=sd, dtype="fp32")
    host.rng = ("hash", 9)
    dev = AdaPoseEstimator_v5(None, dict(cfg, hip_prepare="device", hip_prepare_seed=9), None, state_dict=sd, dtype="fp32")
    with _generic_kernels_only():      # the host path batches the 3 valid poses, the device path all 5, the pipeline 2 at a time
        _estimate_paths_agree(host, dev, cfg, sd, K, rgb, mask, E1, rgb2, mask2, E2, N)


def _estimate_paths_agree(host, dev, cfg, sd, K, rgb, mask, E1, rgb2, mask2, E2, N):
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    b_host = host.estimate(K, rgb, mask, E1, rgb2, mask2, E2)
    b_dev = dev.estimate(K, rgb, mask, E1, rgb2, mask2, E2)
    assert b_host.shape == (N, 8, 3) and b_dev.shape == (N, 8, 3)
    assert np.array_equal(b_host[2], b_dev[2])            # empty mask -> default bbox on both paths
    assert np.allclose(b_host[2], np.asarray([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]]) + 10.0)
    np.testing.assert_allclose(b_dev, b_host, rtol=1e-6, atol=1e-7)
    # the upload path of estimate(): float64 frames (what rl_pose.py:210-218 hands over) and bool masks through pinned chunks — several
    # chunks per call when the staging buffers are small — give the float32 frames' boxes bit for bit; uint8 frames are scaled by 1/255
    # on the device exactly as the host path scales them
    dev._CHUNK_BYTES = 3 * 480 * 640 * 3 * 4 // 2          # one frame per chunk
    b64 = dev.estimate(K, rgb.astype(np.float64), mask.astype(bool), E1, rgb2.astype(np.float64), mask2.astype(bool), E2)
    np.testing.assert_array_equal(b64, b_dev)
    # the same call as a three-stage pipeline over chunks of two poses (host staging | copy engine | kernels; hip_upload_chunk): every
    # pose draws the pixel subset it draws in the unchunked call (rgbm_prepare_inputs_ex: hash offset), so the boxes agree
    pipe = AdaPoseEstimator_v5(None, dict(cfg, hip_prepare="device", hip_prepare_seed=9, hip_upload_chunk=2), None, state_dict=sd, dtype="fp32")
    assert N > 4
    b_pipe = pipe.estimate(K, rgb.astype(np.float64), mask.astype(np.float64), E1, rgb2.astype(np.float64), mask2.astype(bool), E2)
    # BIT FOR BIT (round-5 advice): the caller pinned the tile selection (_generic_kernels_only), so a chunk of two poses and the batch
    # of five sum in the same order — what is left to differ is chunk-boundary indexing, and an error there of any size must show
    np.testing.assert_array_equal(b_pipe, b_dev)
    b_pipe2 = pipe.estimate(K, rgb, mask, E1, rgb2, mask2, E2)          # other dtypes through the same estimator: staging buffers are rebuilt
    np.testing.assert_array_equal(b_pipe2, b_dev)
    o = np.arange(N)[::-1].copy()                                       # the frame that takes the random-subset branch now sits in the LAST chunk
    r = lambda x: np.ascontiguousarray(x[o])                            # noqa: E731
    np.testing.assert_array_equal(pipe.estimate(r(K), r(rgb), r(mask), r(E1), r(rgb2), r(mask2), r(E2)),
                                  dev.estimate(r(K), r(rgb), r(mask), r(E1), r(rgb2), r(mask2), r(E2)))
    u1, u2 = (np.clip(np.rint(x * 255.0), 0, 255).astype(np.uint8) for x in (rgb, rgb2))
    b_u8_host = host.estimate(K, u1, mask, E1, u2, mask2, E2)
    b_u8_dev = dev.estimate(K, u1, mask.astype(np.float32), E1, u2, mask2.astype(np.float32), E2)
    np.testing.assert_allclose(b_u8_dev, b_u8_host, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pipe.estimate(K, u1, mask, E1, u2, mask2, E2), b_u8_host, rtol=1e-6, atol=1e-7)


def test_device_control_queue_matches_reference_golden(golden_dir):
    """rgbmanip_amd.control_interface.ControlInterface (device-resident queues, rgbm_mask_extent, index-arithmetic view
    selection) against the reference ControlInterface's own outputs (tests/golden/control.npz)."""
    from test_oracle_golden import _control_fake_estimator, drive_control_queue
    from rgbmanip_amd.control_interface import ControlInterface
    g = np.load(os.path.join(golden_dir, "control.npz"))
    for task in ("cabinet", "mugs"):
        est = _control_fake_estimator(task)
        ci = ControlInterface.queue_only(3, est, 5)
        obs, states, boxes = drive_control_queue(ci, est, lambda: ci.get_observation().cpu().numpy(), lambda: ci.get_state().cpu().numpy())
        np.testing.assert_array_equal(obs, g[task + "_obs"])
        np.testing.assert_array_equal(states, g[task + "_state"])
        np.testing.assert_array_equal(boxes, g[task + "_pred"])
        for key in ("id1", "id2", "m1", "m2", "K", "E1", "E2"):
            np.testing.assert_array_equal(np.stack([c[key] for c in est.calls]), g[task + "_" + key], err_msg=key)
        np.testing.assert_array_equal(ci.available.cpu().numpy(), g[task + "_available"])
        np.testing.assert_array_equal(ci.available_num.cpu().numpy(), g[task + "_available_num"])
        np.testing.assert_array_equal(ci.bbox_queue.cpu().numpy(), g[task + "_bbox_queue"])


def test_device_control_queue_with_hip_estimator():
    """The queue feeding the real estimator without leaving the GPU (`estimate_device`) gives the boxes the numpy-facing
    `estimate` gives for the same two selected views."""
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    N = 3
    cfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=1)
    est = AdaPoseEstimator_v5(None, cfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="fp32")
    ci = ControlInterface.queue_only(N, est, 5)
    for t in range(3):
        img, pose, gt = synth.control_view(N, t, seed=6)
        ci.add_view(img, pose)
        ci.accumulate_steps += 1
    got = ci.get_estimation().cpu().numpy()
    idx, has = ci.select_views()
    sel = lambda q, s: ci._gather(q, idx[s], has[s]).cpu().numpy()
    ref = est.estimate(sel(ci.intrinsic_queue, 0), sel(ci.image_queue, 0), sel(ci.mask_queue, 0), sel(ci.extrinsic_queue, 0),
                       sel(ci.image_queue, 1), sel(ci.mask_queue, 1), sel(ci.extrinsic_queue, 1))
    assert got.shape == (N, 8, 3)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "bf16x3", "fp32"])
def test_psp_stage_paths_agree(dtype):
    """PSP stage (pspnet.py:76-94).  Round 6: the concat (backbone channels + the four resized stages) is ONE launch at any batch size,
    and up to 32 views the pooling and the four 512 -> 128 convs are one launch on the vector pipe.  Against the rounds 1-5 sequence
    (debug flag 1024: copy, pooling, four GEMM launches, four resizes): 34 views — `cat` bit for bit (same pooling, same GEMMs, the
    resize arithmetic is the same expression); 4 views — channels 0..511 bit for bit, the stage channels to the storage type's rounding
    (fp32 FMAs + a lane butterfly instead of the MFMA's sums over the same rounded operands)."""
    from rgbmanip_amd import _lib
    lib = _lib.load()
    for B in (2, 17):
        inp = synth.adapose_inputs(B, seed=6)
        cats = {}
        for flag in (0, 1024):
            _lib.check(lib.rgbm_debug_flags(flag))
            try:
                net = _net(dtype)
                _run(net, inp, stop_after=1)      # (the cost-volume phase reuses the PSPNet buffers: taps are read behind a forward that stops there)
                cats[flag] = net.fetch(B, "cat", 2 * B * 28 * 28 * 1024).view(2 * B, 28, 28, 1024).cpu().numpy()
            finally:
                _lib.check(lib.rgbm_debug_flags(0))
        a, b = cats[0], cats[1024]
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a[..., :512], b[..., :512])
        if B == 17:
            np.testing.assert_array_equal(a, b)
        else:
            tol = {"bf16": 2.0 ** -7, "fp16": 2.0 ** -10, "bf16x3": 2e-5, "fp32": 2e-6}[dtype]
            scale = np.abs(b[..., 512:]).max()
            assert scale > 0 and np.abs(a[..., 512:] - b[..., 512:]).max() <= tol * scale, (dtype, np.abs(a[..., 512:] - b[..., 512:]).max(), scale)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "bf16x3", "fp32"])
def test_point_mlp_matches_the_per_layer_launches(dtype):
    """Per-point NOCS branch (network_v5.py:432-444): point_mlp_kernel (gather + six fp32 layers in one launch, weights and activations in
    LDS) against the rounds 1-5 sequence (debug flag 2048: gather + six implicit-GEMM launches of the same fp32 layers).  Same fp32
    products; the sums may associate differently: nocs to fp32 rounding, and what is computed from the branch's 64 output channels
    (r / t / s through the pose MLP, which bf16 / fp16 nets run in fp16 storage) to that storage's rounding.  Depth does not depend on it."""
    from rgbmanip_amd import _lib
    lib = _lib.load()
    inp = synth.adapose_inputs(3, seed=8)
    outs = {}
    for flag in (0, 2048):
        _lib.check(lib.rgbm_debug_flags(flag))
        try:
            outs[flag] = _run(_net(dtype), inp)
        finally:
            _lib.check(lib.rgbm_debug_flags(0))
    a, b = outs[0], outs[2048]
    errs = {k: _rel(a[k], b[k]) for k in OUT_KEYS}
    print(dtype, "fused point MLP vs per-layer launches:", errs)
    for k in OUT_KEYS:
        assert np.isfinite(a[k]).all(), k
        kind = k.split("_")[1]
        if kind == "depth":
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        elif kind == "nocs":
            assert errs[k] < 2e-6, (k, errs)
        else:
            assert errs[k] < (1e-5 if dtype in ("fp32", "bf16x3") else 2e-3), (k, errs)


def test_bf16_batch_invariance_across_kernel_selection():
    """bf16, B = 9 in two cost-volume chunks (max 10 views) against the same poses run one by one: the batched run takes the
    persistent kernels for more layers (their M >= 65536 rule) and a ragged last chunk; results may differ only by fp32
    accumulation order."""
    inp = synth.adapose_inputs(9, seed=11)
    out9 = _run(_net("bf16", max_chunk_views=10), inp)
    net1 = _net("bf16")
    for b in (0, 4, 8):
        one = {k: v[b:b + 1] for k, v in inp.items()}
        o1 = _run(net1, one)
        for k in OUT_KEYS:
            assert np.isfinite(out9[k][b]).all(), (b, k)
            assert _rel(out9[k][b:b + 1], o1[k]) < 2e-2, (b, k, _rel(out9[k][b:b + 1], o1[k]))


def test_bf16_batch256_one_chunk_identical_to_128_view_chunks():
    """BASELINE size: batch 256 with all 512 views in ONE cost-volume chunk (the default; c0 alone is 4.9e9 elements, past 32-bit
    indices) gives bit-identical outputs to 128-view chunks."""
    inp = synth.adapose_inputs(256, seed=0)
    a = _run(_net("bf16"), inp)
    b = _run(_net("bf16", max_chunk_views=128), inp)
    for k in OUT_KEYS:
        assert np.isfinite(a[k]).all(), k
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_mixed_object_batch_equals_per_head_runs(dtype):
    """BASELINE configs[4] (mixed cabinet / drawer / mug / pot batch, fp16 as the config names it; bf16 too): four weight sets
    (independent seeds), a shuffled batch routed by head — every pose's outputs are bit-identical to running its
    head's estimator on that head's samples alone, and the rank shards reassemble the batch."""
    from rgbmanip_amd.mixed import MixedObjectNet, shard_by_head
    sds = {h: synth.adapose_state_dict(seed=10 + h, prefix="module.") for h in range(4)}
    inp = synth.adapose_inputs(8, seed=21)
    heads = np.array([2, 0, 3, 1, 0, 2, 1, 3])
    mixed = MixedObjectNet(sds, dtype=dtype, head_streams=False)
    keys = ("img1", "choose1", "img2", "choose2", "P1", "P2", "depths")
    got = {k: v.cpu().numpy() for k, v in mixed(heads, *[inp[k] for k in keys]).items()}
    for h in range(4):
        sel = np.nonzero(heads == h)[0]
        alone = _run(_net_sd(sds[h], dtype), {k: inp[k][sel] for k in inp})
        for k in OUT_KEYS:
            np.testing.assert_array_equal(got[k][sel], alone[k], err_msg=f"head {h} {k}")
    # two "ranks": each runs its shard, together they cover the batch with identical numbers
    for r in range(2):
        idx = shard_by_head(heads, r, 2)
        part = MixedObjectNet(sds, dtype=dtype)(heads[idx], *[inp[k][idx] for k in keys])
        for k in OUT_KEYS:
            np.testing.assert_array_equal(part[k].cpu().numpy(), got[k][idx], err_msg=f"rank {r} {k}")
    # every head on its own stream (head_streams, the default: the heads' kernels overlap on the device): the same numbers, three times over
    streamed = MixedObjectNet(sds, dtype=dtype)
    assert streamed.head_streams
    for rep in range(3):
        o = streamed(heads, *[inp[k] for k in keys])
        torch.cuda.synchronize()
        for k in OUT_KEYS:
            np.testing.assert_array_equal(o[k].cpu().numpy(), got[k], err_msg=f"head_streams run {rep} {k}")


def test_fp16_sweep_conv0_stable_and_matches_tile_conv0(inputs):
    """The f16_t instantiation of the depth-sweeping conv0: bit-identical over repeated runs (an earlier build, in which hipcc
    had hoisted the reference-feature conversions out of the plane loop, varied from run to run) and within fp16 rounding of
    the halo-tile conv0 (debug flag 4096 selects the tile kernel for fp16 nets)."""
    from rgbmanip_amd import _lib
    lib = _lib.load()

    def c0(flag):
        _lib.check(lib.rgbm_debug_flags(flag))
        try:
            net = _net("fp16", cost_impl=3, options={"sparse_dec": 0})      # whole c0 volume
            _run(net, inputs, stop_after=2)
            return net.fetch(2, "c0", 4 * 24 * 224 * 224 * 8).view(4, 24, 224, 224, 8).float().cpu().numpy()
        finally:
            _lib.check(lib.rgbm_debug_flags(0))
    runs = [c0(0) for _ in range(4)]
    for r in runs[1:]:
        np.testing.assert_array_equal(runs[0], r)
    tile = c0(4096)
    scale = np.abs(tile).max()
    assert np.abs(runs[0] - tile).max() / scale < 2e-3
    assert np.abs(runs[0] - tile).mean() / np.abs(tile).mean() < 1e-4
