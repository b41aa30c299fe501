"""-m gpu: parity at the sizes BASELINE.json names, where the dispatcher picks the kernels that carry the benched step
(M >= 65536 rule: the persistent role-specialised implicit GEMM with the 256 x 128 tile, the persistent split-pair plane sweep,
the one-chunk cost volume) — the B = 2 golden tests never reach them.

 * configs[1] (batch 256): bf16x3 and bf16 forward at B = 256, fp32 at B = 48 (still >= 65536 GEMM rows in layer3 / layer4 / up_1),
   four strided poses' ten outputs against the CPU oracle (`oracle/adapose_ref`, pinned to the reference's golden vectors by
   tests/test_oracle_golden.py; the oracle is batch-independent: eval-mode BatchNorm, so pose b alone == pose b in the batch);
 * the run-to-run determinism gate at the benched shape (the plane-sweep kernels once showed a ~1e-3 per-workgroup
   corruption; every shipped dtype is run 20 times and bit-compared);
 * configs[4]: the per-rank workload of the 2048-pose mixed-object fp16 batch over 8 ranks (256 poses, one head);
 * configs[3]: the per-rank workload of the 4096-env PPO loop (512 envs per rank) on two ranks that share this box's GPU
   through gloo: one in-loop `PPO.run` iteration, identical parameters on both ranks.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from rgbmanip_amd import synth  # noqa: E402

RTOL_FP32 = 1e-4            # north_star: 1e-4 relative, fp32
OUT_KEYS = ["view1_nocs", "view2_nocs", "view1_depth", "view2_depth", "view1_r", "view2_r", "view1_t", "view2_t",
            "view1_s", "view2_s"]
B_FULL, B_FP32 = 256, 48
POSES_FULL = (0, 85, 170, 255)
POSES_FP32 = (0, 16, 32, 47)
# 2x the errors measured against the reference golden at B = 2 (tests/test_gpu_adapose.py)
GATE_BF16 = {"nocs": 2.7e-2, "depth": 1.0e-2, "r": 6.0e-3, "t": 3.5e-3, "s": 1.2e-3}
GATE_FP16 = {"nocs": 2.4e-3, "depth": 1.7e-3, "r": 1.2e-4, "t": 3.0e-4, "s": 7.0e-5}


@pytest.fixture(autouse=True)
def _release_device_memory():
    """every test here builds batch-256 networks (34-67 GB of workspace each): give the blocks back before the next one"""
    yield
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def _oracle_poses(sd_np, inp, poses):
    """{pose index: the oracle's ten outputs for that pose alone}."""
    from oracle import adapose_ref
    sd = adapose_ref.to_torch_sd(sd_np)
    out = {}
    for b in poses:
        t = {k: torch.from_numpy(v[b:b + 1]) for k, v in inp.items()}
        o = adapose_ref.adapose_forward(sd, t["img1"], t["choose1"], t["img2"], t["choose2"], t["P1"], t["P2"], t["depths"])
        out[b] = {k: v[0].numpy() for k, v in o.items()}
    return out


@pytest.fixture(scope="module")
def inputs256():
    return synth.adapose_inputs(B_FULL, seed=0)


@pytest.fixture(scope="module")
def oracle256(inputs256):
    return _oracle_poses(synth.adapose_state_dict(seed=0), inputs256, sorted(set(POSES_FULL) | set(POSES_FP32)))


def _net(dtype, **kw):
    from rgbmanip_amd.adapose import AdaPoseNet
    return AdaPoseNet(synth.adapose_state_dict(seed=0, prefix="module."), dtype=dtype, **kw)


def _dev_inputs(inp, n=None):
    keys = ("img1", "choose1", "img2", "choose2", "P1", "P2", "depths")
    return [torch.from_numpy(inp[k][:n]).cuda() for k in keys]


def _forward(net, args):
    out = net(*args)
    torch.cuda.synchronize()
    return out


def _errors(out, oracle, poses):
    errs = {}
    for k in OUT_KEYS:
        got = out[k].cpu().numpy()
        errs[k] = max(_rel(got[b], oracle[b][k]) for b in poses)
        assert np.isfinite(got).all(), k
    return errs


def test_bf16x3_batch256_vs_oracle(inputs256, oracle256):
    """The mode that carries the 1e-4 claim, at the benched batch: `conv_igemm_ws_kernel<bx3_t, WIDE>`, the persistent tile walk
    of `conv0_sweep_x3_kernel` and the one-chunk cost volume, end to end against the oracle."""
    out = _forward(_net("bf16x3"), _dev_inputs(inputs256))
    errs = _errors(out, oracle256, POSES_FULL)
    print("bf16x3 B=256 vs oracle (poses %s):" % (POSES_FULL,), errs)
    for k in OUT_KEYS:
        assert errs[k] < RTOL_FP32, (k, errs)


def test_bf16x3_batch256_on_the_headline_crop_inputs_vs_oracle():
    """The inputs the driver-timed `value` is measured on (bench.make_inputs_crop: 480x640 frames -> the reference's crop window ->
    rgbm_prepare_inputs; masks span their crops, 84.5 % of the sweep's tiles needed) in the mode that carries the 1e-4 claim, at the
    benched batch, four poses against the oracle.  (test_bf16x3_batch256_vs_oracle draws SURVEY 8d's in-crop ellipses.)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    host, dev, _ = bench.make_inputs_crop(B_FULL, torch.device("cuda:0"), seed=0)
    oracle = _oracle_poses(synth.adapose_state_dict(seed=0), {k: host[k] for k in ("img1", "choose1", "img2", "choose2", "P1", "P2", "depths")}, POSES_FULL)
    net = _net("bf16x3")
    out = net(dev["img1"], dev["choose1"], dev["img2"], dev["choose2"], dev["P1"], dev["P2"], dev["depths"])
    torch.cuda.synchronize()
    errs = _errors(out, oracle, POSES_FULL)
    print("bf16x3 B=256, crop-path inputs, vs oracle (poses %s):" % (POSES_FULL,), errs)
    for k in OUT_KEYS:
        assert errs[k] < RTOL_FP32, (k, errs)


def test_fp32_batch48_vs_oracle(inputs256, oracle256):
    """fp32 at B = 48: 2 * 48 * 784 = 75264 GEMM rows, so layer3 / layer4 / up_1 run the persistent kernels here too."""
    out = _forward(_net("fp32"), _dev_inputs(inputs256, B_FP32))
    errs = _errors(out, oracle256, POSES_FP32)
    print("fp32 B=48 vs oracle:", errs)
    for k in OUT_KEYS:
        assert errs[k] < RTOL_FP32, (k, errs)


@pytest.mark.parametrize("dtype,gate", [("bf16", GATE_BF16), ("fp16", GATE_FP16)])
def test_16bit_batch256_vs_oracle(inputs256, oracle256, dtype, gate):
    """The throughput modes at the benched batch, gated at twice the errors they show at B = 2."""
    out = _forward(_net(dtype), _dev_inputs(inputs256))
    errs = _errors(out, oracle256, POSES_FULL)
    print(f"{dtype} B=256 vs oracle:", errs)
    for k in OUT_KEYS:
        assert errs[k] < gate[k.split("_")[1]], (k, errs)


@pytest.mark.parametrize("upconv", [3, 7])
@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
def test_upconv_path_equals_resize_then_conv_at_batch(inputs256, dtype, upconv):
    """up_1 / up_2 as low-resolution 1x1 GEMM + tap combination (upconv = 3), and additionally up_3 + final in the one-kernel
    tail of upconv_final.hip (upconv = 7), against the x2 resize + 3x3 conv on the up-sampled grid (option upconv = 0) at
    B = 64: the same function in a different summation order."""
    args = _dev_inputs(inputs256, 64)
    a = {k: v.cpu().numpy() for k, v in _forward(_net(dtype, options={"upconv": upconv}), args).items()}
    b = {k: v.cpu().numpy() for k, v in _forward(_net(dtype, options={"upconv": 0}), args).items()}
    tol = RTOL_FP32 if dtype == "bf16x3" else None
    for k in OUT_KEYS:
        e = _rel(a[k], b[k])
        print(dtype, k, "upconv vs resize+conv", e)
        assert e < (tol if tol else 2 * GATE_BF16[k.split("_")[1]]), (k, e)


@pytest.mark.parametrize("dtype", ["bf16", "bf16x3", "fp16"])
def test_sparse_cost_regularisation_is_bit_identical_to_dense(inputs256, dtype):
    """Option sparse_dec = 2 (default: plane sweep, conv1 .. conv5, conv7, conv9 only on the tiles inside the chosen pixels'
    dependency cones) and 1 (conv7 / conv9 only) against 0 (every layer dense) at B = 32: all ten outputs must agree BIT FOR BIT —
    nothing that is read may depend on a skipped tile.  Four poses get adversarial pixel sets: a 32 x 32 block in the top-left
    corner, the last rows of the crop, one pixel repeated 1024 times, and pixels scattered over the whole crop (nothing to skip)."""
    args = _dev_inputs(inputs256, 32)
    g = torch.Generator().manual_seed(11)
    for ch in (args[1], args[3]):                      # choose1, choose2: [B, 1024] pixel indices of the 224 x 224 crop
        yy, xx = torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij")
        ch[0] = (yy * 224 + xx).reshape(-1).to(ch)
        ch[1] = torch.arange(224 * 224 - 1024, 224 * 224).to(ch)
        ch[2] = torch.full((1024,), 100 * 224 + 7).to(ch)
        ch[3] = torch.sort(torch.randperm(224 * 224, generator=g)[:1024])[0].to(ch)
    outs = {}
    # The sparse nets run FIRST and with a poisoned workspace (every byte 0xFF = NaN / -1 in front of each forward): torch.empty +
    # the caching allocator would otherwise hand them the block a previous dense net left behind — correct dense values in the same
    # layout, so that a read of a skipped tile could not fail (round-3 verdict).  The dense reference runs last, also poisoned.
    for sd in (2, 1, 0):
        net = _net(dtype, options={"sparse_dec": sd}, poison_workspace=True)
        _forward(net, args)
        outs[sd] = {k: v.clone() for k, v in _forward(net, args).items()}
        net.close()
        del net
    for sd in (1, 2):
        for k in OUT_KEYS:
            assert torch.equal(outs[sd][k].view(torch.int32), outs[0][k].view(torch.int32)), (dtype, sd, k,
                                                                                                float((outs[sd][k] - outs[0][k]).abs().max()))
    for k in OUT_KEYS:
        assert torch.isfinite(outs[2][k]).all(), k


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
def test_sparse_cost_regularisation_with_a_chunked_cost_volume(inputs256, dtype):
    """The tile masks and the sweep's tile list are built per cost-volume chunk: 40 views in chunks of 24 + 16 (max_chunk_views = 24,
    ragged last chunk) with sparse_dec = 2 must give the dense result of the same chunking bit for bit (another chunk size may pick
    other kernels for some layers, i.e. another summation order)."""
    args = _dev_inputs(inputs256, 20)
    net = _net(dtype, max_chunk_views=24, poison_workspace=True)      # sparse net first, workspace poisoned before every forward
    _forward(net, args)
    cur = {k: v.clone() for k, v in _forward(net, args).items()}
    net.close()
    del net
    ref = {k: v.clone() for k, v in _forward(_net(dtype, max_chunk_views=24, options={"sparse_dec": 0}, poison_workspace=True), args).items()}
    for k in OUT_KEYS:
        assert torch.equal(cur[k].view(torch.int32), ref[k].view(torch.int32)), (dtype, k, float((cur[k] - ref[k]).abs().max()))


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16", "fp16"])
def test_one_kernel_stem_equals_copy_conv_pool(inputs256, dtype):
    """conv1 7x7 + ReLU + max-pool in one kernel from the NCHW images (option stem = 1, the 16-bit default) against the padded copy,
    the implicit-GEMM conv and the pool kernel (stem = 0, the split-pair default) at B = 64: same function, another K order."""
    args = _dev_inputs(inputs256, 64)
    a = {k: v.cpu().numpy() for k, v in _forward(_net(dtype, options={"stem": 1}), args).items()}
    b = {k: v.cpu().numpy() for k, v in _forward(_net(dtype, options={"stem": 0}), args).items()}
    gate16 = GATE_BF16 if dtype == "bf16" else GATE_FP16
    for k in OUT_KEYS:
        e = _rel(a[k], b[k])
        print(dtype, k, "fused stem vs unfused", e)
        assert e < (RTOL_FP32 if dtype == "bf16x3" else 2 * gate16[k.split("_")[1]]), (k, e)


@pytest.mark.parametrize("dtype,B,reps", [("bf16", B_FULL, 20), ("bf16x3", B_FULL, 20), ("fp16", B_FULL, 20), ("fp32", B_FULL, 4)])
def test_forward_is_bit_stable_at_bench_shape(inputs256, dtype, B, reps):
    """`reps` forwards of the same batch must agree bit for bit in all ten outputs (what tools/check_determinism.py does by hand)."""
    net = _net(dtype)
    args = _dev_inputs(inputs256, B)
    ref = {k: v.clone() for k, v in _forward(net, args).items()}
    bad = []
    for r in range(1, reps):
        cur = _forward(net, args)
        for k in OUT_KEYS:
            if not torch.equal(ref[k].view(torch.int32), cur[k].view(torch.int32)):
                bad.append((r, k, float((ref[k] - cur[k]).abs().max())))
    assert not bad, bad[:10]


@pytest.mark.parametrize("dtype", ["bf16", "bf16x3", "fp16"])
def test_overlapping_half_batches_equal_the_one_stream_forward(inputs256, dtype):
    """AdaPoseNet(split_streams=True) runs a batch as two half batches on two HIP streams (their kernels overlap on the device).  All ten
    outputs must equal the one-stream forward bit for bit, run after run (poisoned workspaces): with hipcc's packed fp32 instructions in
    the library they intermittently did not (14-19 of 25 runs; the library is built without them, tests/test_cabi_symbols.py
    checks the shipped code objects).  12 overlapped forwards per storage type."""
    args = _dev_inputs(inputs256, B_FULL)
    ref = {k: v.clone() for k, v in _forward(_net(dtype), args).items()}
    net = _net(dtype, poison_workspace=True, split_streams=True)
    bad = []
    for r in range(12):
        cur = _forward(net, args)
        bad += [(r, k, float((ref[k] - cur[k]).abs().max())) for k in OUT_KEYS if not torch.equal(ref[k].view(torch.int32), cur[k].view(torch.int32))]
    assert net._last_split and not bad, bad[:10]
    with pytest.raises(Exception, match="split_streams"):
        net.fetch(B_FULL, "prob", 16)


def test_mixed_object_rank_workload_of_2048_pose_batch():
    """BASELINE configs[4] as one of its 8 ranks sees it: a 2048-pose batch with interleaved heads, sorted by head and cut into
    8 contiguous shards -> rank 5 owns 256 poses of ONE head; its `MixedObjectNet` result in fp16 must be bit-identical to that
    head's estimator alone, and two of its poses must sit inside the fp16 gates against the oracle with that head's weights."""
    from rgbmanip_amd.adapose import AdaPoseNet
    from rgbmanip_amd.mixed import MixedObjectNet, shard_by_head
    world, rank, per = 8, 5, 256
    heads_global = np.arange(world * per) % 4
    idx = shard_by_head(heads_global, rank, world)
    my_heads = heads_global[idx]
    assert len(idx) == per and len(np.unique(my_heads)) == 1
    head = int(my_heads[0])
    covered = np.concatenate([shard_by_head(heads_global, r, world) for r in range(world)])
    assert np.array_equal(np.sort(covered), np.arange(world * per))          # every pose on exactly one rank
    inp = synth.adapose_inputs(per, seed=100 + rank)                         # this rank's 256 poses of the synthetic batch
    sds = {h: synth.adapose_state_dict(seed=10 + h, prefix="module.") for h in range(4)}
    args = _dev_inputs(inp)
    mixed = MixedObjectNet(sds, dtype="fp16")
    got = {k: v.cpu().numpy() for k, v in mixed(my_heads, *args).items()}
    assert list(mixed.nets) == [head]                                        # only that head's weights were ever uploaded
    alone = {k: v.cpu().numpy() for k, v in _forward(AdaPoseNet(sds[head], dtype="fp16"), args).items()}
    for k in OUT_KEYS:
        np.testing.assert_array_equal(got[k], alone[k], err_msg=k)
    poses = (3, 200)
    ora = _oracle_poses({k[len("module."):]: v for k, v in sds[head].items()}, inp, poses)
    errs = {k: max(_rel(got[k][b], ora[b][k]) for b in poses) for k in OUT_KEYS}
    print("fp16 mixed-object rank workload vs oracle:", errs)
    # other weights (seed 10 + head) and other poses than the B = 2 golden: gates at 2x what this check measured
    # (nocs 2.6e-3, depth 3.9e-4, R 6.7e-5, t 1.1e-4, s 2.9e-4)
    gate = {"nocs": 5.3e-3, "depth": 8.0e-4, "r": 1.4e-4, "t": 2.3e-4, "s": 6.0e-4}
    for k in OUT_KEYS:
        assert errs[k] < gate[k.split("_")[1]], (k, errs)


# ------------------------------------------------------------------------------------------------ configs[3] per-rank workload
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ppo_rank(rank, world, port, n_envs, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_ppo import CFG
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import ADAPOSE_CFGS
    from rgbmanip_amd.control_interface import ControlInterface
    from rgbmanip_amd.estimator import AdaPoseEstimator_v5
    from rgbmanip_amd.ppo import PPO
    ecfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_prepare_seed=1)
    est = AdaPoseEstimator_v5(None, ecfg, None, state_dict=synth.adapose_state_dict(seed=0, prefix="module."), dtype="bf16")
    env = se.SyntheticMultiVecEnv(n_envs, "cuda", seed=1, env_id_offset=rank * n_envs)        # global env ids [r*N, (r+1)*N)
    ppo = PPO(ControlInterface(env, est, se.SyntheticManipulation(env), synth.control_cfg("cabinet", 0.0)), CFG)
    before = ppo.actor_critic.flat.clone()
    ppo.run(1, log_interval=1, save_interval=10 ** 9)
    torch.cuda.synchronize()
    q.put(dict(rank=rank, world=ppo.world, flat=ppo.actor_critic.flat.cpu().numpy(), changed=not torch.equal(before, ppo.actor_critic.flat),
               obs_shape=tuple(ppo.storage.observations.shape), fps=ppo.last_fps, lr=ppo.step_size,
               episodes=int(env.episode.min()), first_env=int(rank * n_envs)))
    dist.barrier()
    dist.destroy_process_group()


def test_ppo_rank_workload_of_4096_env_loop_two_ranks():
    """BASELINE configs[3] shards 4096 envs as 512 per GPU.  Two such ranks (global env ids 0..511 and 512..1023) run one full
    in-loop iteration each — 16 ControlInterface steps with the bf16 estimator, GAE with the global advantage statistics,
    32 optimiser steps with the flat-gradient all-reduce — on this box's one GPU (gloo transport; RCCL wants one device per
    rank): both ranks must end with bit-identical, finite, updated parameters and the same adaptive learning rate."""
    world, n_envs, port = 2, 512, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ppo_rank, args=(r, world, port, n_envs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in res:
        assert r["world"] == 2 and r["changed"] and np.isfinite(r["flat"]).all()
        assert r["obs_shape"] == (16, n_envs, 60) and r["fps"] > 0 and r["episodes"] >= 3
    np.testing.assert_array_equal(res[0]["flat"], res[1]["flat"])
    assert res[0]["lr"] == res[1]["lr"]
