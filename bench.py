#!/usr/bin/env python3
"""Headline benchmark: AdaPose poses/sec at batch 256 (BASELINE.json configs[1]) on N MI355X.

One "step" = one pass of the estimator hot path over one batch of 256 synthetic poses per GPU:
  HIP network forward (2 x 256 views of 224x224, all 10 outputs)  +  device post-processing to world boxes,
with every input already resident in HBM.  The headline inputs come through the reference's own crop logic (480x640 frames with an
elliptical object mask -> crop window of lib/utils.py:10-38 -> 224x224 crop + 1024 chosen pixels, rgbm_prepare_inputs), so the
masks span their crops as the reference's do; `value_dense` (every layer computed densely), `value_worst_case` (chosen pixels all
over the crop) and `value_survey_masks` (SURVEY 8d's ellipses inside the crop, the headline of rounds 1-3) are timed in the same run.  N > 1: one process per GPU (torch.distributed, RCCL); poses are
independent, so ranks shard the batch with no data-path collective (weak scaling, 256 poses per GPU).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the
dominant kernel (the library's rgbm_prof_* hooks, same stream as the launches); `cpu_baseline` times the CPU
oracle (PyTorch-CPU restatement of the reference, proven equal to it on golden vectors) on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

GFLOP_PER_POSE = 163.68          # hooks on the reference module, 2*MAC, full 10-output forward
# MI355X_MICROARCH.md dense MFMA peaks (the profiler rows of the 16-bit kernels are labelled bf16).  bf16x3 spends three bf16
# MFMAs per algorithmic multiply-add (hi*hi + lo*hi + hi*lo), so its ceiling for ALGORITHMIC flops is a third of the bf16 peak
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}



def make_inputs(B, device, unique=16):
    from rgbmanip_amd import synth
    u = min(unique, B)
    base = synth.adapose_inputs(u, seed=0)
    reps = (B + u - 1) // u
    out = {}
    for k, v in base.items():
        t = np.concatenate([v] * reps, axis=0)[:B]
        out[k] = t
    if os.environ.get("RGBM_BENCH_CHOOSE") == "uniform":     # worst case of the sparse cost regularisation: pixels all over the crop need every tile
        g = np.random.default_rng(7)
        for k in ("choose1", "choose2"):
            out[k] = np.sort(np.stack([g.permutation(224 * 224)[:out[k].shape[1]] for _ in range(B)]), axis=1).astype(np.int64)
    dev = {
        "img1": torch.from_numpy(out["img1"]).to(device), "img2": torch.from_numpy(out["img2"]).to(device),
        "choose1": torch.from_numpy(out["choose1"]).to(device=device, dtype=torch.int32),
        "choose2": torch.from_numpy(out["choose2"]).to(device=device, dtype=torch.int32),
        "P1": torch.from_numpy(out["P1"]).to(device), "P2": torch.from_numpy(out["P2"]).to(device),
        "depths": torch.from_numpy(out["depths"]).to(device),
        "K1": torch.from_numpy(out["K1"]).to(device), "E1": torch.from_numpy(out["E1"]).to(device),
    }
    return out, dev


def make_inputs_crop(B, device, seed=0, keep_frames=False):
    """The headline workload: B stereo pairs as the reference's pipeline produces them.  Two look-at cameras per pose (the geometry of
    synth.adapose_inputs), a 480x640 frame per view with an elliptical object mask around the projected target (semi-axes 40-120 px
    vertically, 40-150 px horizontally), then `prepare_model_input` on the device (rgbm_prepare_inputs: crop window per
    lib/utils.py:10-38 = the mask's longer side rounded up to a multiple of 40, INTER_LINEAR crop-resize to 224, normalisation,
    1024-pixel subset, cropped intrinsics; interface_v5.py:58-170).  Returns (host dict of numpy arrays for the oracle, device dict,
    frames dict or None)."""
    from rgbmanip_amd import synth
    from rgbmanip_amd.adapose import prepare_inputs
    g = np.random.default_rng(424242 + seed)
    E = np.empty((B, 2, 4, 4))
    ell = np.empty((B, 2, 4), dtype=np.float32)
    for b in range(B):
        target = g.uniform(-0.05, 0.05, size=3) + np.array([0.0, 0.0, 0.5])
        d0 = g.normal(size=3)
        d0[2] = abs(d0[2]) * 0.3
        d0 /= np.linalg.norm(d0)
        eye1 = target + d0 * g.uniform(0.55, 0.9)
        side = np.cross(d0, np.array([0.0, 0.0, 1.0]))
        side /= np.linalg.norm(side)
        eye2 = eye1 + side * g.uniform(0.15, 0.35) + np.array([0, 0, g.uniform(-0.05, 0.05)])
        for v, eye in enumerate((eye1, eye2)):
            E[b, v] = synth._lookat_extrinsic(eye, target)
            ell[b, v] = (240 + g.uniform(-50, 50), 320 + g.uniform(-80, 80), g.uniform(40, 120), g.uniform(40, 150))     # cy, cx, ry, rx
    fx = 240.0 / np.tan(0.5)
    K0 = torch.tensor([[fx, 0, 320.0], [0, fx, 240.0], [0, 0, 1.0]], dtype=torch.float64, device=device).repeat(B, 1, 1)
    gen = torch.Generator(device=device).manual_seed(1000 + seed)
    yy, xx = torch.meshgrid(torch.arange(480, device=device, dtype=torch.float32), torch.arange(640, device=device, dtype=torch.float32), indexing="ij")
    prep, frames = [], {}
    for v in range(2):
        noise = torch.rand(B, 480, 640, 3, generator=gen, device=device)
        ph = torch.rand(B, 3, generator=gen, device=device) * 6.2832
        smooth = 0.5 + 0.125 * (torch.cos(0.02 * xx[None] + ph[:, 0, None, None]) + torch.cos(0.03 * yy[None] + ph[:, 1, None, None]) +
                                torch.cos(0.011 * (xx + yy)[None] + ph[:, 2, None, None]))
        rgb = (0.5 * noise + 0.5 * smooth[..., None]).clamp_(0.0, 1.0)
        e = torch.from_numpy(ell[:, v]).to(device)
        mask = ((((yy[None] - e[:, 0, None, None]) / e[:, 2, None, None]) ** 2 + ((xx[None] - e[:, 1, None, None]) / e[:, 3, None, None]) ** 2) <= 1).to(torch.uint8)
        prep.append(prepare_inputs(rgb, mask, K0, 224, 1024, seed + v))
        if keep_frames:
            frames[f"rgb{v + 1}"], frames[f"mask{v + 1}"] = rgb, mask
        del noise, smooth
    assert int(prep[0]["valid"].sum()) == B and int(prep[1]["valid"].sum()) == B
    Ed = torch.from_numpy(E).to(device)
    P = []
    for v in range(2):
        Pm = torch.eye(4, dtype=torch.float64, device=device).repeat(B, 1, 1)
        Pm[:, :3, :] = prep[v]["Kcrop"] @ Ed[:, v, :3, :]
        P.append(Pm.float())
    depths = torch.from_numpy(np.tile(np.arange(0.1, 0.1 * (24 - 0.5) + 0.1, 0.1, dtype=np.float32)[None], (B, 1))).to(device)
    dev = {"img1": prep[0]["img"], "img2": prep[1]["img"], "choose1": prep[0]["choose"], "choose2": prep[1]["choose"],
           "P1": P[0], "P2": P[1], "depths": depths, "K1": prep[0]["Kcrop"], "E1": Ed[:, 0].contiguous()}
    host = {k: v.cpu().numpy() for k, v in dev.items()}
    host["choose1"], host["choose2"] = host["choose1"].astype(np.int64), host["choose2"].astype(np.int64)
    if keep_frames:
        frames.update(K=K0, E1=Ed[:, 0].contiguous(), E2=Ed[:, 1].contiguous())
    return host, dev, (frames if keep_frames else None)


def _mark(msg):
    """Progress marker on stderr (RGBM_BENCH_TRACE=1): which leg a rank is in, for diagnosing multi-rank stalls."""
    if os.environ.get("RGBM_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')} t={time.perf_counter():.1f}] {msg}", file=sys.stderr, flush=True)


def tree_hash():
    """sha256 (first 16 hex digits) over the library sources: PMC measurements are only valid for the tree
    they were taken on (tools/pmc_traffic.py stamps the same value into profiles/hbm_traffic.json)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "rgbmanip_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "rgbmanip_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "rgbmanip_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "rgbmanip_amd", "csrc", "*.inc")) +
                   [os.path.join(ROOT, "rgbmanip_amd", "csrc", "build.sh")])      # the compiler flags are part of what was measured
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def accuracy_vs_golden(net, device):
    """Worst relative error (max |a - b| / max |b|) of the 10 network outputs and of the world box against the reference's
    own outputs on the committed golden inputs (tests/golden/adapose_b2.npz: B = 2, produced by importing the reference
    module, tools/make_goldens.py).  The reference box is the device post-processing applied to the golden network outputs,
    so the figure isolates the network's arithmetic.  north_star's gate is 1e-4."""
    from rgbmanip_amd import synth
    from rgbmanip_amd.adapose import postprocess
    g = np.load(os.path.join(ROOT, "tests", "golden", "adapose_b2.npz"))
    inp = synth.adapose_inputs(2, seed=0)
    out = net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
    torch.cuda.synchronize()
    keys = ["view1_nocs", "view2_nocs", "view1_depth", "view2_depth", "view1_r", "view2_r", "view1_t", "view2_t", "view1_s", "view2_s"]
    errs = {k: float(np.abs(out[k].cpu().double().numpy() - g[k]).max() / max(np.abs(g[k]).max(), 1e-12)) for k in keys}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    bb, _, _ = postprocess(out["view1_nocs"], out["view1_depth"], out["view1_r"], t(inp["choose1"]).int(), t(inp["K1"]), t(inp["E1"]))
    rb, _, _ = postprocess(t(g["view1_nocs"]).float(), t(g["view1_depth"]).float(), t(g["view1_r"]).float(), t(inp["choose1"]).int(),
                           t(inp["K1"]), t(inp["E1"]))
    bbox_err = float((bb - rb).abs().max() / rb.abs().max())
    return {"worst_output_rel_err": float(f"{max(errs.values()):.3e}"), "world_bbox_rel_err": float(f"{bbox_err:.3e}"),
            "per_output": {k: float(f"{v:.2e}") for k, v in errs.items()}, "meets_1e-4": bool(max(errs.values()) < 1e-4)}


def cpu_baseline_ppo(n_envs=4):
    """PPO leg on the host cores: the oracle's controller step — numpy camera + render of the same synthetic
    env, prepare_model_input, PyTorch-CPU network, numpy post-processing, policy act — timed on a bounded sample of envs, and
    the oracle's learn phase (GAE + 32 optimiser steps) on a full 16 x 512 rollout; env-steps/s is extrapolated to 512 envs."""
    from oracle import adapose_ref, postproc_ref, ppo_ref, synth_env_ref as sr
    from rgbmanip_amd import synth
    from rgbmanip_amd import synthetic_env as se
    from rgbmanip_amd.config import RL_CONTROLLER_CFG
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    tsd = adapose_ref.to_torch_sd(synth.adapose_state_dict(seed=0))
    psd = {k: torch.from_numpy(v) for k, v in synth.policy_state_dict(seed=0).items()}
    robots, boxes = (np.stack(a) for a in zip(*[se.sample_scene(i, 0) for i in range(n_envs)]))
    rng = np.random.default_rng(0)
    f = se.CAM_F

    def frame(shift):
        cam = np.zeros((n_envs, 7))
        cam[:, :3] = [0.0, 0.05 * shift, 0.7]
        cam[:, 3:] = [1.0, 0.0, 0.0, 0.0]
        K, E, rays = sr.camera_ref(cam, robots, boxes, f, f, 320.0, 240.0)
        color, mask = sr.render_ref(rays, boxes, f, f, 320.0, 240.0, 480, 640, env0=0)
        return K, E, color, mask

    def env_step():
        K, E1, c1, m1 = frame(0)
        _, E2, c2, m2 = frame(1)
        ins = []
        for i in range(n_envs):
            a = postproc_ref.prepare_model_input(c1[i], m1[i], K[i], 224, rng=rng)
            b = postproc_ref.prepare_model_input(c2[i], m2[i], K[i], 224, rng=rng)
            if a[0] is None or b[0] is None:
                continue
            P1, P2 = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
            P1[:3] = a[3] @ E1[i][:3]
            P2[:3] = b[3] @ E2[i][:3]
            ins.append((a, b, P1, P2, E1[i]))
        if ins:
            st = lambda xs, dt: torch.from_numpy(np.stack(xs)).to(dt)  # noqa: E731
            dep = (torch.arange(24, dtype=torch.float32) * 0.1 + 0.1)[None].repeat(len(ins), 1)
            o = adapose_ref.adapose_forward(tsd, st([i[0][0] for i in ins], torch.float32), st([i[0][1] for i in ins], torch.int64),
                                            st([i[1][0] for i in ins], torch.float32), st([i[1][1] for i in ins], torch.int64),
                                            st([i[2] for i in ins], torch.float32), st([i[3] for i in ins], torch.float32), dep)
            for q, it in enumerate(ins):
                postproc_ref.bbox_world(o["view1_nocs"][q].numpy(), o["view1_depth"][q].numpy(), o["view1_r"][q].numpy(), it[0][1], it[0][3], it[4])
        ppo_ref.act(psd, torch.zeros(n_envs, 60), torch.zeros(n_envs, 12))
        return len(ins)
    t0 = time.perf_counter()
    used = env_step()
    per_env_step = (time.perf_counter() - t0) / max(used, 1)
    T, N = 16, 512
    roll = {k: torch.from_numpy(v) for k, v in synth.ppo_rollout(T, N, seed=0).items()}
    with torch.no_grad():      # the rollout's recorded policy outputs (untimed: the env-step leg above already counts policy act)
        lp, _, _, mu, sig = ppo_ref.evaluate(psd, roll["observations"].reshape(T * N, -1), roll["actions"].reshape(T * N, -1))
    roll["actions_log_prob"] = lp.reshape(T, N, 1) - 0.01
    roll["mu"] = mu.reshape(T, N, -1)
    roll["sigma"] = sig.reshape(T, N, -1) - 0.005
    t0 = time.perf_counter()
    ret, adv = ppo_ref.compute_returns(roll["rewards"], roll["dones"], roll["values"], roll["last_values"], 0.98, 0.98)
    lc = RL_CONTROLLER_CFG["learn"]
    ppo_ref.ppo_update(psd, roll, ret, adv, lc, lc["learning_rate"])
    learn_s = time.perf_counter() - t0
    value = T * N / (T * N * per_env_step + learn_s)
    return {"value": round(value, 3), "unit": "env-steps/s", "cores": cores, "kind": "port", "learn_s": round(learn_s, 3),
            "cpu_s_per_env_step": round(per_env_step, 3),
            "sample": f"controller step of the oracle (numpy synthetic camera, prepare_model_input, PyTorch-CPU network, numpy post-processing, "
                      f"policy act) timed on {used} envs x 1 step, oracle learn phase (GAE + 32 optimiser steps) on a full 16 x 512 rollout; "
                      f"extrapolated to 16 transitions x 512 envs, {cores} threads"}


OUT_KEYS = ["view1_nocs", "view2_nocs", "view1_depth", "view2_depth", "view1_r", "view2_r", "view1_t", "view2_t", "view1_s", "view2_s"]


def cpu_baseline(host, n_chunks=6, chunk=2, n_unique=16):
    """Oracle (kind "port") on the host cores: network forward + numpy post-processing, bounded sample of the benched workload's
    own poses (chunks of `chunk` of the batch's unique poses, last ones first).  Returns (result dict, {pose index: oracle
    outputs}) — the second is the checker for `accuracy.at_batch`: what the device produced for those poses inside the timed
    batch is compared with it after the timing."""
    from oracle import adapose_ref, postproc_ref
    from rgbmanip_amd import synth
    cores = min(os.cpu_count() or 1, 32)       # oneDNN convs at batch 2 stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(cores)
    sd = adapose_ref.to_torch_sd(synth.adapose_state_dict(seed=0))
    n_unique = min(n_unique, host["img1"].shape[0])
    chunks = [[(n_unique - 1 - (c * chunk + j)) % n_unique for j in range(chunk)] for c in range(n_chunks + 1)]
    ref = {}

    def one(idx):
        t = {k: torch.from_numpy(np.ascontiguousarray(v[idx])) for k, v in host.items()}
        o = adapose_ref.adapose_forward(sd, t["img1"], t["choose1"], t["img2"], t["choose2"], t["P1"], t["P2"], t["depths"])
        for j, b in enumerate(idx):
            postproc_ref.bbox_world(o["view1_nocs"][j].numpy(), o["view1_depth"][j].numpy(), o["view1_r"][j].numpy(),
                                    host["choose1"][b], host["K1"][b], host["E1"][b])
            ref[b] = {k: o[k][j].numpy() for k in OUT_KEYS}
    one(chunks[0])                          # warm-up chunk
    t0 = time.perf_counter()
    for c in chunks[1:]:
        one(c)
    dt = time.perf_counter() - t0
    return ({"value": round(n_chunks * chunk / dt, 4), "unit": "poses/s", "cores": cores, "kind": "port",
             "sample": f"{n_chunks} timed chunks of {chunk} poses of the benched batch (1 warm-up chunk), fp32 PyTorch-CPU oracle forward + "
                       f"numpy compute_scale/bbox, {torch.get_num_threads()} threads"}, ref)


def at_batch_accuracy(dev_out, ref, B, n_unique=16):
    """Worst relative error of the ten outputs of the TIMED batch against the oracle, over the poses the CPU baseline evaluated:
    each at its first position in the batch and at its last replica (the batch tiles `n_unique` poses)."""
    errs = {k: 0.0 for k in OUT_KEYS}
    positions = []
    for u, o in ref.items():
        last = u + ((B - 1 - u) // n_unique) * n_unique
        for pos in sorted({u, last}):
            if pos >= B:
                continue
            positions.append(pos)
            for k in OUT_KEYS:
                got = dev_out[k][pos].double().cpu().numpy()
                errs[k] = max(errs[k], float(np.abs(got - o[k]).max() / max(np.abs(o[k]).max(), 1e-12)))
    worst = max(errs.values())
    return {"batch": B, "poses_checked": len(ref), "positions": len(positions), "worst_output_rel_err": float(f"{worst:.3e}"),
            "per_output": {k: float(f"{v:.2e}") for k, v in errs.items()}, "meets_1e-4": bool(worst < 1e-4),
            "checker": "oracle/adapose_ref outputs of the cpu_baseline leg (same poses, same run)"}


def time_steps(fn, warmup, steps):
    """seconds per call of fn() (device-synchronised on both sides)"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a child
    process (one rank per GPU over RCCL, what the driver's own N > 1 command does), pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def achievable_peaks(lib, device):
    """the peaks this box reaches, measured next to the datasheet ones (rgbm_microbench_*: a bare bf16 MFMA stream in the
    register shape of the implicit GEMM's multiply waves on constant and on pseudo-random operands, and a 16-byte copy).  ~0.3 s."""
    from rgbmanip_amd import _lib
    n = C.c_int()
    _lib.check(lib.rgbm_microbench_mfma_scratch_floats(C.byref(n)), "rgbm_microbench_mfma_scratch_floats")
    scratch = torch.empty(n.value, dtype=torch.float32, device=device)
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for name, rnd in (("mfma_bf16_constant_operands_TFLOPs", 0), ("mfma_bf16_random_operands_TFLOPs", 1)):
        fl = C.c_double()
        run = lambda it: _lib.check(lib.rgbm_microbench_mfma(_lib.ptr(scratch), it, rnd, C.byref(fl), _lib.stream_ptr()), "rgbm_microbench_mfma")
        run(20000)                                        # warm-up, clocks settle under load
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(3):
            run(60000)                                    # ~25-35 ms each: long enough for the power limit to act
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(3 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).fill_(1)
    dst = torch.empty_like(src)
    cp = lambda: _lib.check(lib.rgbm_microbench_copy(_lib.ptr(src), _lib.ptr(dst), nbytes, _lib.stream_ptr()), "rgbm_microbench_copy")
    cp()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        cp()
    e1.record()
    torch.cuda.synchronize()
    out["hbm_copy_GBps_read_plus_write"] = round(5 * 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    out["note"] = ("measured in this run on this GPU; roofline.peak / whole_net_hbm.peak_GBps stay the datasheet figures "
                   "(2.5 PFLOP/s dense bf16, 8 TB/s), these are additional denominators")
    return out


def prof_table(stats, steps):
    from rgbmanip_amd import _lib
    st = np.array(list(stats)).reshape(_lib.PROF_ROWS, 4)
    kernels = []
    for v in range(_lib.PROF_ROWS):
        n, ms, fl, by = st[v]
        if n > 0:
            kernels.append({"kernel": _lib.PROF_KERNELS[v][0], "dtype": _lib.PROF_KERNELS[v][1], "row": v, "launches_per_step": n / steps,
                            "avg_launch_ms": ms / n, "total_ms_per_step": ms / steps,
                            "tflops": fl / (ms * 1e-3) / 1e12, "algo_GBps": by / (ms * 1e-3) / 1e9, "flops_per_step": fl / steps})
    kernels.sort(key=lambda k: -k["total_ms_per_step"])
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="poses per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16", "bf16x3"])
    ap.add_argument("--inputs", default="crop", choices=["crop", "survey", "uniform"],
                    help="headline inputs: crop = 480x640 frames through the reference's crop window (masks span their crops); survey = "
                         "SURVEY 8d's ellipses inside the 224 crop (rounds 1-3); uniform = chosen pixels all over the crop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="views per cost-volume chunk (0 = library default)")
    ap.add_argument("--ppo-envs", type=int, default=512, help="envs per GPU for the PPO leg (0 = skip it)")
    ap.add_argument("--ppo-env", default="full", choices=["full", "bank"],
                    help="full: ControlInterface over the synthetic MultiVecEnv (480x640 frames); bank: pre-cropped 224x224 view bank")
    ap.add_argument("--no-prepare", action="store_true", help="skip the device-side prepare_model_input leg")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-object (4 heads) leg")
    ap.add_argument("--mixed-dtype", default="fp16", choices=["bf16", "fp16", "fp32", "bf16x3"], help="storage type of the mixed-object leg (configs[4] names fp16)")
    ap.add_argument("--ppo-iters", type=int, default=2, help="PPO learning iterations (the last one is reported)")
    ap.add_argument("--cost-impl", type=int, default=-1, help="A/B switch of the cost-volume path (see rgbm.h); -1 = default")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the dense / worst-case / survey-mask legs of the headline dtype")
    ap.add_argument("--no-modes", action="store_true", help="skip the fp32 / bf16x3 throughput + accuracy legs (the modes inside the 1e-4 gate)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the estimate() plugin-boundary leg (host numpy frames in, H2D included)")
    ap.add_argument("--no-ppo-default", action="store_true", help="skip the PPO leg's second run with the estimator's default cfg (bf16x3)")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the B = 1 / B = 8 latency leg (eager launches and hipGraph replay)")
    ap.add_argument("--no-peaks", action="store_true", help="skip the achievable-peak probes (bare MFMA stream, copy kernel; ~0.3 s)")
    ap.add_argument("--no-prof", action="store_true", help="timing experiment: no per-launch HIP events in the timed region (the roofline object is then empty)")
    ap.add_argument("--mode-steps", type=int, default=10, help="timed steps of each extra mode leg (behind --mode-warmup untimed ones)")
    ap.add_argument("--mode-warmup", type=int, default=2, help="untimed steps in front of each extra mode leg")
    ap.add_argument("--debug-flags", type=int, default=0, help="kernel A/B switches (rgbm_debug_flags); a non-zero value is echoed in config")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the golden-vector accuracy leg (profiling runs: keeps the trace to the timed steps)")
    args = ap.parse_args()
    if os.environ.get("RGBM_BENCH_CHOOSE") == "uniform":      # rounds 1-3 spelling of --inputs uniform
        args.inputs = "uniform"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} under a launcher that set WORLD_SIZE={world}: start it with --nproc-per-node {args.gpus}")
        # Launched plainly (`python bench.py --gpus N`, the shape of the N = 1 command): start the ranks as a CHILD process — this process has
        # not touched the GPU yet, and it never replaces itself (no exec) —, relay rank 0's JSON line and the child's exit code.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    # RGBM_BENCH_ONE_DEVICE=1 (+ RGBM_DIST_BACKEND=gloo) lets a 1-GPU box rehearse the multi-rank code path: every rank uses cuda:0
    if os.environ.get("RGBM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("RGBM_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from rgbmanip_amd import _lib, synth
    from rgbmanip_amd.adapose import AdaPoseNet, postprocess, sweep_tiles_needed_fraction
    lib = _lib.load()
    if args.debug_flags:
        lib.rgbm_debug_flags(args.debug_flags)
    sd0 = synth.adapose_state_dict(seed=0)
    net = AdaPoseNet(sd0, dtype=args.dtype, device=local_rank,
                     max_chunk_views=args.chunk or None, cost_impl=None if args.cost_impl < 0 else args.cost_impl)
    B = args.batch
    want_boundary = rank == 0 and world == 1 and not args.no_boundary
    frames = None
    if args.inputs == "crop":
        host, d, frames = make_inputs_crop(B, device, seed=rank, keep_frames=want_boundary)
        n_unique = B
    else:
        if args.inputs == "uniform":
            os.environ["RGBM_BENCH_CHOOSE"] = "uniform"
        host, d = make_inputs(B, device)
        n_unique = min(16, B)

    def mkstep(n, dd):
        def f():
            out = n(dd["img1"], dd["choose1"], dd["img2"], dd["choose2"], dd["P1"], dd["P2"], dd["depths"])
            bbox, ts, valid = postprocess(out["view1_nocs"], out["view1_depth"], out["view1_r"], dd["choose1"], dd["K1"], dd["E1"])
            return out, bbox, valid
        return f
    step = mkstep(net, d)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # Per-kernel table: ONE untimed step with every conv launch between two HIP events.  Inside the timed region only the launches
    # of the dominant kernel (the row with the largest total of that step) are bracketed — the roofline object is measured live over
    # the timed steps, and the other ~80 launches carry no events (bracketing all of them costs 0.35 ms per step, measured).
    stats_all = (C.c_double * (4 * _lib.PROF_ROWS))()
    dom_row = -1
    if not args.no_prof:
        _lib.check(lib.rgbm_prof_select(-1), "rgbm_prof_select")
        _lib.check(lib.rgbm_prof_start(), "rgbm_prof_start")
        step()
        torch.cuda.synchronize()
        _lib.check(lib.rgbm_prof_stop(stats_all), "rgbm_prof_stop")
        dom_row = prof_table(stats_all, 1)[0]["row"]
        barrier()
        _lib.check(lib.rgbm_prof_select(dom_row), "rgbm_prof_select")
        _lib.check(lib.rgbm_prof_start(), "rgbm_prof_start")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, bbox, valid = step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    if args.no_prof:      # timing experiment (one rank): the step time without the per-launch events, nothing else
        print(json.dumps({"ms_per_step": round(elapsed / args.steps * 1e3, 3), "note": "--no-prof: no per-launch HIP events in the timed region"}))
        return
    stats = (C.c_double * (4 * _lib.PROF_ROWS))()
    _lib.check(lib.rgbm_prof_stop(stats), "rgbm_prof_stop")
    _lib.check(lib.rgbm_prof_select(-1), "rgbm_prof_select")
    # how much of the plane sweep this rank's chosen pixels need (sparse cost regularisation is data dependent: a straggler shows here)
    ch_all = torch.cat([d["choose1"], d["choose2"]]).cpu().numpy()
    my_frac = sweep_tiles_needed_fraction(ch_all[:: max(1, len(ch_all) // 64)]) if args.dtype != "fp32" else 1.0
    per_rank = None
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # every rank's own numbers through one SUM all-reduce of a [world, 2] table (all-reduce is what RCCL and gloo both offer for device tensors)
        allr = torch.zeros(world, 2, dtype=torch.float64, device=device)
        allr[rank] = torch.tensor([elapsed_local / args.steps * 1e3, my_frac], dtype=torch.float64, device=device)
        dist.all_reduce(allr)
        per_rank = {"ms_per_step": [round(float(a[0]), 3) for a in allr], "sweep_tiles_needed_frac": [round(float(a[1]), 4) for a in allr]}
        per_rank["ms_per_step_min"], per_rank["ms_per_step_max"] = min(per_rank["ms_per_step"]), max(per_rank["ms_per_step"])
    n_valid = int(valid.sum().item())
    finite = bool(torch.isfinite(bbox).all().item())
    batch_outs = {}
    poison_same = None
    if rank == 0 and args.no_accuracy:
        batch_outs[args.dtype] = {k: v.clone() for k, v in out.items()}
    elif rank == 0:
        # the at-batch accuracy check uses one more forward of the same batch on a POISONED workspace (every byte 0xFF in front of the
        # forward): it must reproduce the timed step bit for bit — nothing read may come from a previous run — and is what the oracle sees
        net.poison_workspace = True
        pout, _, _ = step()
        torch.cuda.synchronize()
        net.poison_workspace = False
        poison_same = all(torch.equal(pout[k].view(torch.int32), out[k].view(torch.int32)) for k in OUT_KEYS)
        batch_outs[args.dtype] = {k: v.clone() for k, v in pout.items()}

    _mark("timed region done")
    # ---- accuracy of the benched mode and throughput + accuracy of the modes that meet north_star's 1e-4 gate (not part of `value`) ----
    acc_res, modes_res = None, None
    if rank == 0 and not args.no_accuracy:
        acc_res = accuracy_vs_golden(net, device)
    if rank == 0 and world == 1 and not args.no_modes:      # N = 1 information; at N > 1 the other ranks would idle at the next barrier meanwhile
        modes_res = {}
        for md in ("bf16x3", "fp32"):
            if md == args.dtype:
                continue
            mnet2 = AdaPoseNet(sd0, dtype=md, device=local_rank, max_chunk_views=args.chunk or None)
            macc = accuracy_vs_golden(mnet2, device)
            mstep2 = mkstep(mnet2, d)
            mdt = time_steps(mstep2, args.mode_warmup, args.mode_steps)
            mnet2.poison_workspace = True
            batch_outs[md] = {k: v.clone() for k, v in mstep2()[0].items()}
            torch.cuda.synchronize()
            modes_res[md] = {"poses_per_sec": round(B / mdt, 1), "ms_per_step": round(mdt * 1e3, 2), "batch": B, "steps": args.mode_steps,
                             "warmup": args.mode_warmup, "accuracy": macc}
            if md == "bf16x3" and not args.no_dense_leg:
                dn = AdaPoseNet(sd0, dtype=md, device=local_rank, max_chunk_views=args.chunk or None, options={"sparse_dec": 0})
                ddt = time_steps(mkstep(dn, d), args.mode_warmup, args.mode_steps)
                modes_res[md]["dense"] = {"poses_per_sec": round(B / ddt, 1), "ms_per_step": round(ddt * 1e3, 2)}
                del dn
            del mnet2
            torch.cuda.empty_cache()

    # ---- the same step with every layer dense, on the worst-case pixel sets, and on SURVEY 8d's in-crop ellipses (N = 1 information) ----
    legs = None
    if rank == 0 and world == 1 and args.dtype != "fp32" and not args.no_dense_leg:
        legs = {}
        dnet = AdaPoseNet(sd0, dtype=args.dtype, device=local_rank, max_chunk_views=args.chunk or None, options={"sparse_dec": 0})
        dstep = mkstep(dnet, d)
        for _ in range(args.warmup):
            dstep()
        torch.cuda.synchronize()
        _lib.check(lib.rgbm_prof_start(), "rgbm_prof_start")
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dstep()
        torch.cuda.synchronize()
        ddt = (time.perf_counter() - t1) / args.steps
        dstats = (C.c_double * (4 * _lib.PROF_ROWS))()
        _lib.check(lib.rgbm_prof_stop(dstats), "rgbm_prof_stop")
        dk = prof_table(dstats, args.steps)
        # flops the conv launches of one dense step EXECUTE (each launch's own shape: the commuted PSPUpsample GEMMs count a quarter of
        # the reference's multiply-adds; the tap-combination rows 37 / 38 are vector work and left out)
        exec_flops = sum(k["flops_per_step"] for k in dk if k["row"] not in (37, 38))
        legs["dense"] = {"poses_per_sec": round(B / ddt, 1), "ms_per_step": round(ddt * 1e3, 2), "steps": args.steps, "warmup": args.warmup,
                         "executed_conv_tflop_per_step": round(exec_flops / 1e12, 3),
                         "mfma_frac_executed_flops": round(exec_flops / ddt / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                         "algorithmic_tflops_over_peak": round(B / ddt * GFLOP_PER_POSE / 1e3 / PEAK_TFLOPS[args.dtype], 4)}
        del dnet
        torch.cuda.empty_cache()
        # worst case of the sparse path: pixels all over the crop need every tile (mask + list kernels and the indirection on top of dense)
        gq = np.random.default_rng(7)
        du = dict(d)
        for k in ("choose1", "choose2"):
            du[k] = torch.from_numpy(np.sort(np.stack([gq.permutation(224 * 224)[:1024] for _ in range(B)]), axis=1).astype(np.int32)).to(device)
        wdt = time_steps(mkstep(net, du), args.warmup, args.steps)
        legs["worst_case"] = {"poses_per_sec": round(B / wdt, 1), "ms_per_step": round(wdt * 1e3, 2), "sweep_tiles_needed_frac": 1.0,
                              "inputs": "the headline's crops with 1024 chosen pixels drawn uniformly over the whole 224 x 224 crop"}
        if args.inputs == "crop":
            _, ds = make_inputs(B, device)
            sdt = time_steps(mkstep(net, ds), args.warmup, args.steps)
            sfrac = sweep_tiles_needed_fraction(torch.cat([ds["choose1"][:16], ds["choose2"][:16]]).cpu().numpy())
            legs["survey_masks"] = {"poses_per_sec": round(B / sdt, 1), "ms_per_step": round(sdt * 1e3, 2), "sweep_tiles_needed_frac": round(sfrac, 4),
                                    "inputs": "SURVEY 8d: ellipses with semi-axes 30-90 px anywhere inside the 224 crop, 16 unique poses tiled (the "
                                              "headline workload of rounds 1-3)"}
            del ds
        del du
        torch.cuda.empty_cache()

    # ---- two batches in flight: consecutive steps of a serving loop on two HIP streams, each with its own network instance and
    # workspace (the kernels of one fill the launch tails and quantisation gaps of the other; N = 1 information, not `value`) ----
    two_res = None
    if rank == 0 and world == 1 and not args.no_dense_leg:
        nets2 = [net, AdaPoseNet(sd0, dtype=args.dtype, device=local_rank, max_chunk_views=args.chunk or None)]
        st2 = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]

        last2 = [None, None]
        ref2 = net(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])      # one stream, nothing else on the device
        ref2 = {k: v.clone() for k, v in ref2.items()}

        def pair():
            for i in range(2):
                with torch.cuda.stream(st2[i]):
                    o = nets2[i](d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"], stream=st2[i])
                    postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"], stream=st2[i])
                    last2[i] = o
        torch.cuda.synchronize()
        for s_ in st2:
            s_.wait_stream(torch.cuda.current_stream(device))
        t2 = time_steps(pair, 1, max(2, args.steps // 2)) / 2
        torch.cuda.current_stream(device).wait_stream(st2[0])
        torch.cuda.current_stream(device).wait_stream(st2[1])
        torch.cuda.synchronize()
        # the overlapped forwards against the one-stream forward of the timed step (same inputs): a throughput figure whose outputs differ
        # is not a result.  (Round 4: with hipcc's packed fp32 instructions in the library, forwards that overlapped on the device differed
        # intermittently; it is built without them since — tools/check_two_stream_forwards.py)
        differing = sorted({k for o in last2 for k in ref2 if not torch.equal(o[k].view(torch.int32), ref2[k].view(torch.int32))})
        two_res = {"poses_per_sec": round(B / t2, 1), "ms_per_step": round(t2 * 1e3, 2),
                   "outputs_bit_identical_to_one_stream": not differing, "differing_outputs": differing,
                   "note": "two steps in flight on two HIP streams (two network instances, two workspaces); `value` is the one-stream figure"}
        del last2, ref2
        del nets2
        torch.cuda.empty_cache()

    _mark("accuracy / modes / dense legs done")
    # ---- plugin boundary (the full estimate()-equivalent incl. H2D): AdaPoseEstimator_v5.estimate with numpy frames ----
    boundary_res = None
    if want_boundary and frames is not None:
        from rgbmanip_amd.config import ADAPOSE_CFGS
        from rgbmanip_amd.estimator import AdaPoseEstimator_v5
        ecfg = dict(ADAPOSE_CFGS["adapose_cabinet"], load=False)
        # the estimator as the plugin ships it: its own network with the view-2 heads skipped (estimate() builds the box from the view-1
        # outputs alone, interface_v5.py:318-374)
        est_d = AdaPoseEstimator_v5(None, dict(ecfg, hip_prepare="device"), None, state_dict=sd0, dtype=args.dtype)
        est_full = AdaPoseEstimator_v5(None, dict(ecfg, hip_prepare="device"), None, dtype=args.dtype, net=net)      # all ten outputs computed
        Kh, E1h, E2h = frames["K"].cpu().numpy(), frames["E1"].cpu().numpy(), frames["E2"].cpu().numpy()
        # what rl_pose.py:210-218 hands over: [N,480,640,3] float64 frames and [N,480,640] masks, host numpy
        r1, r2 = frames["rgb1"].cpu().numpy().astype(np.float64), frames["rgb2"].cpu().numpy().astype(np.float64)
        m1, m2 = frames["mask1"].cpu().numpy().astype(np.float64), frames["mask2"].cpu().numpy().astype(np.float64)
        est_d.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            bb = est_d.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        e_s = (time.perf_counter() - t1) / 2
        # the same call without the chunk pipeline (upload everything, then compute), and the upload alone
        est_1 = AdaPoseEstimator_v5(None, dict(ecfg, hip_prepare="device", hip_upload_chunk=0), None, dtype=args.dtype, net=est_d.estimator)
        est_1.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        t1 = time.perf_counter()
        bb1 = est_1.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        e1_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        for _ in range(2):
            ups = [est_1._upload_frames(r1), est_1._upload_masks(m1), est_1._upload_frames(r2), est_1._upload_masks(m2)]
            torch.cuda.synchronize()
        up_s = (time.perf_counter() - t1) / 2
        del ups, est_1
        dev_s = time_steps(lambda: est_d.estimate_device(frames["K"], frames["rgb1"], frames["mask1"], frames["E1"], frames["rgb2"], frames["mask2"],
                                                         frames["E2"]), 1, 3)
        devf_s = time_steps(lambda: est_full.estimate_device(frames["K"], frames["rgb1"], frames["mask1"], frames["E1"], frames["rgb2"], frames["mask2"],
                                                             frames["E2"]), 1, 3)
        bb_full = est_full.estimate_device(frames["K"], frames["rgb1"], frames["mask1"], frames["E1"], frames["rgb2"], frames["mask2"], frames["E2"]).cpu().numpy()
        bb_v1 = est_d.estimate_device(frames["K"], frames["rgb1"], frames["mask1"], frames["E1"], frames["rgb2"], frames["mask2"], frames["E2"]).cpu().numpy()
        r1f, r2f = r1.astype(np.float32), r2.astype(np.float32)
        m1b, m2b = m1 != 0, m2 != 0
        est_d.estimate(Kh, r1f, m1b, E1h, r2f, m2b, E2h)
        t1 = time.perf_counter()
        for _ in range(2):
            est_d.estimate(Kh, r1f, m1b, E1h, r2f, m2b, E2h)
        e32_s = (time.perf_counter() - t1) / 2
        # the reference's own structure (per-frame numpy crop / resize on the host, then one batched forward): bounded sample of 16 poses
        est_h = AdaPoseEstimator_v5(None, dict(ecfg, hip_prepare="host"), None, dtype=args.dtype, net=est_d.estimator)
        nh = min(16, B)
        est_h.estimate(Kh[:2], r1[:2], m1[:2], E1h[:2], r2[:2], m2[:2], E2h[:2])
        t1 = time.perf_counter()
        est_h.estimate(Kh[:nh], r1[:nh], m1[:nh], E1h[:nh], r2[:nh], m2[:nh], E2h[:nh])
        h_s = time.perf_counter() - t1
        # what `python train.py ... pose_estimator=adapose_cabinet` gets with the yaml files unchanged: NO hip_* key, no dtype argument
        # (storage type bf16x3, the mode inside north_star's 1e-4; hip_prepare: device since round 5)
        del est_h
        torch.cuda.empty_cache()
        est_def = AdaPoseEstimator_v5(None, dict(ecfg), None, state_dict=sd0)
        est_def.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            bb_def = est_def.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
        def_s = (time.perf_counter() - t1) / 2
        def_res = {"poses_per_sec": round(B / def_s, 1), "ms_per_call": round(def_s * 1e3, 1), "dtype": est_def.dtype,
                   "hip_prepare": est_def.prepare_mode, "view2_heads": bool(est_def.view2_heads), "finite": bool(np.isfinite(bb_def).all()),
                   "max_abs_box_difference_to_the_bf16_leg_m": float(np.abs(bb_def - bb).max()),
                   "note": "AdaPoseEstimator_v5(env, ADAPOSE_CFGS['adapose_cabinet'], logger) with no hip_* override: the estimate() a "
                           "train.py user gets with the reference's yaml files unchanged"}
        del est_def
        torch.cuda.empty_cache()
        gb = (r1.nbytes + r2.nbytes + m1.nbytes + m2.nbytes) / 1e9
        boundary_res = {
            "call": "AdaPoseEstimator_v5.estimate(K, rgb1, mask1, E1, rgb2, mask2, E2) -> [N,8,3] world boxes, numpy in, numpy out "
                    "(interface_v5.py:213-227 as rl_pose.py:210-218 calls it)", "poses": B, "dtype": args.dtype,
            "frames": f"[{B},480,640,3] float64 + [{B},480,640] float64 masks per view, {gb:.2f} GB of host arrays per call",
            "device_prepare": {"poses_per_sec": round(B / e_s, 1), "ms_per_call": round(e_s * 1e3, 1),
                               "pipeline": "chunks of 32 poses: host threads stage chunk c+1 into pinned memory | copy engine moves chunk c | kernels run chunk c-1",
                               "unpipelined_ms_per_call": round(e1_s * 1e3, 1), "unpipelined_poses_per_sec": round(B / e1_s, 1),
                               "pipelined_close_to_unpipelined": bool(np.allclose(bb, bb1, rtol=2e-2, atol=1e-3)),
                               "pipelined_vs_unpipelined_max_abs_m": float(np.abs(bb - bb1).max()),
                               "upload_alone_ms": round(up_s * 1e3, 1), "upload_alone_host_GBps": round(gb / up_s, 1),
                               "upload_share_of_unpipelined_call": round(up_s / e1_s, 3),
                               "device_resident_ms": round(dev_s * 1e3, 1), "device_resident_poses_per_sec": round(B / dev_s, 1),
                               "device_resident_with_view2_heads_ms": round(devf_s * 1e3, 1),
                               "view1_only_boxes_bit_identical_to_full_forward": bool(np.array_equal(bb_full, bb_v1)),
                               "view2_heads": "skipped (hip_view2_heads default: the box tail reads view-1 outputs only; all ten outputs stay the "
                                              "network API's default and are what `value` times)",
                               "float32_frames_bool_masks": {"poses_per_sec": round(B / e32_s, 1), "ms_per_call": round(e32_s * 1e3, 1)},
                               "note": "hip_prepare: device — float frames are cast to float32 by the copy into pinned double-buffered staging (host thread pool), "
                                       "so float64 frames cross PCIe at half their size; uint8 frames cross as bytes; crop / resize / subset / network / "
                                       "post-processing on the GPU"},
            "default_cfg": def_res,
            "host_prepare": {"poses_per_sec": round(nh / h_s, 2), "ms_per_pose": round(h_s / nh * 1e3, 1), "sample_poses": nh,
                             "note": "hip_prepare: host — the reference's per-frame numpy crop / resize on one host core, then one batched forward"},
            "finite": bool(np.isfinite(bb).all())}
        del r1, r2, m1, m2, r1f, r2f, est_d, est_full
    frames = None
    torch.cuda.empty_cache()

    _mark("boundary leg done")
    # ---- small batches (the deployment path: the reference calls the network per env, num_envs: 8 ships): B = 1 and 8, latency of
    # forward + post-processing, launched one by one and replayed from a hipGraph ----
    peaks_res = achievable_peaks(lib, device) if rank == 0 and not args.no_peaks else None
    small_res = None
    if rank == 0 and world == 1 and not args.no_small_batch:
        small_res = {"what": "median wall-clock latency of one forward + post-processing call incl. the final device synchronisation, "
                             "50 calls per entry; eager = ~150 launches from the host, graph = one hipGraphLaunch of the captured sequence "
                             "(AdaPoseNet(graph=True), rgbm_adapose_forward_graph)", "entries": []}
        for md in dict.fromkeys((args.dtype, "bf16x3")):
            nets = {"eager": AdaPoseNet(sd0, dtype=md, device=local_rank), "graph": AdaPoseNet(sd0, dtype=md, device=local_rank, graph=True)}
            for sb in (1, 8):
                dd = {k: v[:sb].contiguous() for k, v in d.items()}
                ent = {"dtype": md, "batch": sb}
                for how, nn in nets.items():
                    f = mkstep(nn, dd)
                    for _ in range(3):
                        f()
                    torch.cuda.synchronize()
                    lat = []
                    for _ in range(50):
                        t1 = time.perf_counter()
                        f()
                        torch.cuda.synchronize()
                        lat.append(time.perf_counter() - t1)
                    ent[how + "_ms"] = round(float(np.median(lat)) * 1e3, 3)
                    ent[how + "_ms_p90"] = round(float(np.percentile(lat, 90)) * 1e3, 3)
                    # device-side duration of the same call: back-to-back calls, one synchronisation (launch overhead hidden when the GPU is the limit)
                    ent[how + "_pipelined_ms"] = round(time_steps(f, 2, 20) * 1e3, 3)
                ent["graph_nodes"] = nets["graph"].last_graph_nodes
                ent["poses_per_sec_graph"] = round(sb / (ent["graph_pipelined_ms"] * 1e-3), 1)
                small_res["entries"].append(ent)
            del nets
            torch.cuda.empty_cache()

    # ---- SURVEY 8f-1 leg (not part of `value`): device-side prepare_model_input on 480x640 frames, vs the host numpy path ----
    prep_res = None
    if rank == 0 and not args.no_prepare:
        from rgbmanip_amd.adapose import prepare_inputs
        from rgbmanip_amd.estimator import AdaPoseEstimator_v5
        nf = 256
        gen = torch.Generator(device=device).manual_seed(0)
        pframes = torch.rand(nf, 480, 640, 3, generator=gen, device=device)
        yy, xx = torch.meshgrid(torch.arange(480, device=device), torch.arange(640, device=device), indexing="ij")
        cy = 140 + 200 * torch.rand(nf, generator=gen, device=device)
        cx = 160 + 320 * torch.rand(nf, generator=gen, device=device)
        ry = 40 + 80 * torch.rand(nf, generator=gen, device=device)
        rx = 40 + 110 * torch.rand(nf, generator=gen, device=device)
        masks = ((((yy[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 + ((xx[None] - cx[:, None, None]) / rx[:, None, None]) ** 2) < 1).to(torch.uint8)
        Kf = torch.tensor([[439.31, 0, 320.0], [0, 439.31, 240.0], [0, 0, 1.0]], dtype=torch.float64, device=device).repeat(nf, 1, 1)
        prepare_inputs(pframes, masks, Kf, 224, 1024, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for it in range(3):
            po = prepare_inputs(pframes, masks, Kf, 224, 1024, it)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t1) / 3 * 1e3
        est = AdaPoseEstimator_v5.__new__(AdaPoseEstimator_v5)          # host arithmetic only: no network needed
        est.rng = np.random.default_rng(0)
        est._frame = 0
        fh, mh = pframes[:4].cpu().numpy(), masks[:4].cpu().numpy()
        t1 = time.perf_counter()
        for i in range(4):
            est.prepare_model_input(fh[i], mh[i], Kf[0].cpu().numpy(), 224)
        host_ms = (time.perf_counter() - t1) / 4 * 1e3
        prep_res = {"frames": nf, "device_ms_per_frame": round(dev_ms / nf, 4), "device_frames_per_sec": round(nf / dev_ms * 1e3, 1),
                    "host_numpy_ms_per_frame": round(host_ms, 2), "valid_frames": int(po["valid"].sum().item()),
                    "note": "480x640x3 f32 frame + mask -> 224x224 normalised crop, 1024 choose indices, cropped intrinsics (interface_v5.py:58-170)"}
        del pframes, masks

    _mark("prepare leg done")
    # ---- configs[4] leg (not part of `value`): mixed-object batch, 4 heads, sorted by head and sharded over the ranks ----
    mixed_res = None
    if not args.no_mixed:
        from rgbmanip_amd.mixed import HEADS, MixedObjectNet, shard_by_head
        heads_global = np.arange(world * B) % 4                        # interleaved heads, like requests arriving in any order
        idx = shard_by_head(heads_global, rank, world)
        my_heads = heads_global[idx]
        mnet = MixedObjectNet({h: synth.adapose_state_dict(seed=h) for h in np.unique(my_heads).tolist()}, dtype=args.mixed_dtype,
                              device=local_rank, max_chunk_views=args.chunk or None)
        margs = (d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])

        def mstep():
            o = mnet(my_heads, *margs)
            return postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"])
        mstep()
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            mstep()
        barrier()
        mdt = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([mdt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mdt = float(tt.item())
        mixed_res = {"poses_per_sec": round(world * B * 3 / mdt, 1), "heads": list(HEADS), "batch_total": world * B,
                     "heads_on_rank0": sorted(set(int(h) for h in my_heads)),
                     "dtype": args.mixed_dtype,
                     "head_streams": bool(mnet.head_streams),
                     "note": "BASELINE configs[4] layout: batch sorted by head, contiguous shard per rank, one AdaPoseNet per head, every head's samples on "
                             "that head's own stream (MixedObjectNet(head_streams=True), bit-identical to one head after the other)"}
        del mnet

    _mark("mixed leg done")
    # ---- PPO leg: AdaPose-in-the-loop rollout (synthetic vec-env stand-in) + HIP learn phase, cfg/controller/rl.yaml ----
    ppo_res = None
    if args.ppo_envs > 0:
        from rgbmanip_amd.config import rl_cfg
        from rgbmanip_amd.ppo import PPO
        cfg = rl_cfg(task="cabinet", device=str(device), print_log=False, log_dir="/tmp/rgbm_bench_logs", save_dir="/tmp/rgbm_bench_saves")
        if args.ppo_env == "full":
            # the reference's loop: ControlInterface.step over a (synthetic) MultiVecEnv partition of this rank — camera move,
            # 480x640 render, view queue, prepare_model_input, AdaPose, post-processing, reward — all on the device
            from rgbmanip_amd.config import ADAPOSE_CFGS
            from rgbmanip_amd.control_interface import ControlInterface
            from rgbmanip_amd.estimator import AdaPoseEstimator_v5
            from rgbmanip_amd.synthetic_env import SyntheticManipulation, SyntheticMultiVecEnv
            # the estimator builds its own network: view-2 heads skipped, as the plugin ships (ControlInterface reads the box only)
            est = AdaPoseEstimator_v5(None, dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device"), None,
                                      state_dict=sd0, dtype=args.dtype, device=local_rank)
            venv = SyntheticMultiVecEnv(args.ppo_envs, device, seed=0, env_id_offset=rank * args.ppo_envs)
            env = ControlInterface(venv, est, SyntheticManipulation(venv), cfg, device=device)
            env_name = ("ControlInterface.step over SyntheticMultiVecEnv: render 480x640 -> view queue -> prepare_model_input -> "
                        "AdaPose -> bbox -> 14-term reward, per env step, all on the device")
        else:
            from rgbmanip_amd.synthetic_env import SyntheticPoseVecEnv
            env = SyntheticPoseVecEnv(args.ppo_envs, net, device, seed=0, rank=rank)
            env_name = "SyntheticPoseVecEnv (pre-cropped view bank; one batched AdaPose estimate per env step)"
        ppo = PPO(env, cfg)
        ppo.run(args.ppo_iters, log_interval=1, save_interval=10 ** 9)
        barrier()
        ppo_res = {"env_steps_per_sec": round(ppo.last_fps, 1), "num_envs_per_gpu": args.ppo_envs, "transitions_per_env": 16,
                   "collection_s": round(ppo.last_collection_time, 3), "learn_s": round(ppo.last_learn_time, 4),
                   "optimizer_steps": 32, "env": env_name,
                   "estimator_view2_heads": (bool(est.view2_heads) if args.ppo_env == "full" else True),
                   "lr_after": ppo.step_size}
        prim = (ppo.last_collection_time, ppo.last_learn_time, ppo.last_fps)
        dflt = None
        if args.ppo_env == "full" and not args.no_ppo_default:
            # the same loop with the estimator a train.py user gets from the unchanged yaml files: no hip_* key, no dtype argument
            # (bf16x3 storage, device-side prepare_model_input)
            del ppo, env, est
            torch.cuda.empty_cache()
            est = AdaPoseEstimator_v5(None, dict(ADAPOSE_CFGS["adapose_cabinet"], load=False), None, state_dict=sd0, device=local_rank)
            venv = SyntheticMultiVecEnv(args.ppo_envs, device, seed=0, env_id_offset=rank * args.ppo_envs)
            env = ControlInterface(venv, est, SyntheticManipulation(venv), cfg, device=device)
            ppo = PPO(env, cfg)
            ppo.run(args.ppo_iters, log_interval=1, save_interval=10 ** 9)
            barrier()
            ppo_res["default_cfg"] = {"env_steps_per_sec": round(ppo.last_fps, 1), "collection_s": round(ppo.last_collection_time, 3),
                                      "learn_s": round(ppo.last_learn_time, 4), "estimator_dtype": est.dtype, "hip_prepare": est.prepare_mode,
                                      "estimator_view2_heads": bool(est.view2_heads),
                                      "note": "pose_estimator=adapose_cabinet controller=rl with no hip_* key in the yaml files"}
            dflt = (ppo.last_collection_time, ppo.last_learn_time, ppo.last_fps)
        if dist is not None:      # per-rank collection / learn times: a straggler of the PPO loop is visible the day SCALE runs
            allr = torch.zeros(world, 6, dtype=torch.float64, device=device)
            allr[rank] = torch.tensor(list(prim) + list(dflt or (0.0, 0.0, 0.0)), dtype=torch.float64, device=device)
            dist.all_reduce(allr)
            ppo_res["per_rank"] = {"collection_s": [round(float(a[0]), 3) for a in allr], "learn_s": [round(float(a[1]), 4) for a in allr]}
            ppo_res["env_steps_per_sec_all_ranks"] = round(world * args.ppo_envs * 16 / max(float(a[0] + a[1]) for a in allr), 1)
            if dflt is not None:
                ppo_res["default_cfg"]["per_rank"] = {"collection_s": [round(float(a[3]), 3) for a in allr], "learn_s": [round(float(a[4]), 4) for a in allr]}
                ppo_res["default_cfg"]["env_steps_per_sec_all_ranks"] = round(world * args.ppo_envs * 16 / max(float(a[3] + a[4]) for a in allr), 1)

    _mark("ppo leg done")
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B / (elapsed / args.steps)
        # the dominant kernel from the timed steps, every other row from the untimed profiled step in front of them
        dom = prof_table(stats, args.steps)[0]
        assert dom["row"] == dom_row
        kernels = [dom] + [k for k in prof_table(stats_all, 1) if k["row"] != dom_row]
        peak = PEAK_TFLOPS[dom["dtype"]]
        # HBM bytes per launch of the dominant kernel: PMC passes cannot run inside this process, so the figure is read
        # from the committed rocprofv3 --pmc measurement of this same command (tools/pmc_traffic.py -> profiles/)
        traffic, traffic_src, hbm_step = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json" if args.dtype == "bf16" else f"hbm_traffic_{args.dtype}.json")
        if os.path.exists(tpath) and B == 256:
            tj = json.load(open(tpath))
            meta = tj.get("_meta", {})
            # a PMC measurement describes the tree it was taken on: a stale file is refused, not silently reused
            if meta.get("tree") == tree_hash() and meta.get("dtype") == args.dtype:
                key = dom["kernel"].split(" (")[0].rstrip(">")       # kernel name incl. template arguments as rocprofv3 prints it
                for kname, nbytes in tj["bytes_per_launch"].items():
                    if key in kname:
                        traffic = float(nbytes)
                        traffic_src = (f"profiles/{os.path.basename(tpath)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this tree (2*FETCH+WRITE), "
                                       f"taken by the builder on another box of the pool ({meta.get('box', 'box not recorded')}), not in this run")
                        break
                hbm_step = meta.get("hbm_bytes_per_step")
            else:
                traffic_src = f"{os.path.basename(tpath)} is for tree {meta.get('tree')} / {meta.get('dtype')}, this is {tree_hash()} / {args.dtype}: not used"
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(dom["tflops"], 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": round(dom["avg_launch_ms"], 4), "launches_per_step": dom["launches_per_step"],
                    "flops_per_launch": dom["tflops"] * 1e12 * dom["avg_launch_ms"] * 1e-3}
        # the implicit GEMM that carries layer3 / layer4 / up_1 (rows 31 / 33: the dominant kernel of rounds 1-4; since round 5 the plane
        # sweep and this GEMM are within a few per cent of each other in total time, so whichever is `roofline`, this one is always reported)
        gemm_row = 33 if args.dtype == "bf16x3" else 31
        gk = [k for k in kernels if k["row"] == gemm_row]
        roofline_gemm = None
        if gk:
            gpeak = PEAK_TFLOPS[gk[0]["dtype"]]
            roofline_gemm = {"bound": "mfma", "kernel": gk[0]["kernel"], "achieved": round(gk[0]["tflops"], 2), "peak": gpeak, "unit": "TFLOP/s",
                             "frac": round(gk[0]["tflops"] / gpeak, 4), "avg_launch_ms": round(gk[0]["avg_launch_ms"], 4),
                             "launches_per_step": gk[0]["launches_per_step"],
                             "measured_in": "the timed steps" if gemm_row == dom_row else "the untimed profiled step in front of the timed ones"}
        inputs_desc = {
            "crop": "480x640 synthetic frames with an elliptical object mask (semi-axes 40-120 x 40-150 px) -> the reference's crop window "
                    "(lib/utils.py:10-38) -> 224x224 crop + 1024 chosen pixels on the device (interface_v5.py:58-170): masks span their crops",
            "survey": "SURVEY 8d: 1024 chosen pixels inside an elliptical mask of 5-50 % of the 224 crop, 16 unique poses tiled",
            "uniform": "1024 chosen pixels uniform over the whole crop (worst case of the sparse cost regularisation)"}[args.inputs]
        res = {
            "metric": "adapose_poses_per_sec_batch256", "value": round(value, 3), "unit": "poses/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"adapose_cabinet forward + post-processing, batch={B} poses ({2 * B} views of 224x224) per GPU, "
                                   "synthetic RGB, random-init weights of the reference architecture; inputs: " + inputs_desc,
                       "poses_per_gpu": B, "unique_poses": n_unique, "outputs": "all 10 network outputs + world bbox", "parallelism": f"dp{world}",
                       "inputs": args.inputs,
                       "cost_regularisation": ("library default (sparse_dec = 2): computed only where the chosen pixels' outputs depend on it, all ten "
                                               "outputs bit-identical to the dense computation; value_dense / value_worst_case of the same run are "
                                               "top-level keys" if args.dtype != "fp32" else "dense"),
                       **({"storage_detail": "bf16 net with sweep_f16 = 1 (library default): the 32-channel feature map `final` writes, the conv0 weights of "
                                             "the plane sweep and the y / z of the one-kernel PSPNet tail are IEEE f16 (saturating at +-65504), f16 MFMAs in the "
                                             "sweep; every other tensor is bf16; hip_options {sweep_f16: 0} gives the all-bf16 net",
                           "sweep_f16": int(net.options.get("sweep_f16", 1))} if args.dtype == "bf16" else {}),
                       **({"debug_flags": args.debug_flags} if args.debug_flags else {})},
            "world_size": (dist.get_world_size() if dist is not None else 1), "dist_backend": (dist.get_backend() if dist is not None else None),
            "tree": tree_hash(),
            "sweep_tiles_needed_frac": round(my_frac, 4),
            # ALGORITHMIC flops of the reference's dense forward per second over the MFMA peak: a SPEED figure, not a utilisation (the
            # commuted PSPUpsample and the sparse cost regularisation do not execute all of them); the executed-flop fraction is
            # mfma_frac_executed_flops_dense
            "algorithmic_tflops": round(value * GFLOP_PER_POSE / 1e3, 2),
            "algorithmic_tflops_over_peak": round(value * GFLOP_PER_POSE / 1e3 / world / PEAK_TFLOPS[args.dtype], 4),
            "valid_poses_last_step": n_valid, "outputs_finite": finite,
            "roofline": roofline, "roofline_gemm": roofline_gemm, "conv_kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items() if k != "flops_per_step"} for kk in kernels],
        }
        if legs is not None:
            res["value_dense"] = legs["dense"]["poses_per_sec"]
            res["ms_per_step_dense"] = legs["dense"]["ms_per_step"]
            res["mfma_frac_executed_flops_dense"] = legs["dense"]["mfma_frac_executed_flops"]
            res["value_worst_case"] = legs["worst_case"]["poses_per_sec"]
            if "survey_masks" in legs:
                res["value_survey_masks"] = legs["survey_masks"]["poses_per_sec"]
            res["sparse_cost_regularisation"] = {
                "enabled": True, **legs,
                "note": "exact: outputs are bit-identical to the dense computation (tests/test_gpu_at_batch.py, poisoned workspace); data "
                        "dependent: `value` is on masks that span their crops like the reference's, value_worst_case needs every tile, "
                        "value_survey_masks is SURVEY 8d's ellipses of 5-50 % of the crop"}
        if two_res is not None:
            res["value_two_streams"] = two_res["poses_per_sec"]
            res["two_streams"] = two_res
        if per_rank is not None:
            res["per_rank"] = per_rank
        # whole-net HBM rate: measured bytes per step (PMC, every kernel of one forward) against the algorithmic minimum of
        # SURVEY 8(d) (0.54 GB per pose at 2 bytes per element: every conv reads its input once and writes its output once)
        algo_gb_pose = 0.54 * (2.0 if args.dtype in ("fp32", "bf16x3") else 1.0)
        res["whole_net_hbm"] = {"algorithmic_GBps": round(algo_gb_pose * value / world, 1), "algorithmic_GB_per_pose": algo_gb_pose,
                                "measured_GBps": round(hbm_step / (ms_per_step * 1e-3) / 1e9, 1) if hbm_step else None,
                                "measured_GB_per_step": round(hbm_step / 1e9, 2) if hbm_step else None, "peak_GBps": 8000.0}
        res["accuracy"] = acc_res
        if modes_res is not None:
            res["modes"] = modes_res
        res["timed_region_note"] = ("inside the timed region only the dominant kernel's launches sit between two HIP events (rgbm_prof_select); the "
                                    "other rows of conv_kernels come from one untimed step in front of it.  The headline includes "
                                    "that overhead" + ("" if n_unique == B else f"; inputs are {n_unique} unique poses tiled to the batch (no dedupe exists in the library)"))
        res["frames_per_sec"] = round(2 * value, 1)      # one pose = one stereo pair = two 224 x 224 frames
        if peaks_res is not None:
            res["achievable_peaks"] = peaks_res
            rnd = peaks_res["mfma_bf16_random_operands_TFLOPs"] / (3.0 if dom["dtype"] == "bf16x3" else 1.0)      # split pairs: three MFMAs per product
            if roofline["unit"] == "TFLOP/s" and dom["dtype"] in ("bf16", "bf16x3") and rnd > 0:
                roofline["frac_of_achievable"] = round(roofline["achieved"] / rnd, 4)
                roofline["achievable_peak"] = round(rnd, 1)
        if boundary_res is not None:
            res["plugin_boundary"] = boundary_res
            res["value_plugin_boundary"] = boundary_res["device_prepare"]["poses_per_sec"]
        if small_res is not None:
            res["small_batch"] = small_res
        if ppo_res is not None:
            res["ppo"] = ppo_res
        if prep_res is not None:
            res["prepare_model_input"] = prep_res
        if mixed_res is not None:
            res["mixed_object"] = mixed_res
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], oref = cpu_baseline(host, n_unique=n_unique)
            res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
            # the timed batch itself against the oracle (the CPU baseline's outputs are the checker): B = 2 goldens never reach the
            # kernels the dispatcher picks at batch 256
            if acc_res is not None:
                acc_res["at_batch"] = at_batch_accuracy(batch_outs[args.dtype], oref, B, n_unique)
                acc_res["at_batch"]["workspace_poisoned"] = poison_same is not None
                acc_res["at_batch"]["poisoned_run_bit_identical_to_timed_step"] = poison_same
            for md, mr in (modes_res or {}).items():
                mr["accuracy"]["at_batch"] = at_batch_accuracy(batch_outs[md], oref, B, n_unique)
                mr["accuracy"]["at_batch"]["workspace_poisoned"] = True
            if ppo_res is not None:
                res["ppo"]["cpu_baseline"] = cpu_baseline_ppo()
        # the fastest mode of this run whose outputs meet north_star's 1e-4 (on the reference's golden vectors AND, when the
        # oracle ran, inside the timed batch): the number that satisfies ">= 10x CPU with pose error <= 1e-4" by itself
        cands = [(args.dtype, value, acc_res)] + [(md, mr["poses_per_sec"], mr["accuracy"]) for md, mr in (modes_res or {}).items()]
        ok = [(v, md) for md, v, a in cands if a and a.get("meets_1e-4") and a.get("at_batch", {"meets_1e-4": True})["meets_1e-4"]]
        res["value_within_tolerance"] = round(max(ok)[0], 3) if ok else None
        res["dtype_within_tolerance"] = max(ok)[1] if ok else None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
