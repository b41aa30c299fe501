"""-m gpu: the trainer's real multi-rank path — `RolloutStorage.compute_returns` and `PPO.update` of two
ranks that each own half of the environments must leave every rank with the parameters a single process gets from all
of them.  The box has one GPU: both ranks use cuda:0 and exchange through gloo (RCCL refuses two ranks on one device);
the kernels and the host logic are exactly those of an 8-GPU run, only the transport differs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, N_TOTAL = 16, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_once(lo, hi):
    """One compute_returns + update on envs [lo, hi) of the recorded rollout (the fill of test_gpu_ppo.py / make_goldens.py)."""
    from test_gpu_ppo import CFG, FakeEnv
    from rgbmanip_amd import synth
    from rgbmanip_amd.ppo import PPO
    ppo = PPO(FakeEnv(hi - lo), CFG)
    ppo.actor_critic.load_state_dict({k: torch.from_numpy(v) for k, v in synth.policy_state_dict(seed=0).items()})
    roll = synth.ppo_rollout(T, N_TOTAL, seed=0)
    tr = {k: torch.from_numpy(v).cuda() for k, v in roll.items()}
    ac = ppo.actor_critic
    for t in range(T):
        obs, act = tr["observations"][t][lo:hi].contiguous(), tr["actions"][t][lo:hi].contiguous()
        lp, _, _, mm, ss, _ = ac.evaluate(obs, None, act)
        mm = mm + 0.02 * torch.sin(torch.arange(12.0)).cuda()[None]
        ppo.storage.add_transitions(obs, tr["states"][t][lo:hi], act, tr["rewards"][t].view(-1)[lo:hi], tr["dones"][t].view(-1)[lo:hi],
                                    tr["values"][t][lo:hi], lp - 0.01, mm, ss - 0.005)
    ppo.storage.compute_returns(tr["last_values"][lo:hi], 0.98, 0.98)
    mvl, msl = ppo.update(0)
    torch.cuda.synchronize()
    return dict(flat=ac.flat.cpu().numpy(), adv=ppo.storage.advantages.cpu().numpy(), mvl=mvl, msl=msl, lr=ppo.step_size,
                world=ppo.world)


def _worker(rank, world, port, q, backend="gloo"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if backend == "nccl":                      # RCCL: one device per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_TOTAL // world
    res = _train_once(rank * per, (rank + 1) * per)
    res["rank"] = rank
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _two_rank_update(backend):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = _train_once(0, N_TOTAL)                                   # this process: no process group, all 64 envs
    assert one["world"] == 1 and all(r["world"] == 2 for r in res)
    # advantages: each rank's slice of the globally normalised vector (storage.py:63-64 over all T*N values)
    adv = np.concatenate([r["adv"] for r in res], axis=1)
    np.testing.assert_allclose(adv, one["adv"], rtol=1e-5, atol=1e-6)
    # every rank ends with the same parameters, equal to the single-process ones; losses and the adaptive LR agree
    np.testing.assert_array_equal(res[0]["flat"], res[1]["flat"])
    scale = np.abs(one["flat"]).max()
    assert np.abs(res[0]["flat"] - one["flat"]).max() / scale < 1e-5
    for r in res:
        assert abs(r["mvl"] - one["mvl"]) < 1e-5 * abs(one["mvl"]) and abs(r["msl"] - one["msl"]) < 1e-5 + 1e-4 * abs(one["msl"])
        assert r["lr"] == one["lr"]


def test_two_ranks_reproduce_the_single_process_update():
    _two_rank_update("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL wants one device per rank: needs >= 2 GPUs")
def test_two_ranks_over_rccl_reproduce_the_single_process_update():
    """The same check with `backend="nccl"` (RCCL over xGMI), one rank per device — armed for boxes with >= 2 GPUs; the
    advantage-statistics all-reduce and the flat-gradient all-reduce are then the real collectives of an 8-GPU run."""
    _two_rank_update("nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_bench_two_gpus_smoke():
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one rank per GPU, RCCL): the line must
    report both ranks and the nccl backend, and the weak-scaling value must cover both shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "32", "--no-modes", "--ppo-envs", "32", "--ppo-iters", "1", "--no-prepare"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["dist_backend"] == "nccl"
    assert line["config"]["parallelism"] == "dp2" and line["value"] > 0 and line["ppo"]["env_steps_per_sec"] > 0


def test_bench_two_ranks_rehearsal_on_one_device():
    """The multi-rank code path of bench.py (rank-local inputs, max-over-ranks timing, the per-rank ms / tile-fraction table, the PPO
    leg's per-rank times) on this box's ONE GPU: two ranks share cuda:0 and talk over gloo (RGBM_BENCH_ONE_DEVICE=1,
    RGBM_DIST_BACKEND=gloo) — launched exactly as the driver launches `--gpus 2`.  RCCL itself needs one device per rank
    (test_bench_two_gpus_smoke, armed for multi-GPU boxes)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RGBM_BENCH_ONE_DEVICE="1", RGBM_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "16", "--no-modes", "--ppo-envs", "16", "--ppo-iters", "1", "--no-prepare", "--no-mixed"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["dist_backend"] == "gloo" and line["scaling"] == "weak"
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["sweep_tiles_needed_frac"]) == 2 and pr["ms_per_step_min"] <= pr["ms_per_step_max"]
    assert abs(line["ms_per_step"] - pr["ms_per_step_max"]) / line["ms_per_step"] < 0.25      # `value` is priced on the slowest rank
    assert pr["sweep_tiles_needed_frac"][0] != pr["sweep_tiles_needed_frac"][1]               # rank-local inputs (seeded by rank)
    assert len(line["ppo"]["per_rank"]["collection_s"]) == 2 and line["ppo"]["env_steps_per_sec_all_ranks"] > 0


def test_bench_plain_launch_spawns_its_ranks_on_one_device():
    """`python bench.py --gpus 2` with NO launcher in front (the shape of the driver's N = 1 command): bench.py starts
    `torch.distributed.run --nproc-per-node 2 bench.py <same args>` as a child process before it touches the GPU, relays rank 0's JSON
    line and returns the child's exit code (round-5 verdict item 6).  Rehearsed on this box's one GPU over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", RGBM_BENCH_ONE_DEVICE="1", RGBM_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "16", "--no-modes", "--ppo-envs", "0", "--no-prepare", "--no-mixed"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                             # ONE JSON line: rank 0's, relayed
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    # and the exit code is the child's: an argument the ranks refuse must fail the parent too
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, RGBM_HIP_LIB="/nonexistent/librgbm_hip.so"), cwd=root)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_single_process_line_has_every_leg():
    """`python bench.py` as the driver runs it at N = 1, shrunk (batch 16, 1 step): every leg of the line must come back — a crash in
    one of them at round end would lose the whole bench record.  Checks the keys the round-3 verdict asked for (dense / worst-case /
    survey legs promoted to top level, the renamed algorithmic figure, plugin boundary, small batches) and basic sanity of each."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--batch", "16", "--steps", "1", "--warmup", "1", "--ppo-envs", "16", "--ppo-iters", "1",
           "--mode-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["metric"] == "adapose_poses_per_sec_batch256" and line["n_gpus"] == 1 and line["value"] > 0 and line["dtype"] == "bf16"
    assert line["config"]["inputs"] == "crop" and 0.0 < line["sweep_tiles_needed_frac"] <= 1.0
    for k in ("value_dense", "value_worst_case", "value_survey_masks", "value_two_streams", "mfma_frac_executed_flops_dense",
              "algorithmic_tflops_over_peak", "value_plugin_boundary", "value_within_tolerance"):
        assert line[k] is not None and line[k] > 0, k
    assert "whole_net_frac_of_mfma_peak" not in line                                  # renamed: a speed figure, not a utilisation
    rl = line["roofline"]
    assert rl["bound"] == "mfma" and rl["peak"] == 2500.0 and 0 < rl["frac"] < 1 and "traffic" in rl
    assert not any(k["kernel"] == "unused" for k in line["conv_kernels"])
    assert isinstance(line["two_streams"]["outputs_bit_identical_to_one_stream"], bool)   # checked and reported, whichever way
    assert line["frames_per_sec"] == pytest.approx(2 * line["value"], rel=1e-3)
    pk = line["achievable_peaks"]                                                      # peaks measured on this box, as extra denominators
    assert 1000 < pk["mfma_bf16_random_operands_TFLOPs"] <= pk["mfma_bf16_constant_operands_TFLOPs"] * 1.02 < 2700
    assert 2000 < pk["hbm_copy_GBps_read_plus_write"] < 8000
    assert rl["achievable_peak"] == pk["mfma_bf16_random_operands_TFLOPs"] and rl["frac"] < rl["frac_of_achievable"] < 1
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    ab = line["accuracy"]["at_batch"]
    assert ab["workspace_poisoned"] and ab["poisoned_run_bit_identical_to_timed_step"] is True
    pb = line["plugin_boundary"]["device_prepare"]
    assert pb["pipelined_close_to_unpipelined"] and pb["view1_only_boxes_bit_identical_to_full_forward"] and line["plugin_boundary"]["finite"]
    dc = line["plugin_boundary"]["default_cfg"]                                        # round 5: the estimator a train.py user gets from the unchanged yaml files
    assert dc["poses_per_sec"] > 0 and dc["dtype"] == "bf16x3" and dc["hip_prepare"] == "device" and dc["finite"]
    pd_ = line["ppo"]["default_cfg"]
    assert pd_["env_steps_per_sec"] > 0 and pd_["estimator_dtype"] == "bf16x3" and pd_["hip_prepare"] == "device"
    rg = line["roofline_gemm"]                                                         # the implicit GEMM of layer3 / layer4 / up_1, whichever kernel is dominant by time
    assert "conv_igemm" in rg["kernel"] and 0 < rg["frac"] < 1 and rg["avg_launch_ms"] > 0
    sm = line["small_batch"]["entries"]
    assert {(e["dtype"], e["batch"]) for e in sm} == {("bf16", 1), ("bf16", 8), ("bf16x3", 1), ("bf16x3", 8)}
    assert all(e["eager_ms"] > 0 and e["graph_ms"] > 0 and e["graph_nodes"] > 50 for e in sm)
    assert line["modes"]["bf16x3"]["accuracy"]["at_batch"]["meets_1e-4"] and line["dtype_within_tolerance"] == "bf16x3"
    assert line["ppo"]["env_steps_per_sec"] > 0 and line["ppo"]["estimator_view2_heads"] is False
    assert line["mixed_object"]["poses_per_sec"] > 0 and line["prepare_model_input"]["valid_frames"] > 0
