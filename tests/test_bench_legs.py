"""The CPU legs of bench.py run on their own (no GPU): a crash in one of them loses the whole bench line at round end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_baseline_ppo_leg_runs():
    import bench
    r = bench.cpu_baseline_ppo(n_envs=1)
    assert r["unit"] == "env-steps/s" and r["kind"] == "port" and r["value"] > 0 and r["learn_s"] > 0


def test_cpu_baseline_leg_runs():
    """the leg runs on the benched batch's own poses and hands the oracle outputs back for `accuracy.at_batch`"""
    import numpy as np
    import torch
    import bench
    from rgbmanip_amd import synth
    host = synth.adapose_inputs(3, seed=0)
    r, ref = bench.cpu_baseline(host, n_chunks=1, chunk=1)
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1
    assert sorted(ref) == [1, 2] and set(ref[1]) == set(bench.OUT_KEYS)
    # the checker compares what a device run would have left at those batch positions: feed it the oracle's own numbers, tiled
    B = 7
    dev = {k: torch.from_numpy(np.stack([ref[[1, 2][b % 2]][k] for b in range(B)])) for k in bench.OUT_KEYS}
    fake_ref = {0: ref[1], 1: ref[2]}
    acc = bench.at_batch_accuracy(dev, fake_ref, B, n_unique=2)
    assert acc["meets_1e-4"] and acc["worst_output_rel_err"] == 0.0 and acc["positions"] == 4
    dev["view1_depth"][6] += 1.0                      # last replica of unique pose 0 is position 6
    assert not bench.at_batch_accuracy(dev, fake_ref, B, n_unique=2)["meets_1e-4"]


def test_tree_hash_is_stable_and_ignores_profiles(tmp_path):
    import bench
    assert bench.tree_hash() == bench.tree_hash() and len(bench.tree_hash()) >= 12


def test_plain_multi_gpu_launch_starts_its_ranks_as_children():
    """`python bench.py --gpus 2` without a launcher must not die with a usage message (round-5 verdict item 6): it starts
    torch.distributed.run as a child before any GPU call, relays the ranks' output and returns their exit code.  On this GPU-less
    box both ranks stop at bench.py's own "needs a GPU" assertion — which proves they were started with WORLD_SIZE = 2."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: tests/test_gpu_dist.py runs the full rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode != 0                                            # the children's failure is the parent's exit code
    assert "needs `python -m torch.distributed.run" not in (r.stderr + r.stdout)
    assert r.stderr.count("bench.py needs a GPU") >= 2                  # both ranks ran bench.py's main()
