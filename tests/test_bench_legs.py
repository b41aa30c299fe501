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
