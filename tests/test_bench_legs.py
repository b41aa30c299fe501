"""The CPU legs of bench.py run on their own (no GPU): a crash in one of them loses the whole bench line at round end."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_baseline_ppo_leg_runs():
    import bench
    r = bench.cpu_baseline_ppo(n_envs=1)
    assert r["unit"] == "env-steps/s" and r["kind"] == "port" and r["value"] > 0 and r["learn_s"] > 0


def test_cpu_baseline_leg_runs():
    import bench
    r = bench.cpu_baseline(n_chunks=1, chunk=1)
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1


def test_tree_hash_is_stable_and_ignores_profiles(tmp_path):
    import bench
    assert bench.tree_hash() == bench.tree_hash() and len(bench.tree_hash()) >= 12
