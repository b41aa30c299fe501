"""Mixed-object AdaPose batches (BASELINE.json configs[4], SURVEY.md §8d/e "cfg-5"): cabinet / drawer / mug / pot estimators
share one architecture and differ only in their checkpoint (`cfg/pose_estimator/adapose_*.yaml`: `task_name`,
`checkpoint_path`), so a mixed batch is sorted by head, split contiguously over the ranks (each rank then touches as few
weight sets as possible: 2048 poses = 4 x 512 over 8 GPUs -> one head per GPU) and every head's run of samples goes through
that head's `AdaPoseNet`.  The reference has no such batching (it serves one estimator per process, one pose per call); this
is the data-parallel layout the north star asks for on top of the same per-pose arithmetic, so results per pose are exactly
what the single-head estimator returns."""
from __future__ import annotations

import numpy as np
import torch

from .adapose import AdaPoseNet
from .dist_utils import shard_range

HEADS = ("adapose_cabinet", "adapose_drawer", "adapose_mug", "adapose_pot")


def shard_by_head(head_ids, rank: int = 0, world: int = 1):
    """Indices (into the batch) of the samples rank `rank` processes: the batch stably sorted by head id, cut into `world`
    contiguous blocks.  Every sample belongs to exactly one rank; a rank sees at most ceil(n_heads / world) + 1 heads."""
    head_ids = np.asarray(head_ids)
    order = np.argsort(head_ids, kind="stable")
    lo, hi = shard_range(len(order), rank, world)
    return order[lo:hi]


class MixedObjectNet:
    def __init__(self, state_dicts: dict, dtype: str = "bf16", device: int = 0, head_streams: bool = True, **net_kw):
        """state_dicts: head id (int or name) -> state_dict.  Networks are built on first use, so a rank only ever holds the
        weight sets its shard needs (25 M parameters = 50 MB in bf16 each).
        head_streams: every head's run of samples goes to its own HIP stream (forked from / joined to the caller's stream): the
        heads are independent networks with their own workspaces, and a head's share of a mixed batch is a fraction of a
        full batch, whose launches leave CUs idle (partial last rounds of the GEMM tiles, one-tile launches) that another
        head's kernels can take.  The launch sequence of a head is the same either way: per-pose results are bit-identical
        (tests/test_gpu_adapose.py::test_mixed_object_batch_equals_per_head_runs).  Measured at 4 x 64 poses (tools/mixed_ab.py, one box):
        fp16 50.7 -> 45.6 ms, bf16 45.6 -> 41.1 ms per mixed batch of 256.  (One network's batch cut into equal parts on as many
        streams does not gain: tools/split_parts_ab.py, -1 % with two parts, +3 % with four.)"""
        self.state_dicts, self.dtype, self.device, self.net_kw = dict(state_dicts), dtype, device, net_kw
        self.nets = {}
        self.head_streams = bool(head_streams)
        self._streams = {}

    def net(self, head):
        if head not in self.nets:
            self.nets[head] = AdaPoseNet(self.state_dicts[head], dtype=self.dtype, device=self.device, **self.net_kw)
        return self.nets[head]

    def __call__(self, head_ids, view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values):
        """Forward of a (local) mixed batch; outputs come back in the order of the inputs."""
        head_ids = np.asarray(head_ids)
        args = [torch.as_tensor(a) for a in (view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values)]
        heads = np.unique(head_ids)
        if self.head_streams and len(heads) > 1:
            return self._call_on_streams(head_ids, heads, args)
        out = None
        for head in heads:
            sel = torch.from_numpy(np.nonzero(head_ids == head)[0])
            res = self.net(head.item() if hasattr(head, "item") else head)(*[a[sel.to(a.device)] for a in args])
            if out is None:
                out = {k: torch.empty((len(head_ids),) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for k, v in res.items()}
            for k, v in res.items():
                out[k][sel.to(v.device)] = v
        return out

    def _call_on_streams(self, head_ids, heads, args):
        dev = torch.device("cuda", self.device)
        cur = torch.cuda.current_stream(dev)
        args = [a.to(dev) for a in args]
        n = len(head_ids)
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"view1_nocs": torch.empty(n, 1024, 3, **f32), "view2_nocs": torch.empty(n, 1024, 3, **f32),
               "view1_depth": torch.empty(n, 1024, **f32), "view2_depth": torch.empty(n, 1024, **f32),
               "view1_r": torch.empty(n, 3, 3, **f32), "view2_r": torch.empty(n, 3, 3, **f32),
               "view1_t": torch.empty(n, 3, **f32), "view2_t": torch.empty(n, 3, **f32),
               "view1_s": torch.empty(n, 3, **f32), "view2_s": torch.empty(n, 3, **f32)}
        fork = torch.cuda.Event()
        fork.record(cur)                                   # the inputs (and `out`'s memory) are ordered on the caller's stream
        joins = []
        for head in heads:
            key = head.item() if hasattr(head, "item") else head
            net = self.net(key)
            st = self._streams.get(key)
            if st is None:
                st = self._streams[key] = torch.cuda.Stream(device=dev)
            st.wait_event(fork)
            with torch.cuda.stream(st):
                sel = torch.from_numpy(np.nonzero(head_ids == head)[0]).to(dev)      # (uploaded on the head's stream, like everything that reads it)
                res = net(*[a[sel] for a in args])
                assert set(res) == set(out), sorted(res)
                for k, v in res.items():
                    out[k][sel] = v
                    v.record_stream(st)
                j = torch.cuda.Event()
                j.record(st)
            joins.append(j)
        for j in joins:
            cur.wait_event(j)
        return out
