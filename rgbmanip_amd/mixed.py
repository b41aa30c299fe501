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
    def __init__(self, state_dicts: dict, dtype: str = "bf16", device: int = 0, **net_kw):
        """state_dicts: head id (int or name) -> state_dict.  Networks are built on first use, so a rank only ever holds the
        weight sets its shard needs (25 M parameters = 50 MB in bf16 each)."""
        self.state_dicts, self.dtype, self.device, self.net_kw = dict(state_dicts), dtype, device, net_kw
        self.nets = {}

    def net(self, head):
        if head not in self.nets:
            self.nets[head] = AdaPoseNet(self.state_dicts[head], dtype=self.dtype, device=self.device, **self.net_kw)
        return self.nets[head]

    def __call__(self, head_ids, view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values):
        """Forward of a (local) mixed batch; outputs come back in the order of the inputs."""
        head_ids = np.asarray(head_ids)
        args = [torch.as_tensor(a) for a in (view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values)]
        out = None
        for head in np.unique(head_ids):
            sel = torch.from_numpy(np.nonzero(head_ids == head)[0])
            res = self.net(head.item() if hasattr(head, "item") else head)(*[a[sel.to(a.device)] for a in args])
            if out is None:
                out = {k: torch.empty((len(head_ids),) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for k, v in res.items()}
            for k, v in res.items():
                out[k][sel.to(v.device)] = v
        return out
