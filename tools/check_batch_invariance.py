"""Per-pose outputs of a batch of B poses against the same poses in a batch of B + 1 (and alone), per storage type and per
`ws_min_rows` setting: which kernel families are batch-invariant bit for bit.  usage: check_batch_invariance.py [dtype...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbmanip_amd import _lib, synth
from rgbmanip_amd.adapose import AdaPoseNet

lib = _lib.load()
KEYS = ["view1_nocs", "view1_depth", "view1_r", "view1_t", "view1_s"]
inp = synth.adapose_inputs(6, seed=2)
args = lambda idx: [torch.from_numpy(np.ascontiguousarray(inp[k][idx])).cuda() for k in ("img1", "choose1", "img2", "choose2", "P1", "P2", "depths")]  # noqa: E731
for dt in (sys.argv[1:] or ["fp32", "bf16", "bf16x3"]):
    for rows in (0, 65536, 1 << 30):
        lib.rgbm_set_tuning(b"ws_min_rows", rows)
        net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dt)
        outs = {}
        for name, idx in (("B4", [0, 1, 2, 3]), ("B5", [0, 1, 2, 3, 4]), ("B1", [0])):
            o = net(*args(idx))
            torch.cuda.synchronize()
            outs[name] = {k: o[k].cpu().numpy() for k in KEYS}
        d45 = max(float(np.abs(outs["B4"][k] - outs["B5"][k][:4]).max() / np.abs(outs["B4"][k]).max()) for k in KEYS)
        d41 = max(float(np.abs(outs["B4"][k][:1] - outs["B1"][k]).max() / np.abs(outs["B4"][k]).max()) for k in KEYS)
        print(f"{dt} ws_min_rows={rows}: B4 vs B5 {d45:.2e}   B4 vs B1 {d41:.2e}")
lib.rgbm_set_tuning(b"ws_min_rows", 0)
