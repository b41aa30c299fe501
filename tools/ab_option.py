"""Same-box A/B of one rgbm_adapose_set_option key at batch 256: ms per forward for each value, interleaved.
usage: ab_option.py <dtype> <key> <v0> <v1> [...]      key "debug": the values are rgbm_debug_flags words (read at launch time) on one net;
key "tuning:<name>": the values go to rgbm_set_tuning(<name>, v) on one net (e.g. tuning:gemm_kernel 0 1 2)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet

dtype, key, vals = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3:]]
B = 256
inp = synth.adapose_inputs(16, seed=0)
inp = {k: torch.from_numpy(np.concatenate([v] * (B // 16), 0)).cuda() for k, v in inp.items()}
sd = synth.adapose_state_dict(seed=0)
if key == "debug" or key.startswith("tuning:"):
    from rgbmanip_amd import _lib
    one = AdaPoseNet(sd, dtype=dtype)
    nets = {v: (one, v) for v in vals}
else:
    nets = {v: AdaPoseNet(sd, dtype=dtype, options={key: v}) for v in vals}
def run(net):
    if isinstance(net, tuple):
        if key == "debug":
            _lib.load().rgbm_debug_flags(net[1])
        else:
            _lib.check(_lib.load().rgbm_set_tuning(key.split(":", 1)[1].encode(), net[1]), "set_tuning")
        net = net[0]
    return net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
for net in nets.values():
    run(net)
torch.cuda.synchronize()
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(nets[v])
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 3)
for v in vals:
    print(dtype, key, v, "ms per forward:", " ".join(f"{t:.2f}" for t in res[v]), " median", f"{np.median(res[v]):.2f}")
