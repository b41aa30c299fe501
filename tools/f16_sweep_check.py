"""fp16 conv0 sweep: run-to-run bit stability and agreement with the halo-tile conv0 (debug flag 4096 selects the tile kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth, _lib
from rgbmanip_amd.adapose import AdaPoseNet
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
inp = synth.adapose_inputs(B, seed=0)
sd = synth.adapose_state_dict(seed=0, prefix="module.")
def c0(flag):
    lib.rgbm_debug_flags(flag)
    net = AdaPoseNet(sd, dtype="fp16", cost_impl=3)
    net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"], stop_after=2)
    torch.cuda.synchronize()
    t = net.fetch(2, "c0", 2 * B * 24 * 224 * 224 * 8).view(2 * B, 24, 224, 224, 8).float().cpu().numpy()
    lib.rgbm_debug_flags(0)
    return t
runs = [c0(0) for _ in range(5)]
print("repeat diffs", [float(np.abs(runs[0] - r).max()) for r in runs[1:]])
tile = c0(4096)
d = np.abs(runs[0] - tile)
print("vs tile conv0: max", float(d.max() / np.abs(tile).max()), "mean", float(d.mean() / np.abs(tile).mean()))
assert all(np.array_equal(runs[0], r) for r in runs[1:]) and d.max() / np.abs(tile).max() < 2e-3
