"""Where does c0 of conv0_sweep_persistent_kernel differ from the one-tile kernel (debug flag 268435456)?  usage: debug_sweep_persistent.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth, _lib
from rgbmanip_amd.adapose import AdaPoseNet
lib = _lib.load()
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.adapose_inputs(2, seed=1).items()}
def c0(flag):
    _lib.check(lib.rgbm_debug_flags(flag))
    net = AdaPoseNet(synth.adapose_state_dict(seed=0, prefix="module."), dtype="bf16", options={"sparse_dec": 0})
    net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"], stop_after=2)
    r = net.fetch(2, "c0", 4 * 24 * 224 * 224 * 8).view(4, 24, 224, 224, 8).float().cpu().numpy()
    _lib.check(lib.rgbm_debug_flags(0))
    return r
a, b, a2 = c0(0), c0(1 << 28), c0(0)
print("persistent run-to-run identical:", np.array_equal(a, a2))
d = a != b
print("differing", d.sum(), "of", d.size, "max abs", np.abs(a - b).max())
print("by view:", d.sum(axis=(1, 2, 3, 4)))
print("by plane:", d.sum(axis=(0, 2, 3, 4)))
rows = d.sum(axis=(0, 1, 3, 4)); cols = d.sum(axis=(0, 1, 2, 4))
print("by row % 12:", [int(rows[i::12].sum()) for i in range(12)])
print("by col % 16:", [int(cols[i::16].sum()) for i in range(16)])
print("by row:", rows.tolist()[:60])
print("by channel:", d.sum(axis=(0, 1, 2, 3)))
