import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet
g = np.load("tests/golden/adapose_b2.npz")
inp = synth.adapose_inputs(2, seed=0)
sd = synth.adapose_state_dict(seed=0, prefix="module.")
def run(**kw):
    net = AdaPoseNet(sd, dtype="fp16", **kw)
    o = net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
    return {k: v.float().cpu().numpy() for k, v in o.items()}
a = run(cost_impl=3); b = run(cost_impl=3); c = run(cost_impl=2)
for k in ("view1_depth", "view1_nocs", "view1_r"):
    print(k, "err", float(np.abs(a[k] - g[k]).max() / np.abs(g[k]).max()), "repeat", float(np.abs(a[k] - b[k]).max()), "vs impl2", float(np.abs(a[k] - c[k]).max()))
