import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet
def rel(a,b): return float(np.abs(a.astype(np.float64)-b).max()/max(np.abs(b).max(),1e-12))
inputs = synth.adapose_inputs(2, seed=0)
bad = {k: v.copy() for k, v in inputs.items()}
bad["P2"][1] = 0.0; bad["P2"][1,3,3] = 1.0
def run(net, inp, **kw):
    o = net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"], **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}
for impl in (0, 1, 2):
    net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype="fp32", cost_impl=impl)
    A = run(net, inputs); Bd = run(net, bad); Cc = run(net, inputs)
    print("impl", impl, "A vs C (stale?)", {k: rel(A[k][0], Cc[k][0]) for k in ("view1_depth","view1_nocs","view1_r","view2_depth")})
    print("impl", impl, "A vs B pose0", {k: rel(Bd[k][0], A[k][0]) for k in ("view1_depth","view1_nocs","view1_r","view2_depth")})
    print("impl", impl, "B pose1 finite", {k: bool(np.isfinite(Bd[k][1]).all()) for k in ("view1_depth","view2_depth","view1_r")})
    # bisect: intermediates of pose0 view1 between clean and bad
    V=4; B=2
    def grab(inp):
        run(net, inp, stop_after=2)
        d = {}
        for name, n in (("c0", V*24*224*224*8), ("c2", V*12*112*112*16), ("c6", V*3*28*28*64), ("u7", V*6*56*56*32), ("u9", V*12*112*112*16), ("u11", V*24*224*224*8), ("prob", V*1024*24)):
            x = net.fetch(B, name, n).view(V, -1)
            d[name] = x[0].cpu().numpy().copy(); d[name+"_v2"] = x[2].cpu().numpy().copy()
        return d
    g1 = grab(inputs); g2 = grab(bad)
    print("impl", impl, "stage diffs pose0:", {k: rel(g2[k], g1[k]) for k in g1})
