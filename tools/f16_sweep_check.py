"""fp16 conv0 sweep: run-to-run bit stability (compared on the device, any batch size) and, for small batches, agreement with
the halo-tile conv0 (debug flag 4096 selects the tile kernel).  usage: f16_sweep_check.py [B] [runs] [dtype: fp16 (default) | bf16 = the persistent kernel on the f16 feature map]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth, _lib
from rgbmanip_amd.adapose import AdaPoseNet
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
inp = synth.adapose_inputs(B, seed=0)
sd = synth.adapose_state_dict(seed=0, prefix="module.")
net = AdaPoseNet(sd, dtype=(sys.argv[3] if len(sys.argv) > 3 else "fp16"), cost_impl=3, options={"sparse_dec": 0})      # the whole c0 volume is compared


def tap(name, per_view, flag=0):
    lib.rgbm_debug_flags(flag)
    net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"], stop_after=2)
    torch.cuda.synchronize()
    t = net.fetch(B, name, 2 * B * per_view).view(2 * B, -1).half()      # exact: the taps are fp16 storage
    lib.rgbm_debug_flags(0)
    return t


def c0(flag):
    return tap("c0", 24 * 224 * 224 * 8, flag).view(2 * B, 24, 224, 224, 8)


bad = 0
for name, shape in (("feat", (224, 224, 32)), ("c0", (24, 224, 224, 8))):
    pv = int(np.prod(shape))
    first = tap(name, pv)
    for r in range(1, runs):
        cur = tap(name, pv)
        views = ((first.view(torch.int16) != cur.view(torch.int16)).sum(dim=1)).cpu().numpy()
        if views.any():
            bad += 1
            v = int(np.flatnonzero(views)[0])
            where = (first[v].view(torch.int16) != cur[v].view(torch.int16)).view(*shape).nonzero().cpu().numpy()
            print(name, "run", r, "views that differ", int((views > 0).sum()), "elements", int(views.sum()), "first such view", v,
                  "index ranges", [(int(where[:, k].min()), int(where[:, k].max())) for k in range(len(shape))],
                  "max diff", float((first[v].float() - cur[v].float()).abs().max()))
            if len(shape) == 4:
                hist = {}
                for vv in np.flatnonzero(views)[:48]:
                    w_ = (first[vv].view(torch.int16) != cur[vv].view(torch.int16)).view(*shape).nonzero()
                    key = tuple(sorted(set(w_[:, 0].tolist())))
                    hist[key] = hist.get(key, 0) + 1
                print("   plane sets over the first 48 differing views:", hist)
                tiles = {}
                for z, y, x, c in where:
                    tiles.setdefault((int(y) // 12, int(x) // 16), set()).add((int(z), int(y) % 12))
                for k in list(tiles)[:6]:
                    print("   tile", k, "(plane, row in tile):", sorted(tiles[k])[:24])
        del cur
    del first
first = c0(0) if B <= 4 else None
print("B", B, "runs", runs, "runs that differ from the first:", bad)
if B <= 4:
    tile = c0(4096).float()
    d = (first.float() - tile).abs()
    print("vs tile conv0: max", float(d.max() / tile.abs().max()), "mean", float(d.mean() / tile.abs().mean()))
    assert float(d.max() / tile.abs().max()) < 2e-3
assert bad == 0
