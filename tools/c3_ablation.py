import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
from rgbmanip_amd import _lib, synth
from rgbmanip_amd.adapose import AdaPoseNet
lib = _lib.load()
B = 128
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
inp = synth.adapose_inputs(8, seed=0)
inp = {k: torch.from_numpy(np.concatenate([v] * (B // 8), 0)).cuda() for k, v in inp.items()}
def run():
    return net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
names = ["conv0t", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"]
for flags in (0, 1, 2, 3):
    lib.rgbm_debug_flags(flags)
    run(); torch.cuda.synchronize()
    lib.rgbm_prof_start()
    for _ in range(3): run()
    torch.cuda.synchronize()
    st = (C.c_double * (4 * _lib.PROF_ROWS))(); lib.rgbm_prof_stop(st)
    st = np.array(list(st)).reshape(_lib.PROF_ROWS, 4)
    rows = st[16:26, 1] / 3 if sys.argv[1:] != ["bf16x3"] else None
    if rows is not None:
        print("flags", flags, " ".join(f"{n}={t:.3f}" for n, t in zip(names, rows)), flush=True)
    else:
        print("flags", flags, "bx3 3-D layers total", st[27, 1] / 3, flush=True)
lib.rgbm_debug_flags(0)
